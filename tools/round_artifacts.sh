#!/bin/bash
# Round artefacts on the GPU box -> gpurun_out/<tag>/ (copied into profiles/ by hand afterwards):
#   bench lines (V=4 default with cpu_baseline; V=8; V=8 x 64^2), rocprofv3 kernel-trace stats of the default command,
#   PMC traffic / MFMA passes per workload, shard emulation, step trace summary.
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
# ONE kernel mix everywhere: every process loads the committed tuner cache mvdfusion_amd/tuned/gemm_f16.json (hip.load_default_tuned;
# tools/tune_all.py wrote it) -- the bench runs, the kernel traces and the counter passes below launch identical kernels, and so does the
# driver's own run (bench config.gemm_tuning says how many problems were tuned in the run: 0 expected)
python bench.py --steps 100 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log
python bench.py --views 8 --steps 50 --warmup 3 --no-cpu-baseline --shard-emulate 0/8,7/8 > $O/bench_v8_s32.json 2> $O/bench_v8_s32.log
python bench.py --views 8 --latent 64 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_v8_s64.json 2> $O/bench_v8_s64.log
# the as-shipped inference view count (configs/mvd_gso.yaml:97), the D = 3 geometry, the training step, the VAE, the 2-rank bench contract
python bench.py --views 15 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_v15_s32.json 2> $O/bench_v15_s32.log
python bench.py --views 8 --depth-samples 3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_v8_s32_d3.json 2> $O/bench_v8_s32_d3.log
python tools/bench_train.py --steps 5 > $O/train_step.json 2> $O/train_step.log
python tools/bench_vae.py --steps 20 > $O/decode_n1.json 2> $O/decode_n1.log
python tools/bench_vae.py --steps 20 --encode > $O/encode_n1.json 2> $O/encode_n1.log
cd /tmp && export TMPDIR=/tmp
out=$O/prof
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
cp $out/bench_kernel_stats.csv $O/bench_n1_kernel_stats.csv
python $R/tools/trace_summary.py $out $O/step_trace_v4.json > $O/step_trace_v4.txt
rm -rf $out
# the rank share of the emulated 8-way view-parallel job: per-kernel table of ITS steps (the shard steps are the last ones of the run)
cd /tmp
rocprofv3 --kernel-trace -d $out -o bench --output-format csv -- python $R/bench.py --views 8 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --shard-emulate 0/8 > $O/prof_shard.log 2>&1
TRACE_LAST_GROUP=1 python $R/tools/trace_summary.py $out $O/step_trace_v8_shard0of8.json > $O/step_trace_v8_shard0of8.txt
rm -rf $out
# weight prefetch on / off, alternating, same box
cd $R
for i in 1 2; do
  for m in ws 0; do
    MVD_PREFETCH=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/prefetch_${m}_$i.json 2> /dev/null
  done
done
python - <<PY > $O/prefetch_ab.txt
import json
for m in ("ws", "0"):
    for i in (1, 2):
        d = json.loads(open("$O/prefetch_%s_%d.json" % (m, i)).read().strip().splitlines()[-1])
        print("MVD_PREFETCH=%s run %d: %.2f steps/s %.3f ms/step" % (m, i, d["value"], d["ms_per_step"]))
PY
bash $R/tools/prof_train.sh $tag/train > $O/prof_train.log 2>&1
cd /tmp
bash $R/tools/pmc_traffic.sh ${tag}_v4_s32_d1 > $O/pmc_traffic_v4.log 2>&1
bash $R/tools/pmc_traffic.sh ${tag}_v8_s32_d1 --views 8 > $O/pmc_traffic_v8.log 2>&1
bash $R/tools/pmc_traffic.sh ${tag}_v8_s64_d1 --views 8 --latent 64 > $O/pmc_traffic_v8s64.log 2>&1
bash $R/tools/pmc_mfma.sh ${tag} > $O/pmc_mfma.log 2>&1
# the bench line again, now that the traffic files of this session exist (roofline.traffic is read from profiles/, see README)
ls $R/gpurun_out | grep pmc_${tag}

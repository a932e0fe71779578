set -x
mkdir -p gpurun_out
B="python bench.py --no-secondary --no-cpu-baseline --steps 50 --tune-cache tools/_tmp_tuned.json"
for i in 1 2; do
  for sm in 0 1048576 8388608; do
    MVD_SELF_PREFETCH_MIN=$sm timeout 600 $B > gpurun_out/r06_selfpf_${sm}_$i.json 2>> gpurun_out/r06_selfpf.err
  done
done
for sm in 0 1048576; do
  MVD_SELF_PREFETCH_MIN=$sm timeout 600 python bench.py --views 8 --no-cpu-baseline --steps 30 --shard-emulate 0/8 --tune-cache tools/_tmp_tuned.json > gpurun_out/r06_selfpf_v8_${sm}.json 2>> gpurun_out/r06_selfpf.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_selfpf_*.json")):
    d = json.load(open(f))
    print(f, round(d["value"], 2), round(d["ms_per_step"], 3), d.get("shard_emulate", {}).get("ms_per_step"))
PY
# training step: same-box A/B against the round-5 tree
for i in 1 2; do
  ( cd _ab_r05 && timeout 900 python tools/bench_train.py --steps 5 ) > gpurun_out/r06_train_r05tree_$i.json 2>> gpurun_out/r06_train_ab.err
  timeout 900 python tools/bench_train.py --steps 5 > gpurun_out/r06_train_new_$i.json 2>> gpurun_out/r06_train_ab.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_train_*_[12].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["s_per_step"])
PY

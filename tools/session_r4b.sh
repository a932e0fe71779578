# GPU session r4b: new tests, bench of the tree incl. the register-staged ws loader in the tuner's candidate set, ws probe variants A/B, precision sweep
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q -k "gridattn_vs_reference or fused_equals_unfused or step_mc320_v15 or step_mc320_v8_d3 or config4 or feed_prev or configurations_agree or conv3x3 or test_gemm or attention" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune-cache $O/tune_v4.json > $O/bench_main.json 2> $O/bench_main.log; grep "bench\]" $O/bench_main.log | head -30; cut -c1-300 $O/bench_main.json
for v in 1 2 3; do
  MVD_HIP_LIB=tools/probes/libmvd_hip_wsv$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune-cache $O/tune_v4.json > $O/bench_wsv$v.json 2> $O/bench_wsv$v.log; cut -c1-200 $O/bench_wsv$v.json
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune-cache $O/tune_v4.json > $O/bench_main2.json 2> $O/bench_main2.log; cut -c1-200 $O/bench_main2.json
MVD_TUNE_EXCLUDE_LOOPS=8 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_no_wsreg.json 2> $O/bench_no_wsreg.log; cut -c1-200 $O/bench_no_wsreg.json
timeout 900 python tools/prec_sweep.py --reps 2 > $O/prec_sweep.json 2> $O/prec_sweep.log; tail -40 $O/prec_sweep.log
timeout 600 python bench.py --views 15 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_v15.json 2> $O/bench_v15.log; cut -c1-300 $O/bench_v15.json

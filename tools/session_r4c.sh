# GPU session r4c: full GPU suite at the new default (f16x3), bench with the register-staged plain kernel (loop 9) in / out of the tuner's set
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune-cache $O/tune_v4.json > $O/bench_main.json 2> $O/bench_main.log; grep "bench\]" $O/bench_main.log | head -30; cut -c1-300 $O/bench_main.json
MVD_TUNE_EXCLUDE_LOOPS=9 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_no_reg.json 2> $O/bench_no_reg.log; cut -c1-200 $O/bench_no_reg.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune-cache $O/tune_v4.json > $O/bench_main2.json 2> $O/bench_main2.log; cut -c1-200 $O/bench_main2.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --precision f16x4 > $O/bench_x4.json 2> $O/bench_x4.log; cut -c1-200 $O/bench_x4.json
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -30 $O/tests.log
timeout 900 python bench.py --views 8 --steps 20 --warmup 3 --no-cpu-baseline --shard-emulate 0/8,7/8 > $O/bench_v8.json 2> $O/bench_v8.log; cut -c1-300 $O/bench_v8.json

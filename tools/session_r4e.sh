# GPU session r4e: full GPU suite of the final tree, headline bench, operand-delivery counters
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune-cache $O/tune_v4.json > $O/bench_main.json 2> $O/bench_main.log; grep "bench\]" $O/bench_main.log | head -8; cut -c1-200 $O/bench_main.json
timeout 1500 bash tools/pmc_delivery.sh r04 --tune-cache $O/tune_v4.json > $O/pmc_delivery.log 2>&1; tail -60 $O/pmc_delivery.log
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -8 $O/tests.log

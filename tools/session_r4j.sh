set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; cut -c1-300 $O/bench_default.json
timeout 900 python bench.py --train-step > $O/train_step.json 2> $O/train_step.log; cut -c1-500 $O/train_step.json
timeout 900 bash tools/prof_train.sh --steps 3 > $O/prof_train.log 2>&1
cp gpurun_out/train_prof_stats.csv $O/train_kernel_stats.csv
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log

set -x
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=30 ) > gpurun_out/r06_pytest_c.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_pytest_c.log
timeout 900 python tools/bench_train.py --steps 5 > gpurun_out/r06_train_step_a.json 2> gpurun_out/r06_train_step_a.log
tail -n 3 gpurun_out/r06_pytest_c.log

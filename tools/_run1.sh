set -x
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=80 ) > gpurun_out/r06_pytest_durations.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_pytest_durations.log
V=4 timeout 600 python tools/gemm_shapes.py > gpurun_out/r06_gemm_shapes_v4.log 2>&1
timeout 600 python bench.py --no-secondary > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
tail -3 gpurun_out/r06_pytest_durations.log

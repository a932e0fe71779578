set -x
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 ) > gpurun_out/r06_pytest_b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_pytest_b.log
# crash-proofing: every tensor its own allocation -> an out-of-bounds access faults deterministically
( time PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_backward.py -x -q -m gpu -p no:cacheprovider ) > gpurun_out/r06_pytest_no_caching.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_pytest_no_caching.log
( time MVD_TUNE_CACHE=0 MVD_TUNE_SPLITS=1 timeout 1500 python tools/tune_all.py --quick ) > gpurun_out/r06_tune_quick.log 2>&1
cp mvdfusion_amd/tuned/gemm_f16.json gpurun_out/r06_gemm_f16_quick.json
for i in 1 2; do
  MVD_TUNE_CACHE=0 timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 50 > gpurun_out/r06_bench_notune_$i.json 2>> gpurun_out/r06_bench_b.err
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 50 > gpurun_out/r06_bench_cache_$i.json 2>> gpurun_out/r06_bench_b.err
done
timeout 600 python bench.py --views 8 --no-cpu-baseline --shard-emulate 0/8 > gpurun_out/r06_bench_v8_shard_cache.json 2>> gpurun_out/r06_bench_b.err
tail -3 gpurun_out/r06_pytest_b.log gpurun_out/r06_pytest_no_caching.log gpurun_out/r06_tune_quick.log

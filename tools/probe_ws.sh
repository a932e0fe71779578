mkdir -p gpurun_out/r3p
python -m pytest tests/test_gpu_ops.py -x -q -k "qkv or geglu or configurations" 2>&1 | tail -3
for x in "" 7 "6,7" "" 7; do MVD_TUNE_EXCLUDE_LOOPS=$x python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r3p/bench_$x.json 2> gpurun_out/r3p/bench_$x.err
python -c "
import json; d=json.load(open('gpurun_out/r3p/bench_$x.json')); print('exclude [$x]', d['value'], d['ms_per_step'], d['roofline']['gemm_share_of_step_ms'])"; done

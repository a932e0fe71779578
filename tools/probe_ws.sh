mkdir -p gpurun_out/r3q
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -k "gridattn or denoise_step or graph_replay" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r3q/bench_$i.json 2> gpurun_out/r3q/bench_$i.err
python -c "
import json; d=json.load(open('gpurun_out/r3q/bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['gemm_share_of_step_ms'], {k:(round(v['ms_per_step'],3), v.get('mfma_pipe_frac')) for k,v in d['roofline_groups'].items()})"; done

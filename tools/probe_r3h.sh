mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_dense or configurations_agree or groupnorm_statistics or layernorm_fold or geglu or qkv or activation_b" 2>&1 | tail -4
T5="81:1,83:1,85:1,85:0"
for s in conv16 conv8 ff1_32 ff1_16 ff1_8 ff2_16 ff2_8 qkv8; do python tools/gemm_probe.py $s 17:1,21:1,25:1,21:0,25:0,$T5 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r3h/probe256.log 2>&1
cat gpurun_out/r3h/probe256.log

#!/bin/bash
# Profiling build of the GEMM kernels with cycle stamps (-DMVD_STAMP): tools/probes/libmvd_hip_stamp.so, read by tools/probes/ws_stamp.py.
set -e
cd "$(dirname "$0")/../../mvdfusion_amd/csrc"
P=../../tools/probes
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMVD_STAMP -c gemm.hip -o $P/gemm_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libmvd_hip_stamp.so api.o $P/gemm_stamp.o norm.o attention.o elementwise.o gridattn.o gridattn_fused.o backward.o

#!/bin/bash
# Profiling build of the fused GridAttn kernel with cycle accounting (-DMVD_G4_STAMP): tools/probes/libmvd_hip_g4stamp.so (tools/probes/g4_time.py --stamp)
set -e
cd "$(dirname "$0")/../../mvdfusion_amd/csrc"
P=../../tools/probes
. $P/objs.sh
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMVD_G4_STAMP -c gridattn_fused.hip -o $P/gridattn_fused_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libmvd_hip_g4stamp.so $GEMM_OBJS gemm_pt.o $REST_OBJS $P/gridattn_fused_stamp.o

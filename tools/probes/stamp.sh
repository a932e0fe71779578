#!/bin/bash
# Profiling build of the GEMM kernels with cycle stamps (-DMVD_STAMP): tools/probes/libmvd_hip_stamp.so, read by tools/probes/ws_stamp.py.
set -e
cd "$(dirname "$0")/../../mvdfusion_amd/csrc"
P=../../tools/probes
. $P/objs.sh
OBJS=""
for o in $GEMM_OBJS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMVD_STAMP -c ${o%.o}.hip -o $P/${o%.o}_stamp.o &
  OBJS="$OBJS $P/${o%.o}_stamp.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libmvd_hip_stamp.so $OBJS gemm_pt.o $REST_OBJS gridattn_fused.o

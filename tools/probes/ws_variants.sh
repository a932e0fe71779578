#!/bin/bash
# Probe builds of csrc/gemm_ws.hip with -DMVD_WS_VARIANT=<1|2|3> (gemm_ws_kernel: 1 = scheduling barrier at the end of a k-tile, 2 = consumer
# wavefronts at s_setprio 3, 3 = both) into tools/probes/libmvd_hip_wsv<N>.so, for same-box A/B runs:
#   MVD_HIP_LIB=tools/probes/libmvd_hip_wsv1.so python bench.py ...
set -e
cd "$(dirname "$0")/../../mvdfusion_amd/csrc"
P=../../tools/probes
. $P/objs.sh
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMVD_WS_VARIANT=$v -c gemm_ws.hip -o $P/gemm_wsv$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libmvd_hip_wsv$v.so ${GEMM_OBJS/gemm_ws.o/$P/gemm_wsv$v.o} gemm_pt.o $REST_OBJS gridattn_fused.o ) &
done
wait

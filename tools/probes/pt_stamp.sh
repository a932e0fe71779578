#!/bin/bash
# Profiling builds of the persistent GEMM with cycle accounting (-DMVD_PT_STAMP): tools/probes/libmvd_hip_ptstamp[_vN].so, read by
# tools/probes/pt_stamp.py.  Arguments: MVD_PT_VARIANT values to build besides the plain one (ablations, see csrc/gemm_pt.hip).
set -e
cd "$(dirname "$0")/../../mvdfusion_amd/csrc"
P=../../tools/probes
. $P/objs.sh
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMVD_PT_STAMP -DMVD_PT_VARIANT=$1 -c gemm_pt.hip -o $P/gemm_pt_stamp$2.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libmvd_hip_ptstamp$2.so $GEMM_OBJS $P/gemm_pt_stamp$2.o $REST_OBJS gridattn_fused.o
}
build 0 "" &
for v in "$@"; do build $v _v$v & done
wait

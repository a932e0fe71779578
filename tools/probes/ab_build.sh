#!/bin/bash
# Build csrc/gemm.hip of another git revision into tools/probes/libmvd_hip_base.so (same ABI; the other objects are the current ones), for
# same-box A/B runs:   MVD_HIP_LIB=tools/probes/libmvd_hip_base.so python bench.py ...
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/../.."
mkdir -p /tmp/mvd_ab/csrc /tmp/mvd_ab/include && git show $REV:mvdfusion_amd/csrc/gemm.hip > /tmp/mvd_ab/csrc/gemm.hip
git show $REV:mvdfusion_amd/csrc/common.hpp > /tmp/mvd_ab/csrc/common.hpp
git show $REV:include/mvd_hip.h > /tmp/mvd_ab/include/mvd_hip.h
sed -i 's#"../../include/mvd_hip.h"#"../include/mvd_hip.h"#' /tmp/mvd_ab/csrc/gemm.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c /tmp/mvd_ab/csrc/gemm.hip -o tools/probes/gemm_base.o
cd mvdfusion_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libmvd_hip_base.so api.o ../../tools/probes/gemm_base.o norm.o attention.o elementwise.o gridattn.o gridattn_fused.o backward.o

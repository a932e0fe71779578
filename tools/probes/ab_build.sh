#!/bin/bash
# Build the library of another git revision into tools/probes/libmvd_hip_base.so (same ABI assumed), for same-box A/B runs:
#   tools/probes/ab_build.sh <rev>;  MVD_HIP_LIB=tools/probes/libmvd_hip_base.so python bench.py ...
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/../.."
rm -rf /tmp/mvd_ab && mkdir -p /tmp/mvd_ab
git archive $REV mvdfusion_amd/csrc include | tar -x -C /tmp/mvd_ab
( cd /tmp/mvd_ab/mvdfusion_amd/csrc && python build.py > /tmp/mvd_ab/build.log 2>&1 )
cp /tmp/mvd_ab/mvdfusion_amd/csrc/libmvd_hip.so tools/probes/libmvd_hip_base.so

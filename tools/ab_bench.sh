#!/bin/bash
# Same-box A/B of the working tree's library against tools/probes/libmvd_hip_base.so (tools/probes/ab_build.sh <rev>), alternating runs, both legs on
# the committed tuner choices (MVD_TUNE_CACHE_ANY=1: the fingerprint of edited GEMM sources is ignored):  tools/ab_bench.sh <out dir> [bench args]
O=${1:-gpurun_out/ab}; shift
mkdir -p $O
export MVD_TUNE_CACHE_ANY=1
for i in 1 2 3; do
  MVD_HIP_LIB=tools/probes/libmvd_hip_base.so python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $O/base_$i.json 2> $O/base_$i.log
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $O/new_$i.json 2> $O/new_$i.log
done
python - <<PY | tee $O/ab.txt
import json
for leg in ("base", "new"):
    v = []
    for i in (1, 2, 3):
        d = json.loads(open("$O/%s_%d.json" % (leg, i)).read().strip().splitlines()[-1])
        v.append((d["value"], d["ms_per_step"]))
    print(leg, " ".join("%.2f steps/s (%.3f ms)" % x for x in v), "| tuning:", d["config"]["gemm_tuning"])
PY

#!/bin/bash
# The CPU baseline on ALL host cores of the GPU box, next to the default 16-thread figure:
#   writes gpurun_out/<tag>_cpu_allcores.json (copy to profiles/<tag>_cpu_allcores.json; bench.py attaches it as
#   cpu_baseline.all_host_cores).  One timed step after one warm-up step: PyTorch eager oversubscribes badly on big hosts.
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 5 --warmup 2 --cpu-threads 0 --cpu-steps 1 2> gpurun_out/${tag}_cpu_allcores.log | tail -1 > /tmp/line.json
python - <<PY
import json, os
j = json.load(open("/tmp/line.json"))
cb = j["cpu_baseline"]
cb.pop("all_host_cores", None)
cb["host_cpu_count"] = os.cpu_count()
json.dump(cb, open("$R/gpurun_out/${tag}_cpu_allcores.json", "w"), indent=1)
print(cb)
PY

#!/bin/bash
# A/B kernel trace of the bench step: tools/ab_trace.sh <tag> [bench flags]  ->  gpurun_out/trace_<tag>.{json,txt}
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/trace_${tag}
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline "$@" > $out.log 2>&1
python $R/tools/trace_summary.py $out $out.json > $out.txt
rm -rf $out
head -30 $out.txt

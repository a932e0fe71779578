#!/bin/bash
# rocprofv3 kernel trace of the default bench run; writes gpurun_out/prof_<tag>/ and a per-kernel summary CSV.
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python $R/bench.py --steps ${2:-100} --warmup 5 --no-cpu-baseline > $out.log 2>&1
tail -1 $out.log | cut -c1-300
ls $out

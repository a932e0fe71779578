#!/bin/bash
# rocprofv3 kernel trace of the training step (tools/bench_train.py): the first run tunes and writes the tuner cache, the traced run loads it.
#   tools/prof_train.sh <out prefix under gpurun_out/> [bench_train arguments]
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/$name
mkdir -p $out
python $R/tools/bench_train.py --tuned $out/tuned.json --cprofile $out/cprofile.txt "$@" > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-400
rm -rf $out/trace
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $R/tools/bench_train.py --tuned $out/tuned.json --steps 2 "$@" > $out/traced.log 2>&1
TRACE_LAST_STEPS=2 python $R/tools/trace_summary.py $out/trace $out/step_trace.json > $out/step_trace.txt
head -60 $out/step_trace.txt
rm -rf $out/trace

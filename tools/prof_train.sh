#!/bin/bash
# rocprofv3 kernel-trace stats of the training step (tools/bench_train.py) -> gpurun_out/train_prof_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/train_prof
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python $R/tools/bench_train.py "$@" > $out.log 2>&1
tail -1 $out.log
cp $out/t_kernel_stats.csv $R/gpurun_out/train_prof_stats.csv
rm -rf $out
head -25 $R/gpurun_out/train_prof_stats.csv | cut -c1-150

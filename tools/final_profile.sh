#!/bin/bash
# Round artefacts: the bench line (with cpu_baseline) and the rocprofv3 kernel-trace stats of the same workload.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && python bench.py --steps 200 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.log
tail -1 gpurun_out/bench_n1.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_final
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o bench --output-format csv -- python $R/bench.py --steps 200 --warmup 5 --no-cpu-baseline > $out.log 2>&1
grep -v "rocprofv3\|amdgpu.ids\|Opened" $out.log | tail -1 | cut -c1-200
ls $out

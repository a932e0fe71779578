#!/bin/bash
# HBM traffic of every kernel of the bench step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share
# one: TCC has 4 slots, they cost 3 + 2), counters only -- no trace domains.  Summarised by tools/pmc_traffic.py.
#   tools/pmc_traffic.sh <tag> [bench.py workload flags, e.g. --views 8 --latent 64]
tag=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmc_${tag}_$c
  rm -rf $out
  rocprofv3 --pmc $c --output-format csv -d $out -o pmc -- python $R/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out.log 2>&1
  ls $out | head -3
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE $R/gpurun_out/pmc_${tag}_traffic.json "$@"
rm -rf $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE

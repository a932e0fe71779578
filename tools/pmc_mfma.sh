#!/bin/bash
# MFMA-pipe utilisation and wave-time breakdown of every kernel of the bench step (one rocprofv3 --pmc pass, SQ + GRBM
# counters only, no trace domains).  Summarised by tools/pmc_mfma.py.
tag=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_${tag}_mfma
rm -rf $out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $out -o pmc -- python $R/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out.log 2>&1
python $R/tools/pmc_mfma.py $out $R/gpurun_out/pmc_${tag}_mfma.json
rm -rf $out

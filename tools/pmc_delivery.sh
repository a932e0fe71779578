#!/bin/bash
# Operand-delivery counters of the bench step, one small rocprofv3 --pmc pass per counter group (counters only, no trace domains):
# LDS activity / conflicts, L1 (TCP) -> L2 (TCC) requests and latency, L2 hit / miss / tag stalls, texture-addresser stalls.
# A group whose counter names this rocprofv3 does not know fails fast and is skipped.  -> gpurun_out/pmc_<tag>_delivery_<group>.json
tag=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc_${tag}_counter_list.txt 2>&1
declare -A G
G[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
G[tcp]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
G[ta]="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
G[vmem]="SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
for g in ${PMC_GROUPS:-lds tcp tcc ta vmem}; do
  out=$R/gpurun_out/pmc_${tag}_delivery_$g
  rm -rf $out
  timeout 600 rocprofv3 --pmc ${G[$g]} --output-format csv -d $out -o pmc -- python $R/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out.log 2>&1
  python $R/tools/pmc_generic.py $out $out.json 2>&1 | head -20
  rm -rf $out
done

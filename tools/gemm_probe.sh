#!/bin/bash
# tools/gemm_probe.sh <shape> <variants> <tag>: timing, then two rocprofv3 --pmc passes (SQ/GRBM, then TCC/TCP) of the variants
shape=$1; variants=$2; tag=${3:-probe}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$tag
python $R/tools/gemm_probe.py $shape $variants > $R/gpurun_out/$tag/time_$shape.log 2>&1
cat $R/gpurun_out/$tag/time_$shape.log | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  out=$R/gpurun_out/$tag/pmc_${shape}_$i
  rm -rf $out
  MVD_PROBE_PLAIN=1 rocprofv3 --pmc $pass --output-format csv -d $out -o r -- python $R/tools/gemm_probe.py $shape $variants > $out.log 2>&1
  python $R/tools/gemm_probe_pmc.py $out | tee $out.txt
  rm -rf $out
done

cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for st in 2 3; do
 for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  out=$R/gpurun_out/pmc_st${st}_$(echo $pass | cut -c1-12 | tr ' ' _)
  rm -rf $out
  MVD_BENCH_NOGRAPH=1 MVD_BENCH_CFG=$((7 + 2 * (st - 2))) rocprofv3 --pmc $pass --kernel-trace -d $out -o r -- python $R/tools/gemm_bench.py 4 "conv 32^2 640" > $out.log 2>&1
  tail -2 $out.log
 done
done
ls $R/gpurun_out/pmc_st2*/ 

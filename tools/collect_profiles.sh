#!/bin/bash
# Copy the round's artefacts (written by tools/round_artifacts.sh on the GPU box, merged back under gpurun_out/) into profiles/.
tag=${1:-r04}
G=gpurun_out; O=$G/$tag; P=profiles
for f in bench_n1.json bench_n1.log bench_n1_kernel_stats.csv bench_v8_s32.json bench_v8_s64.json bench_v8_s32_d3.json bench_v15_s32.json train_step.json decode_n1.json encode_n1.json step_trace_v4.json step_trace_v4.txt; do
  [ -f $O/$f ] && cp $O/$f $P/${tag}_$f
done
for w in v4_s32_d1 v8_s32_d1 v8_s64_d1; do
  [ -f $G/pmc_${tag}_${w}_traffic.json ] && cp $G/pmc_${tag}_${w}_traffic.json $P/${tag}_pmc_traffic_${w}.json
done
[ -f $G/pmc_${tag}_mfma.json ] && cp $G/pmc_${tag}_mfma.json $P/${tag}_pmc_mfma.json
[ -f $G/dist_2rank_one_gpu.json ] && cp $G/dist_2rank_one_gpu.json $P/${tag}_dist_2rank_one_gpu.json
ls $P | grep ${tag}_

#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/: kernel stats + three PMC passes of the default bench command.
#   gpurun -- 'tools/profile_step.sh <tag>'   then   python tools/summarize_profiles.py <tag> ... (see that file)
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# B_EXTRA: extra bench.py arguments, e.g. "--b 8 --dhw 128,128,64" for BASELINE config C4 (PROFILE_LABEL then names the configuration in the summaries' headers)
B="python $R/bench.py --steps ${B_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alone --no-secondary $B_EXTRA"
LABEL=${PROFILE_LABEL:-"b=32, 64x64x32, bf16"}
# (1) the default command: weight gradients on the side stream, kernels overlap (durations are times under contention, as in bench.py's roofline)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/overlap -- $B > $R/gpurun_out/$TAG.overlap.log 2>&1
# (2) PCRL_WGRAD_STREAM=0: one stream, every kernel alone on the chip (per-kernel quality; bench.py's roofline.alone); the counter passes serialize kernels anyway
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- $B > $R/gpurun_out/$TAG.stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/fetch -- $B > $R/gpurun_out/$TAG.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/write -- $B > $R/gpurun_out/$TAG.write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/sq -- $B > $R/gpurun_out/$TAG.sq.log 2>&1
find $R/gpurun_out/$TAG -name "*.csv" | grep -v agent_info | xargs ls -la | awk '{print $5, $9}'
# the counter CSVs hold one row per dispatch and counter: keep only the two big kernels and the HBM-bound streaming kernels (the merge-back limit is 64 MiB)
for d in fetch write sq; do
  f=$(find $R/gpurun_out/$TAG/$d -name "*counter_collection.csv")
  head -1 $f > $R/gpurun_out/$TAG/$d.csv
  grep "brick16_conv_kernel\|brick_conv_kernel\|wgrad_brick_kernel\|wgrad_brick27_kernel\|wgrad_brick_upc2_kernel\|wgrad_upc8_kernel\|igemm_kernelIDF16bLi[0-9]*ELi[34]E\|bn_bwd_apply_rc_kernel\|bn_bwd_reduce_kernel\|bn_apply_rc_kernel\|bn_apply_gap_kernel\|bn_apply_pool_kernel\|bn_bwd_apply_pool_kernel\|bn_bwd_reduce_pool_kernel\|maxpool_\|gap_bwd_kernel\|coltile_sum_kernel" $f >> $R/gpurun_out/$TAG/$d.csv
  rm -f $f $(find $R/gpurun_out/$TAG/$d -name "*kernel_trace.csv")
done
cd $R && PROFILE_CMD="python bench.py ($LABEL), default: three streams (second view, weight gradients + side branches), kernels overlap (sum of kernel time > step time)" python tools/summarize_profiles.py ${TAG}_overlap $(find gpurun_out/$TAG/overlap -name "*kernel_stats.csv") 8 > /dev/null 2>&1
cp profiles/${TAG}_overlap_kernel_stats.txt gpurun_out/ 2>/dev/null
export PROFILE_CMD="PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 python bench.py ($LABEL): one stream, every kernel alone on the chip"
python tools/summarize_profiles.py $TAG $(find gpurun_out/$TAG/stats -name "*kernel_stats.csv") 8 gpurun_out/$TAG/fetch.csv gpurun_out/$TAG/write.csv gpurun_out/$TAG/sq.csv > gpurun_out/$TAG.summary.txt 2>&1
cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc.json gpurun_out/ 2>/dev/null
cp $(find gpurun_out/$TAG/stats -name "*kernel_stats.csv") gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
tail -30 gpurun_out/$TAG.summary.txt
cp $(find gpurun_out/$TAG/overlap -name "*kernel_stats.csv") gpurun_out/${TAG}_overlap_rocprofv3_kernel_stats.csv
head -12 gpurun_out/${TAG}_overlap_kernel_stats.txt

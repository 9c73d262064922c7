cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/tl.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/tl -name "*kernel_trace.csv") 8 3 > gpurun_out/r04q_timeline.txt 2>&1
head -45 gpurun_out/r04q_timeline.txt
rm -rf gpurun_out/tl

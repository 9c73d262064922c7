#!/bin/bash
# gpurun -- 'tools/phase_map.sh <out.txt> [bin_us]': three-stream kernel trace of bench.py -> tools/phase_map.py
OUT=$1; BIN=${2:-250}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $(dirname $R/$OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pm
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pm -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/pm.log 2>&1
cd $R
python tools/phase_map.py $(find gpurun_out/pm -name "*kernel_trace.csv") $BIN > $OUT 2>&1
python tools/timeline.py $(find gpurun_out/pm -name "*kernel_trace.csv") 8 3 >> $OUT 2>&1
rm -rf gpurun_out/pm
cat $OUT

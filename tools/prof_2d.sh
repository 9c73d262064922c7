#!/bin/bash
# rocprofv3 kernel stats of the 2D step probe; prints the per-kernel table
TAG=${1:-prof2d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/tools/bench_2d.py --steps 4 --warmup 2 "$@" > $R/gpurun_out/$TAG.log 2>&1
cd $R && python tools/summarize_profiles.py $TAG $(find gpurun_out/$TAG -name "*kernel_stats.csv") 6 | head -40


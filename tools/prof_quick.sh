#!/bin/bash
# rocprofv3 kernel stats of a short bench run; prints the per-kernel table (tools/summarize_profiles.py) for tag $1
TAG=${1:-quick}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/$TAG.log 2>&1
cd $R && python tools/summarize_profiles.py $TAG $(find gpurun_out/$TAG -name "*kernel_stats.csv") 8 | head -${2:-45}
rm -f profiles/${TAG}_kernel_stats.txt

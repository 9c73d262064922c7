#!/bin/bash
# BASELINE config C4 (128x128x64, b = 8): one-stream rocprofv3 kernel stats at the round's final code -> profiles/<tag>_c4_kernel_stats.txt
TAG=${1:-r05bn}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c4 -- python $R/bench.py --b 8 --dhw 128,128,64 --steps 4 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/${TAG}_c4.log 2>&1
cd $R
PROFILE_CMD="PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 python bench.py --b 8 --dhw 128,128,64 (BASELINE config C4, bf16): one stream, every kernel alone on the chip" python tools/summarize_profiles.py ${TAG}_c4 $(find gpurun_out/${TAG}_c4 -name "*kernel_stats.csv") 6 | head -16
cp profiles/${TAG}_c4_kernel_stats.txt gpurun_out/
python bench.py --b 8 --dhw 128,128,64 --steps 10 --warmup 4 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_c4_bench.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_c4_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['workload'])"
rm -rf gpurun_out/${TAG}_c4

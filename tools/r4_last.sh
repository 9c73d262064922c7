#!/bin/bash
# Last records of round 4: tools/r4_final.sh + the two-rank gloo run of bench.py's multi-rank path + 3D / 2D profiles.
TAG=${1:-r04w}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/r4_final.sh $TAG
PCRL_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > gpurun_out/${TAG}_bench_gloo2.json 2> gpurun_out/${TAG}_bench_gloo2.err
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench_gloo2.json').read().strip().splitlines()[-1]); print('gloo2', d['value'], d['ms_per_step'], json.dumps(d.get('distributed', {}).get('ab'))[:400])"
bash tools/profile_step.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1; head -3 gpurun_out/${TAG}_kernel_stats.txt; rm -rf gpurun_out/${TAG}
bash tools/profile_2d.sh ${TAG}_2d_c5 > gpurun_out/${TAG}_prof2d.log 2>&1; head -3 gpurun_out/${TAG}_2d_c5_kernel_stats.txt
grep -c "at::native" gpurun_out/${TAG}_kernel_stats.txt gpurun_out/${TAG}_2d_c5_kernel_stats.txt

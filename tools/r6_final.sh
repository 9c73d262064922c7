#!/bin/bash
# End-of-round-6 records on ONE box (outputs under gpurun_out/, copied into profiles/ by the author): GPU test log, driver-style bench line, rocprofv3 kernel
# stats (one-stream + three-stream) + PMC passes of C2, the same for C4 (128x128x64, b = 8), 2D C5 kernel stats + PMC, per-launch table, doubling ablation,
# counter table, phase map, board power over the step loop.      gpurun --timeout 3000 -- 'bash tools/r6_final.sh r06z'
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
[ "$SKIP_PYTEST" = 1 ] || python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1
[ "$SKIP_PYTEST" = 1 ] || grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_driver.json").read().strip().splitlines()[-1])
rf = d["roofline"]
print(d["value"], d["ms_per_step"], rf["kernel"], rf["frac"], rf.get("family", {}).get("frac"), rf.get("weighted_matrix_frac"), d["step_mfma_frac"], d["step_mfma_frac_executed"],
      {k: (v.get("value"), v.get("ms_per_step")) for k, v in d["secondary"].items()}, d["cpu_baseline"]["value"])
PY
bash tools/profile_step.sh ${TAG} > gpurun_out/${TAG}_profile_step.log 2>&1
tail -6 gpurun_out/${TAG}_profile_step.log
B_EXTRA="--b 8 --dhw 128,128,64" B_STEPS=4 PROFILE_LABEL="BASELINE config C4: b=8, 128x128x64, bf16" bash tools/profile_step.sh ${TAG}_c4 > gpurun_out/${TAG}_c4_profile_step.log 2>&1
tail -4 gpurun_out/${TAG}_c4_profile_step.log
bash tools/profile_2d.sh ${TAG}_2d_c5 > gpurun_out/${TAG}_2d_profile.log 2>&1
tail -4 gpurun_out/${TAG}_2d_profile.log
bash tools/step_launch_table.sh ${TAG} > gpurun_out/${TAG}_slt.log 2>&1
python tools/double_ablation.py --steps 10 --rounds 3 > gpurun_out/${TAG}_double_ablation.txt 2>&1
tail -20 gpurun_out/${TAG}_double_ablation.txt
bash tools/valu_table.sh ${TAG} 3d > gpurun_out/${TAG}_valu.log 2>&1
bash tools/phase_map.sh gpurun_out/${TAG}_phase_map.txt 250 > /dev/null 2>&1
bash tools/step_power.sh 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_step_power.txt
tail -3 gpurun_out/${TAG}_step_power.txt
rm -rf gpurun_out/${TAG} gpurun_out/${TAG}_c4 gpurun_out/${TAG}_2d_c5 gpurun_out/slt
ls gpurun_out | grep ${TAG} | head -60

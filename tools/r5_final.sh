#!/bin/bash
# End-of-round-5 records on ONE box: GPU test log, driver-style bench line, rocprofv3 kernel stats (one-stream + three-stream) + PMC passes, per-launch table,
# the doubling ablation (what each launch set costs inside the three-stream step).
TAG=${1:-r05bd}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("weighted_matrix_frac"), d["step_mfma_frac"], d["step_mfma_frac_executed"], {k: (v.get("value"), v.get("ms_per_step")) for k, v in d["secondary"].items()})
PY
bash tools/profile_step.sh ${TAG} > gpurun_out/${TAG}_profile_step.log 2>&1
tail -12 gpurun_out/${TAG}_profile_step.log
bash tools/step_launch_table.sh ${TAG} > gpurun_out/${TAG}_slt.log 2>&1
python tools/double_ablation.py --steps 10 --rounds 3 > gpurun_out/${TAG}_double_ablation.txt 2>&1
tail -16 gpurun_out/${TAG}_double_ablation.txt

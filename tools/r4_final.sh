#!/bin/bash
# End-of-round records (run on the GPU box): full GPU test log, the driver-style bench line, C4 one-stream kernel stats.
TAG=${1:-r04z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_driver.json", "gpurun_out/${TAG}_bench.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["step_mfma_frac"], d["step_mfma_frac_executed"], {k: (v.get("value"), v.get("ms_per_step")) for k, v in d["secondary"].items()})
PY
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c4 -- python $R/bench.py --b 8 --dhw 128,128,64 --steps 4 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/${TAG}_c4.log 2>&1
cd $R
PROFILE_CMD="PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 python bench.py --b 8 --dhw 128,128,64 (BASELINE config C4, bf16): one stream, every kernel alone on the chip" python tools/summarize_profiles.py ${TAG}_c4 $(find gpurun_out/${TAG}_c4 -name "*kernel_stats.csv") 6 | head -14
cp profiles/${TAG}_c4_kernel_stats.txt gpurun_out/
rm -rf gpurun_out/${TAG}_c4

#!/bin/bash
# Same-box A/B of two bench configurations given as env strings: tools/ab_bench.sh "<envA>" "<envB>" [rounds] [steps]
A="$1"; B="$2"; R=${3:-2}; S=${4:-15}
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  env $A python bench.py --steps $S --warmup 4 --no-cpu-baseline --no-alone > gpurun_out/ab/a_$i.json 2> gpurun_out/ab/a_$i.err
  env $B python bench.py --steps $S --warmup 4 --no-cpu-baseline --no-alone > gpurun_out/ab/b_$i.json 2> gpurun_out/ab/b_$i.err
done
python - <<PY
import json, glob
for n in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        d = json.load(open(n))
        print(n.split("/")[-1], d["value"], d["ms_per_step"], {k.split("(")[0]: (v["avg_ms"], v["tflops"]) for k, v in d["kernels"].items()})
    except Exception as e:
        print(n, "failed", e, open(n.replace(".json", ".err")).read()[-2000:])
PY

#!/bin/bash
# Same-box comparison of N bench configurations given as env strings (interleaved rounds):
#   tools/abn_bench.sh [-r rounds] [-s steps] "<envA>" "<envB>" ...      ("-" = no extra environment)
R=2; S=15
while getopts "r:s:" o; do case $o in r) R=$OPTARG;; s) S=$OPTARG;; esac; done
shift $((OPTIND-1))
mkdir -p gpurun_out/abn; rm -f gpurun_out/abn/*
for i in $(seq 1 $R); do
  k=0
  for E in "$@"; do
    [ "$E" = "-" ] && E=""
    env $E python bench.py --steps $S --warmup 6 --no-cpu-baseline --no-alone --no-secondary > gpurun_out/abn/c${k}_$i.json 2> gpurun_out/abn/c${k}_$i.err
    k=$((k+1))
  done
done
python - "$@" <<'PY'
import json, glob, sys
cfgs = sys.argv[1:]
for k, e in enumerate(cfgs):
    vals = []
    for n in sorted(glob.glob("gpurun_out/abn/c%d_*.json" % k)):
        try:
            d = json.load(open(n)); vals.append(d["ms_per_step"])
        except Exception as ex:
            print(n, "failed", ex, open(n.replace(".json", ".err")).read()[-1500:])
    print("config %d [%s]: ms/step %s  mean %.3f" % (k, e, vals, sum(vals) / max(len(vals), 1)))
PY

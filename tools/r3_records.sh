#!/bin/bash
# Round-3 side records kept under profiles/ (run on the GPU box through gpurun): tools/r3_records.sh <tag>
#   (1) two ranks on ONE GPU over gloo through the product's multi-rank path (the N > 1 code path with real kernels; RCCL needs N GPUs),
#   (2) python main.py --data synthetic crops/s per epoch for epochs 0..12: the device-side divergence guard is live from epoch 11,
#       with PCRL_GUARD_SYNC=1 (the reference's host-side decision, a forward -> backward synchronisation) beside it,
#   (3) a rocprofv3 --kernel-trace --marker-trace run with PCRL_TRACE_RANGES=1: the roctx ranges around forward / backward / optimizer.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
PCRL_DIST_BACKEND=gloo PCRL_BIND_VERBOSE=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-alone --no-secondary > $O/bench_gloo2.json 2> $O/bench_gloo2.err
for mode in 0 1; do
  PCRL_GUARD_SYNC=$mode python main.py --data synthetic --d 3 --b 32 --epochs 12 --steps_per_epoch 30 --gpus 0 --amp --output /tmp/out_$mode > $O/main_guard_sync$mode.log 2>&1
done
python - <<PY > $O/main_synthetic_epochs.txt
import re
for mode in (0, 1):
    rows = re.findall(r"epoch (\d+), total time ([0-9.]+)", open("$O/main_guard_sync%d.log" % mode).read())
    print("PCRL_GUARD_SYNC=%d (%s): crops/s per epoch (b = 32, 30 steps per epoch, bf16; epoch 0 includes start-up):" % (mode, "reference-style host decision" if mode else "device-side guard, default"))
    print("  " + "  ".join("e%s %.0f" % (e, 32 * 30 / float(t)) for e, t in rows))
PY
cat $O/main_synthetic_epochs.txt
cd /tmp && export TMPDIR=/tmp
PCRL_TRACE_RANGES=1 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $O/markers -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $O/markers.log 2>&1
cd $R
M=$(find $O/markers -name "*marker_api_trace.csv" | head -1)
if [ -n "$M" ]; then
  python - "$M" <<PY > $O/roctx_ranges.txt
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Function") or r.get("Name") or str(r)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# roctx ranges seen by rocprofv3 --marker-trace (PCRL_TRACE_RANGES=1; host-side enqueue spans): name, count, mean ms")
for k, (n, t) in agg.items():
    print(f"{k:60s} {n:5d} {t / n / 1e6:9.3f}")
PY
  cat $O/roctx_ranges.txt
else
  echo "no marker trace produced"; tail -5 $O/markers.log
fi
rm -rf $O/markers

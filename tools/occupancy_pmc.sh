#!/bin/bash
# Achieved occupancy per kernel of a one-stream step: waves per SIMD = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
# (SQ_WAVE_CYCLES counts quad-cycles).   gpurun -- 'bash tools/occupancy_pmc.sh <tag> 3d|2d'
TAG=$1; WHAT=${2:-3d}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 PCRL_VIEW_STREAMS_2D=0
if [ "$WHAT" = "2d" ]; then CMD="python $R/tools/bench_2d.py --steps 2 --warmup 2 --no-roofline"; else CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone --no-secondary"; fi
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$TAG/a -- $CMD > $R/gpurun_out/$TAG.log 2>&1
cd $R
python - $TAG $WHAT <<'PY'
import collections, csv, glob, sys
sys.path.insert(0, "tools")
from summarize_profiles import short
tag, what = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
f = glob.glob(f"gpurun_out/{tag}/a/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])[:46]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
rows = []
for k, c in agg.items():
    dur = c["GRBM_GUI_ACTIVE"] / 8.0
    if dur <= 0:
        continue
    rows.append((dur, k, len(n[k]), 4 * c["SQ_WAVE_CYCLES"] / (dur * 1024), c["SQ_WAVES"] / len(n[k]), c["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 1024)))
tot = sum(r[0] for r in rows)
out = [f"# {tag}: achieved occupancy per kernel over one-stream {what} steps (rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE): waves/SIMD = 4 * SQ_WAVE_CYCLES / (cycles * 1024)",
       "%-46s %8s %7s %11s %12s %9s" % ("kernel", "launches", "% time", "waves/SIMD", "waves/launch", "MFMA busy")]
for dur, k, ln, occ, wv, mf in sorted(rows, reverse=True)[:45]:
    out.append("%-46s %8d %7.2f %11.2f %12.0f %8.1f%%" % (k, ln, 100 * dur / tot, occ, wv, 100 * mf))
open(f"gpurun_out/{tag}_occupancy.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/$TAG

#!/bin/bash
# One-stream rocprofv3 kernel stats + per-dispatch kernel trace of the 2D step probe (every kernel alone on the chip).
#   gpurun -- 'tools/prof_2d_trace.sh <tag> [bench_2d args]'  ->  gpurun_out/<tag>_kernel_stats.txt, gpurun_out/<tag>_trace.csv (last step only)
TAG=${1:-prof2d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
shift
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 PCRL_VIEW_STREAMS_2D=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/tools/bench_2d.py --steps 3 --warmup 2 --no-roofline "$@" > $R/gpurun_out/$TAG.log 2>&1
cd $R
export PROFILE_CMD="PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS_2D=0 python tools/bench_2d.py --steps 3 --warmup 2 $* (C5 per-GPU workload: 512x512, b=64, bf16): one stream, every kernel alone on the chip"
python tools/summarize_profiles.py $TAG $(find gpurun_out/$TAG -name "*kernel_stats.csv") 5 | head -60
cp profiles/${TAG}_kernel_stats.txt gpurun_out/ 2>/dev/null
python - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 5
last = rows[-n:]
with open(f"gpurun_out/{tag}_trace.csv", "w") as o:
    o.write("idx,us,grid,wg,lds,name\n")
    for i, r in enumerate(last):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:110].replace(",", ";")
        o.write(f"{i},{us:.1f},{r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}x{r.get('Grid_Size_Z','')},{r.get('Workgroup_Size_X','')},{r.get('LDS_Block_Size','')},{nm}\n")
print("trace rows per step:", n)
PY
rm -rf gpurun_out/$TAG
tail -2 gpurun_out/$TAG.log

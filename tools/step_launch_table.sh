# Every launch of ONE one-stream step in issue order, with grid and duration, plus a per-(kernel, grid) summary:
#   gpurun -- 'bash tools/step_launch_table.sh [tag] [bench args...]'     -> gpurun_out/<tag>_launches.txt, <tag>_launch_groups.txt
#   STEP2D=1: the 2D step (tools/bench_2d.py) instead of bench.py
TAG=${1:-launches}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
rm -rf $R/gpurun_out/slt
if [ "$STEP2D" = 1 ]; then
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/slt -- python $R/tools/bench_2d.py --steps 4 --warmup 2 "$@" > $R/gpurun_out/slt.log 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/slt -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alone --no-secondary "$@" > $R/gpurun_out/slt.log 2>&1
fi
python - $R/gpurun_out/${TAG} $(find $R/gpurun_out/slt -name "*kernel_trace.csv") <<'PY'
import csv, sys, re, collections
out, f = sys.argv[1:3]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:64]
# a step = the launches between two sgd4 kernels; take the last complete one
idx = [i for i, r in enumerate(rows) if "sgd4" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
with open(out + "_launches.txt", "w") as o:
    o.write("# one-stream step, %d launches, issue order: start(us) dur(us) grid wg lds kernel\n" % len(step))
    for r in step:
        o.write("%9.1f %8.1f  %7d x %4d x %3d  wg %4d lds %6s  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]), r.get("LDS_Block_Size", "?"), short(r["Kernel_Name"])))
g = collections.OrderedDict()
for r in step:
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    g.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
with open(out + "_launch_groups.txt", "w") as o:
    o.write("# one-stream step: %.2f ms of kernel time in %d launches; per (kernel, grid): launches, avg us, total us\n" % (tot / 1e3, len(step)))
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        o.write("%-64s %7d x %4d x %3d  n %3d  avg %8.1f  tot %8.1f\n" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), sum(v)))
print(open(out + "_launch_groups.txt").read()[:6000])
PY
rm -rf $R/gpurun_out/slt

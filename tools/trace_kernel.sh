# per-launch durations and grids of the kernels whose name contains $1, inside a one-stream 3D step:  gpurun -- 'bash tools/trace_kernel.sh upc_class_sums'
PAT=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
rm -rf $R/gpurun_out/trk
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trk -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/trk.log 2>&1
python - "$PAT" $(find $R/gpurun_out/trk -name "*kernel_trace.csv") <<'PY'
import csv, sys
pat, f = sys.argv[1:3]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 5
for r in rows[-n:]:
    print("%-40s grid %8s x %4s x %3s  %8.1f us" % (r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $R/gpurun_out/trk

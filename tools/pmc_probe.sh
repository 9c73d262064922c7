# PMC stall counters of one probe command, per kernel:   gpurun -- 'bash tools/pmc_probe.sh <tag> <command...>'
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/$TAG/a -- "$@" > $R/gpurun_out/$TAG.a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/gpurun_out/$TAG/b -- "$@" > $R/gpurun_out/$TAG.b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/c -- "$@" > $R/gpurun_out/$TAG.c.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$TAG/d -- "$@" > $R/gpurun_out/$TAG.d.log 2>&1
cd $R
python - $TAG <<'PY'
import collections, csv, glob, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(f"gpurun_out/{tag}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in agg:
    print("==", k)
    for c in sorted(agg[k]):
        print("   %-34s %16.0f per launch  (%d launches)" % (c, agg[k][c] / max(len(n[k][c]), 1), len(n[k][c])))
PY
rm -rf gpurun_out/$TAG

# Per kernel of a one-stream step: vector / scalar / LDS instructions per wave, and how busy the SIMDs' vector issue is -- the table that pointed at the first
# layer's address arithmetic (profiles/r05z_c1_addr_ab.txt).   gpurun -- 'bash tools/valu_table.sh <tag> [3d|2d]'  -> gpurun_out/<tag>_valu_table.txt
TAG=${1:-valu}; WHAT=${2:-3d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 PCRL_VIEW_STREAMS_2D=0
if [ "$WHAT" = "2d" ]; then CMD="python $R/tools/bench_2d.py --steps 2 --warmup 2 --no-roofline"; else CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone --no-secondary"; fi
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$TAG/a -- $CMD > $R/gpurun_out/$TAG.a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/$TAG/b -- $CMD > $R/gpurun_out/$TAG.b.log 2>&1
cd $R
python - $TAG $WHAT <<'PY'
import collections, csv, glob, sys
sys.path.insert(0, "tools")
from summarize_profiles import short
tag, what = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(f"gpurun_out/{tag}/*/**/*counter_collection.csv", recursive=True):
    p = f.split(f"gpurun_out/{tag}/")[1][0]
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])[:44]
        agg[k][p + r["Counter_Name"]] += float(r["Counter_Value"]); n[k][p].add(r["Dispatch_Id"])
rows = []
for k, c in agg.items():
    cyc = c["aGRBM_GUI_ACTIVE"] / 8.0
    if cyc <= 0 or c["bSQ_WAVES"] <= 0:
        continue
    w = c["bSQ_WAVES"]
    rows.append((cyc, k, len(n[k]["a"]), 4 * c["aSQ_WAVE_CYCLES"] / (cyc * 1024), 4 * c["aSQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4 * c["aSQ_ACTIVE_INST_SCA"] / (cyc * 1024),
                 4 * c["aSQ_ACTIVE_INST_LDS"] / (cyc * 1024), c["aSQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                 c["bSQ_INSTS_VALU"] / w, c["bSQ_INSTS_SALU"] / w, c["bSQ_INSTS_LDS"] / w, (c["bSQ_INSTS_VMEM_RD"] + c["bSQ_INSTS_VMEM_WR"]) / w, c["bSQ_INSTS_MFMA"] / w))
tot = sum(r[0] for r in rows)
out = [f"# {tag}: one-stream {what} step, per kernel: share of GPU time, achieved waves per SIMD, fraction of SIMD cycles a vector / scalar / LDS instruction of the kernel issues,",
       "# MFMA pipe busy, and instructions per WAVE (vector incl. MFMA, scalar, LDS, global, MFMA).  4 x SQ_ACTIVE_INST_* / (cycles x 1024 SIMDs); cycles = GRBM_GUI_ACTIVE / 8",
       "%-44s %5s %6s %6s %6s %6s %6s %6s | %8s %8s %7s %7s %7s" % ("kernel", "n", "% time", "w/SIMD", "VALU", "SALU", "LDS", "MFMA", "valu/w", "salu/w", "lds/w", "vmem/w", "mfma/w")]
for r in sorted(rows, reverse=True)[:48]:
    out.append("%-44s %5d %6.2f %6.2f %5.1f%% %5.1f%% %5.1f%% %5.1f%% | %8.0f %8.0f %7.0f %7.0f %7.0f" % (r[1], r[2], 100 * r[0] / tot, r[3], 100 * r[4], 100 * r[5], 100 * r[6], 100 * r[7], r[8], r[9], r[10], r[11], r[12]))
open(f"gpurun_out/{tag}_valu_table.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/$TAG

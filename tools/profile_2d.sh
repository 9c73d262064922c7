#!/bin/bash
# rocprofv3 evidence of the 2D step (C5 per-GPU workload) kept under profiles/: one-stream kernel stats + the three PMC passes
# (FETCH_SIZE | WRITE_SIZE | SQ), collected separately as MI355X_MICROARCH.md prescribes.   gpurun -- 'tools/profile_2d.sh <tag>'
TAG=${1:-prof2d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0 PCRL_VIEW_STREAMS_2D=0
B="python $R/tools/bench_2d.py --steps 3 --warmup 2 --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- $B > $R/gpurun_out/$TAG.stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/fetch -- $B > $R/gpurun_out/$TAG.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/write -- $B > $R/gpurun_out/$TAG.write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/sq -- $B > $R/gpurun_out/$TAG.sq.log 2>&1
cd $R
python - "$TAG" <<'PY'
import collections, csv, glob, json, sys
sys.path.insert(0, "tools")
from summarize_profiles import short
tag = sys.argv[1]
STEPS = 5
def k2(name):
    s = short(name)
    for key in ("conv2d_narrow_kernel", "wgrad2d_narrow_kernel", "stem7_fwd_kernel", "stem7_wgrad_kernel", "stem7_wgrad_reduce_kernel", "bn_add_relu_kernel", "relu_mask_sum_kernel",
                "bn_relu_maxpool2d_kernel", "maxpool2d_bwd_sum_kernel", "nearest2_bwd_kernel", "conv1x1_small_bwd_kernel", "mse2d_bwd_pad_kernel", "head_bwd_stage_kernel"):
        if key in name:
            return key
    return s
stats = glob.glob(f"gpurun_out/{tag}/stats/**/*kernel_stats.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(stats)):
    a = agg.setdefault(k2(r["Name"]), [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
def pm(d):
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    f = glob.glob(f"gpurun_out/{tag}/{d}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = k2(r["Kernel_Name"]); a[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return a, n
fa, fn = pm("fetch"); wa, wn = pm("write"); sa, sn = pm("sq")
lines = [f"# {tag}: rocprofv3, PCRL_WGRAD_STREAM=0 PCRL_VIEW_STREAMS_2D=0 python tools/bench_2d.py --steps 3 --warmup 2 (C5 per-GPU workload: 512x512, b=64, bf16): one stream, every kernel alone on the chip",
         f"# total kernel time {tot / 1e6 / STEPS:.2f} ms/step; PMC passes collected separately (FETCH_SIZE | WRITE_SIZE | SQ); HBM bytes = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> x1024",
         "%-34s %10s %9s %9s %6s | %10s %8s %9s" % ("kernel", "calls/step", "ms/step", "avg_us", "pct", "MB/launch", "TB/s", "MFMA busy")]
out = {}
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    row = "%-34s %10.1f %9.3f %9.1f %6.2f" % (k[:34], c / STEPS, t / 1e6 / STEPS, t / 1e3 / c, 100 * t / tot)
    if k in fn and k in wn and 100 * t / tot >= 0.4:
        mb = (2 * fa[k]["FETCH_SIZE"] / len(fn[k]) + wa[k]["WRITE_SIZE"] / len(wn[k])) * 1024 / 1e6
        tbs = mb * 1e6 / (t / c * 1e-9) / 1e12
        mf = sa[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / max(sa[k]["GRBM_GUI_ACTIVE"] * 128, 1) if k in sn else 0.0
        row += " | %10.1f %8.2f %8.1f%%" % (mb, tbs, 100 * mf)
        out[k] = {"hbm_bytes_per_launch": mb * 1e6, "hbm_TBps": tbs, "mfma_busy": mf, "avg_us": t / 1e3 / c, "calls_per_step": c / STEPS}
    lines.append(row)
hb = sum(v["hbm_bytes_per_launch"] * v["calls_per_step"] for v in out.values())
lines.insert(2, f"# HBM bytes of the kernels listed with counters: {hb / 1e9:.1f} GB per step = {hb / (tot / STEPS * 1e-9) / 1e12:.2f} TB/s averaged over the one-stream kernel time")
open(f"profiles/{tag}_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print("\n".join(lines[:45]))
PY
cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/$TAG

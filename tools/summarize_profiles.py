#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small text summaries kept under profiles/.

    python tools/summarize_profiles.py <tag> <kernel_stats.csv> <steps_in_trace> [<pmc_fetch.csv> <pmc_write.csv> <pmc_sq.csv>]
"""
import collections
import csv
import json
import sys


def short(name):
    for key in ("brick16_conv_kernel", "brick_conv_kernel", "wgrad_brick_upc2_kernel", "wgrad_brick27_kernel", "wgrad_brick_kernel", "wgrad_upc8_kernel", "wgrad_reduce_upc_kernel", "wgrad_reduce_kernel", "bn_bwd_reduce_pool_kernel",
                "bn_bwd_apply_pool_kernel", "bn_apply_pool_kernel", "bn_apply_gap_kernel", "bn_bwd_reduce_kernel", "bn_bwd_apply_rc_kernel",
                "bn_apply_rc_kernel", "igemm_kernel", "wgrad_kernel", "coltile_sum_kernel", "sgemm_small_kernel", "shift_sum27_kernel",
                "c1_fwd_kernel", "maxpool_bwd_kernel", "maxpool_fwd_kernel", "im2col27_kernel", "gap_bwd_kernel", "to1_dgrad_kernel",
                "to1_fwd_kernel", "bn_finalize_kernel", "bn_bwd_finalize_kernel", "coltile_finish_kernel", "sgd_kernel", "tri_fwd_kernel",
                "tri_bwd_kernel", "cosine_fwd_kernel", "cosine_bwd_kernel", "bn1d_fwd_kernel", "bn1d_bwd_kernel", "pack_conv3_kernel"):
        if key in name:
            if key == "brick16_conv_kernel":      # template arguments <BN, MODE>: MODE 1 / 2 = the composed up-conv's forward / data gradient
                import re
                m = re.search(r"brick16_conv_kernel(?:ILi\d+ELi(\d)E|<\d+, (\d)[,>])", name)     # <BN, MODE, PERM>
                mode = (m.group(1) or m.group(2)) if m else "0"
                if re.search(r"brick16_conv_kernel(?:ILi\d+ELi\d+ELi\d+ELi\d+ELb1E|<[^>]*true>)", name):
                    return key + "<dgrad+bn_reduce>"       # BNR instantiations (conv_brick16_bnr.hip): the data gradient with the BatchNorm backward's first pass
                return key + {"1": "<upconv_fwd>", "2": "<upconv_dgrad>"}.get(mode, "")
            if key == "igemm_kernel":
                import re
                m = re.search(r"igemm_kernelI(DF16b|f)Li(\d+)ELi(\d)ELb(\d)", name)
                if m:
                    return "igemm_kernel<%s,BN=%s,geom=%s%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2),
                                                                {"0": "conv3", "1": "convT_fwd", "2": "convT_dgrad", "3": "upconv_fwd", "4": "upconv_dgrad"}[m.group(3)],
                                                                ",planes" if m.group(4) == "1" else "")
            return key
    import re
    m = re.search(r"conv2d_kernelI(DF16b|f)Li(\d+)ELi(\d)E", name)
    if m:
        return "conv2d_kernel<%s,BN=%s,%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2), "dgrad" if m.group(3) == "1" else "fwd")
    m = re.search(r"_ZN12_GLOBAL__N_1\d+(\w+?_kernel)I", name)
    if m:
        return m.group(1)
    return name.replace("void ", "").replace("(anonymous namespace)::", "")[:70]


def main():
    tag, stats, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rows = list(csv.DictReader(open(stats)))
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    import os
    what = os.environ.get("PROFILE_CMD", "python bench.py (b=32, 64x64x32, bf16)")
    lines = [f"# {tag}: rocprofv3 --kernel-trace --stats -- {what}; {steps} steps in the trace",
             f"# total kernel time {tot / 1e6 / steps:.2f} ms/step",
             f"{'kernel':52s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>9s} {'pct':>6s}"]
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:52s} {c / steps:10.1f} {t / 1e6 / steps:9.3f} {t / 1e3 / c:9.1f} {100 * t / tot:6.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 4:
        def pm(path):
            a = collections.defaultdict(lambda: collections.defaultdict(float))
            d = collections.defaultdict(set)
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                a[k][r["Counter_Name"]] += float(r["Counter_Value"])
                d[k].add(r["Dispatch_Id"])
            return a, d
        fa, fd = pm(sys.argv[4])
        wa, wd = pm(sys.argv[5])
        ma, md = pm(sys.argv[6])
        out += ("\n# PMC (separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters), per launch averages.\n"
                "# FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced read stream); KB -> bytes x1024.\n"
                "# SQ counters are sampled on ONE XCD (SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE ~ 30 of its 32 CUs): MFMA busy fraction =\n"
                "# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 32 CUs * 4 SIMDs).\n")
        traffic = {}
        mfma_kernels = ["brick16_conv_kernel", "brick16_conv_kernel<dgrad+bn_reduce>", "brick16_conv_kernel<upconv_fwd>", "brick16_conv_kernel<upconv_dgrad>", "brick_conv_kernel", "wgrad_brick_kernel", "wgrad_brick27_kernel", "wgrad_brick_upc2_kernel", "wgrad_upc8_kernel"] + sorted(k for k in fd if k.startswith("igemm_kernel<") and "upconv" in k)
        for k in mfma_kernels:
            if k not in fd or k not in wd or k not in md:
                continue
            nf, nw, nm = len(fd[k]), len(wd[k]), len(md[k])
            f_mb = 2 * fa[k]["FETCH_SIZE"] / nf * 1024 / 1e6
            w_mb = wa[k]["WRITE_SIZE"] / nw * 1024 / 1e6
            m = ma[k]
            util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128)
            conf = m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)
            out += (f"{k:22s} launches {nf:4d}  HBM fetch {f_mb:8.1f} MB  write {w_mb:8.1f} MB  total {f_mb + w_mb:8.1f} MB/launch | "
                    f"MFMA busy {100 * util:5.1f} %  CU busy {m['SQ_BUSY_CU_CYCLES'] / m['GRBM_GUI_ACTIVE'] / 32 * 100:5.1f} %  "
                    f"LDS bank-conflict cycles / LDS active {100 * conf:4.1f} %\n")
            traffic[k] = {"hbm_bytes_per_launch": (f_mb + w_mb) * 1e6, "mfma_busy": util}
        # HBM-bound kernels: measured bytes per launch / average launch duration (kernel stats of the same command) -> TB/s
        dur = {k: t / c for k, (c, t) in agg.items()}      # ns per launch
        hb = [k for k in ("bn_bwd_apply_rc_kernel", "bn_bwd_reduce_kernel", "bn_apply_rc_kernel", "bn_apply_gap_kernel", "bn_apply_pool_kernel",
                          "bn_bwd_apply_pool_kernel", "bn_bwd_reduce_pool_kernel", "maxpool_bwd_kernel", "maxpool_fwd_kernel",
                          "gap_bwd_kernel", "coltile_sum_kernel") if k in fd and k in wd and k in dur]
        if hb:
            out += "\n# HBM-bound kernels: PMC bytes per launch (FETCH x2 + WRITE) / average launch duration from the stats pass\n"
        for k in hb:
            f_mb = 2 * fa[k]["FETCH_SIZE"] / len(fd[k]) * 1024 / 1e6
            w_mb = wa[k]["WRITE_SIZE"] / len(wd[k]) * 1024 / 1e6
            tbs = (f_mb + w_mb) * 1e6 / (dur[k] * 1e-9) / 1e12
            out += f"{k:24s} launches {len(fd[k]):4d}  fetch {f_mb:8.1f} MB  write {w_mb:8.1f} MB  avg {dur[k] / 1e3:7.1f} us  -> {tbs:5.2f} TB/s of ~8\n"
            traffic[k] = {"hbm_bytes_per_launch": (f_mb + w_mb) * 1e6, "hbm_TBps": tbs}
        json.dump(traffic, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
    open(f"profiles/{tag}_kernel_stats.txt", "w").write(out)
    print(out)


if __name__ == "__main__":
    main()

#!/bin/bash
# Stall attribution of the wide-brick convolution kernel's three modes (plain 3x3x3, composed up-conv forward, composed data gradient):
# SQ counter passes over ONE-stream steps of bench.py (every kernel alone on the chip), one table per launch class.
#   gpurun -- 'tools/stall_pmc.sh <tag>'  ->  gpurun_out/<tag>_stall_table.txt
TAG=${1:-stall}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone --no-secondary"
rocprofv3 -L > $R/gpurun_out/$TAG.counters.txt 2>&1
avail() { for c in "$@"; do grep -qw "$c" $R/gpurun_out/$TAG.counters.txt && echo -n "$c "; done; }
P1=$(avail SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES)
P2=$(avail SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16)
P3=$(avail SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)
P4=$(avail GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL)
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  [ -z "$P" ] && continue
  echo "pass $i: $P"
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/$TAG/p$i -- $B > $R/gpurun_out/$TAG.p$i.log 2>&1
  f=$(find $R/gpurun_out/$TAG/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $R/gpurun_out/$TAG/p$i.csv; grep "brick16_conv_kernel\|wgrad_brick_upc2_kernel\|wgrad_brick_kernel" $f >> $R/gpurun_out/$TAG/p$i.csv; fi
  rm -rf $R/gpurun_out/$TAG/p$i
done
cd $R
python - "$TAG" <<'PY'
import collections, csv, glob, sys
sys.path.insert(0, "tools")
from summarize_profiles import short
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))     # pass -> kernel -> counter
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(f"gpurun_out/{tag}/p*.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[f][k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[f][k].add(r["Dispatch_Id"])
kernels = [k for k in ("brick16_conv_kernel", "brick16_conv_kernel<upconv_fwd>", "brick16_conv_kernel<upconv_dgrad>", "wgrad_brick_kernel", "wgrad_brick_upc2_kernel")
           if any(k in agg[f] for f in agg)]
lines = ["# %s: SQ counters per launch class, ONE-stream steps of bench.py (C2, bf16); sums over the sampled XCD's waves, divided by the launches" % tag,
         "# quad-cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) also as a fraction of SQ_WAVE_CYCLES of the SAME pass; WAIT_ANY = wave parked at s_waitcnt / barrier,",
         "# WAIT_INST_ANY = issue stall (MFMA dependency / pipe busy), WAIT_INST_LDS = its LDS-issue part, ACTIVE_INST_* = cycles an instruction of that class issues"]
lines.append("%-34s" % "counter" + "".join("%24s" % k.replace("brick16_conv_kernel", "b16").replace("_kernel", "") for k in kernels))
for f in sorted(agg):
    lines.append("# pass " + f.split("/")[-1])
    for c in sorted({c for k in kernels for c in agg[f][k]}):
        row = "%-34s" % c
        for k in kernels:
            v = agg[f][k].get(c)
            if v is None:
                row += "%24s" % "-"
                continue
            n = max(len(disp[f][k]), 1)
            wc = agg[f][k].get("SQ_WAVE_CYCLES")
            frac = (" (%4.1f%%)" % (100 * v / wc)) if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE_INST", "SQ_INST_CYCLES", "SQ_BUSY_CYCLES")) else ""
            row += "%24s" % ("%.4g%s" % (v / n, frac))
        lines.append(row)
open(f"gpurun_out/{tag}_stall_table.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY

#!/usr/bin/env python3
"""Where the wall-clock of an overlapped (multi-stream) step goes, from a rocprofv3 --kernel-trace CSV.

    python tools/timeline.py <kernel_trace.csv> [steps_in_trace=8] [skip_steps=3]

Steps are delimited by the SGD kernel (one launch per step).  For the steps after `skip_steps` it prints, per step on average:
wall time, time with no kernel in flight, time with only HBM/latency-class kernels in flight, time with >= 1 MFMA-class kernel in
flight, the busy time of every queue, and the kernels that run ALONE (nothing else in flight) most -- the serial stretches that
overlap does not hide."""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_profiles import short  # noqa: E402

MFMA = ("brick16_conv_kernel", "brick_conv_kernel", "wgrad_brick", "igemm_kernel", "wgrad_kernel", "wgrad_upc8", "to1_brick", "c1_brick",
        "upc_gemm", "brick8_", "upcb_")


def is_mfma(k):
    return any(m in k for m in MFMA)


def main():
    path = sys.argv[1]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    rows.sort()
    sgd = [i for i, r in enumerate(rows) if r[2].startswith("sgd")]
    if len(sgd) < skip + 2:
        sys.exit("not enough steps in the trace")
    lo, hi = sgd[skip], sgd[-1]
    nsteps = len(sgd) - 1 - skip
    t0, t1 = rows[lo][1], rows[hi][1]
    sel = [r for r in rows[lo + 1: hi + 1]]
    # sweep
    ev = []
    for s, e, k, q in sel:
        s, e = max(s, t0), min(e, t1)
        if e <= s:
            continue
        ev.append((s, 1, k, q))
        ev.append((e, -1, k, q))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = collections.Counter()
    qbusy = collections.Counter()
    qact = collections.Counter()
    alone = collections.Counter()
    idle = hbm_only = mfma_any = mfma2 = 0
    prev = t0
    for t, d, k, q in ev:
        dt = t - prev
        if dt > 0:
            n = sum(active.values())
            if n == 0:
                idle += dt
            else:
                nm = sum(v for kk, v in active.items() if is_mfma(kk))
                if nm == 0:
                    hbm_only += dt
                else:
                    mfma_any += dt
                    if nm >= 2:
                        mfma2 += dt
                if n == 1:
                    alone[next(kk for kk, v in active.items() if v > 0)] += dt
            for qq, v in qact.items():
                if v > 0:
                    qbusy[qq] += dt
        active[k] += d
        qact[q] += d
        if active[k] == 0:
            del active[k]
        prev = t
    wall = (t1 - t0) / nsteps / 1e6
    f = lambda x: x / nsteps / 1e6
    print(f"steps analysed {nsteps}; wall {wall:.2f} ms/step")
    print(f"  no kernel in flight        {f(idle):7.2f} ms")
    print(f"  only non-MFMA kernels      {f(hbm_only):7.2f} ms")
    print(f"  >= 1 MFMA-class kernel     {f(mfma_any):7.2f} ms   (>= 2 of them: {f(mfma2):.2f})")
    for q, v in sorted(qbusy.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q}: busy {f(v):7.2f} ms")
    ksum = collections.Counter()
    for s, e, k, q in sel:
        ksum[k] += e - s
    print("  kernels running ALONE (no other kernel in flight), ms/step   [their total kernel time]:")
    for k, v in alone.most_common(28):
        print(f"    {k:52s} {f(v):7.3f}   [{f(ksum[k]):7.3f}]")
    print(f"    total alone {f(sum(alone.values())):.2f} ms; of it MFMA-class {f(sum(v for k, v in alone.items() if is_mfma(k))):.2f}")


if __name__ == "__main__":
    main()

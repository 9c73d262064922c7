#!/usr/bin/env python3
"""WHERE in the three-stream step no matrix kernel is in flight (the windows that leave the 1 400 W cap unused: tools/step_power.sh reads ~1 260 W
over the step).  From a rocprofv3 --kernel-trace CSV of bench.py: the last complete step (between two SGD launches) in bins of BIN us; per bin the
share of time with >= 1 MFMA-class kernel in flight, with only other kernels, with nothing, and per queue the kernel that held most of the bin.

    python tools/phase_map.py <kernel_trace.csv> [bin_us=250]"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_profiles import short  # noqa: E402
from timeline import is_mfma  # noqa: E402


def main():
    path, bin_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    rows.sort()
    sgd = [i for i, r in enumerate(rows) if r[2].startswith("sgd")]
    lo, hi = sgd[-2], sgd[-1]
    t0, t1 = rows[lo][1], rows[hi][1]
    sel = [r for r in rows[lo + 1: hi + 1]]
    queues = sorted({r[3] for r in sel})
    nb = int((t1 - t0) / 1e3 / bin_us) + 1
    print(f"# one step: {(t1 - t0) / 1e6:.2f} ms wall (profiled), {len(sel)} launches, queues {queues}; bins of {bin_us:.0f} us")
    print("# t(ms)  mfma%  other-only%  idle%   " + "   ".join(f"queue {q}: kernel holding most of the bin" for q in queues))
    tot = collections.Counter()
    windows = []
    for b in range(nb):
        a, e = t0 + b * bin_us * 1e3, min(t0 + (b + 1) * bin_us * 1e3, t1)
        if e <= a:
            break
        # sweep inside the bin
        ev = []
        perq = {q: collections.Counter() for q in queues}
        for s, f, k, q in sel:
            s2, f2 = max(s, a), min(f, e)
            if f2 > s2:
                ev.append((s2, 1, is_mfma(k)))
                ev.append((f2, -1, is_mfma(k)))
                perq[q][k] += f2 - s2
        ev.sort()
        nm = no = 0
        prev = a
        tm = to = ti = 0.0
        for t, d, m in ev:
            dt = t - prev
            if dt > 0:
                if nm > 0:
                    tm += dt
                elif no > 0:
                    to += dt
                else:
                    ti += dt
            prev = t
            if m:
                nm += d
            else:
                no += d
        ti += e - prev
        w = e - a
        tot["m"] += tm; tot["o"] += to; tot["i"] += ti
        cells = []
        for q in queues:
            if perq[q]:
                k, v = perq[q].most_common(1)[0]
                cells.append(f"{k[:34]:34s} {100 * sum(perq[q].values()) / w:3.0f}%")
            else:
                cells.append(" " * 39)
        flag = " <<" if (to + ti) / w > 0.5 else ""
        print(f"{(a - t0) / 1e6:6.2f}  {100 * tm / w:5.0f}  {100 * to / w:10.0f}  {100 * ti / w:5.0f}   " + " | ".join(cells) + flag)
    T = (t1 - t0) / 1e6
    print(f"# totals: >= 1 MFMA-class kernel {tot['m'] / 1e6:.2f} ms, only others {tot['o'] / 1e6:.2f} ms, nothing {tot['i'] / 1e6:.2f} ms of {T:.2f} ms")


if __name__ == "__main__":
    main()

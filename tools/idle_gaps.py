import csv, sys, collections
sys.path.insert(0, "tools")
from summarize_profiles import short
def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    rows.sort()
    return rows
def gaps(path, skip=3):
    rows = load(path)
    sgd = [i for i, r in enumerate(rows) if r[2].startswith("sgd")]
    lo, hi = sgd[skip], sgd[skip + 1]
    sel = rows[lo:hi + 1]
    # idle intervals
    cur_end = sel[0][1]; out = []
    last = sel[0]
    for r in sel[1:]:
        if r[0] > cur_end:
            out.append((r[0] - cur_end, last[2], r[2], (cur_end - sel[0][1]) / 1e6))
        if r[1] > cur_end:
            cur_end = r[1]; last = r
    out.sort(reverse=True)
    print(path, "step wall %.2f ms, idle total %.2f ms in %d gaps" % ((sel[-1][1] - sel[0][1]) / 1e6, sum(g[0] for g in out) / 1e6, len(out)))
    for g in out[:14]:
        print("   gap %6.1f us at t=%6.2f ms  after %-40s before %s" % (g[0] / 1e3, g[3], g[1][:40], g[2][:40]))
if not (len(sys.argv) > 2 and sys.argv[1] == "--start"):
    for p in sys.argv[1:]:
        gaps(p)


def step_start(path, skip, n=40):
    """The first n kernels after the skip-th SGD launch: start / end relative to the SGD kernel's end (us), queue, name."""
    rows = load(path)
    sgd = [i for i, r in enumerate(rows) if r[2].startswith("sgd")]
    i0 = sgd[skip]
    t0 = rows[i0][1]
    print("step start after SGD #%d (t = 0 at its end):" % skip)
    for r in rows[i0 + 1:i0 + 1 + n]:
        print("   %8.1f .. %8.1f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2][:60]))


if len(sys.argv) > 2 and sys.argv[1] == "--start":
    step_start(sys.argv[3], int(sys.argv[2]))

"""Per-kernel time inside the graph-replayed steps of a rocprofv3 kernel trace of bench.py (the timed region), with the
time each kernel runs ALONE on the device (exclusive) separated from time shared with a concurrent kernel."""
import csv, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r["Kernel_Name"]]
# layout of bench.py --steps K: warm-up (eager + capture) ... K replayed steps [timed], then the roofline legs: re-capture
# (3 eager warm-up steps) + K replayed steps with timestamps + K eager steps.  argv[4] = rasterizer forwards after the timed
# region (default: the two legs; 0 with bench.py --no-roofline-legs)
tail = int(sys.argv[4]) if len(sys.argv) > 4 else 2 * nsteps + 3
first = len(starts) - nsteps - tail
lo, hi = starts[first + 1], starts[first + nsteps - 1]
seg = rows[lo:hi]
n = nsteps - 2
ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1, r["Kernel_Name"]))
    ev.append((int(r["End_Timestamp"]), -1, r["Kernel_Name"]))
ev.sort()
active = collections.Counter()
excl = collections.Counter(); tot = collections.Counter(); cnt = collections.Counter()
prev = ev[0][0]
for t, d, name in ev:
    live = [k for k, v in active.items() if v > 0]
    if live and t > prev:
        for k in live:
            tot[k] += t - prev
        if len(live) == 1:
            excl[live[0]] += t - prev
    active[name] += d
    if d == 1: cnt[name] += 1
    prev = t
wall = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print("graph-replayed steps: %d, %.1f us per step" % (n, wall / n / 1e3))
short = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:60]
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-62s x%4.1f  %7.1f us/step  alone %7.1f" % (short(k), cnt[k] / n, v / n / 1e3, excl[k] / n / 1e3))

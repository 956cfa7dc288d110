"""Timeline of ONE graph-replayed step of bench.py from a rocprofv3 kernel trace: start offset (us since the step's first
kernel), duration, name -- which kernels overlap, where the device idles.  argv: trace.csv [steps=10] [tail forwards after the
timed region, see graph_step_profile.py] [which step of the timed region = 5]"""
import csv, sys
path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 2 * nsteps + 3
which = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r["Kernel_Name"]]
first = len(starts) - nsteps - tail
# a step begins with the copies / KNN that precede the rasterizer forward: cut at the Adam kernels of the previous step
i0 = starts[first + which]
while i0 > 0 and "adam_kernel" not in rows[i0 - 1]["Kernel_Name"]:
    i0 -= 1
i1 = starts[first + which + 1]
while i1 > 0 and "adam_kernel" not in rows[i1 - 1]["Kernel_Name"]:
    i1 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
short = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:56]
end_prev = 0.0
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    gap = s - end_prev
    print("%8.1f  %7.1f us  %s%s" % (s, e - s, short(r["Kernel_Name"]), "   <- idle %.1f us before" % gap if gap > 1.5 else ""))
    end_prev = max(end_prev, e)
print("step: %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))

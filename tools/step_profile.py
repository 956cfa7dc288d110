"""Per-step kernel summary from a rocprofv3 kernel_trace.csv of bench.py: averages over the last `n` steps
(a step = from one preprocess_fwd_kernel launch to the next)."""
import csv, sys, collections
path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r["Kernel_Name"]]
# step k spans [deform start ... next step's deform start): approximate with preprocess-to-preprocess windows
lo, hi = starts[-n - 1], starts[-1]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    name = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc[name][0] += 1
    acc[name][1] += d
tot = sum(v[1] for v in acc.values())
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print("steps %d  wall/step %.3f ms  kernel-busy/step %.3f ms  launches/step %.0f" % (n, span / n / 1e6, tot / n / 1e6, sum(v[0] for v in acc.values()) / n))
grp = collections.defaultdict(float)
for k, v in acc.items():
    g = "dgs raster" if k.startswith("dgs::") or "dgs::" in k else ("dgs train_ops" if "anonymous namespace)::" in k and "at::native" not in k else ("gemm" if k.startswith("Cijk") else ("adam" if "multi_tensor" in k else "torch elementwise/reduce/copy")))
    grp[g] += v[1]
for g, t in sorted(grp.items(), key=lambda x: -x[1]):
    print("  %-32s %.3f ms/step" % (g, t / n / 1e6))
for k, v in sorted(acc.items(), key=lambda x: -x[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print("%-100s x%5.1f  %7.1f us  %.3f ms/step" % (k[:100], v[0] / n, v[1] / v[0] / 1e3, v[1] / n / 1e6))

"""One step of tools/diag/concurrent_run.py from a rocprofv3 kernel trace: every kernel between two updates, by start time, with its
queue -- which kernels of the two lanes overlap.  argv: trace.csv [which step from the end = 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with its LAST adam_kernel launch (the update may be two launches: surfels, then the deformation): the one that is followed by
# the next step's first kernel (neighbour search or weight packing)
adam = [i for i, r in enumerate(rows[:-1]) if "adam_kernel" in r["Kernel_Name"] and any(k in rows[i + 1]["Kernel_Name"] for k in ("knn_refine", "mlp_pack", "knn_kernel"))]
i0, i1 = adam[-back - 1] + 1, adam[-back] + 1
t0 = int(rows[i0]["Start_Timestamp"])
short = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:48]
queues = {}
busy = []
for r in rows[i0:i1]:
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy.append((s, e))
    print("%8.1f %8.1f  %7.1f us  q%d  %s" % (s, e, e - s, q, short(r["Kernel_Name"])))
end = max(e for _, e in busy)
# device-idle time: union of the busy intervals
busy.sort()
cover, cur_s, cur_e = 0.0, busy[0][0], busy[0][1]
for s, e in busy[1:]:
    if s > cur_e:
        cover += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cover += cur_e - cur_s
nxt = (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3 if i1 < len(rows) else end
print("step: %.1f us to the next step's first kernel; last kernel ends at %.1f; some kernel running for %.1f us; sum of kernel durations %.1f us"
      % (nxt, end, cover, sum(e - s for s, e in busy)))

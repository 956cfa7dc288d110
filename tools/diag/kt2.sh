cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kr -o k -- python $GRAFT_REPO_ROOT/tools/diag/knn_timing.py > /tmp/kp.log 2>&1
tail -2 /tmp/kp.log
python - /tmp/kr <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "knn_refine" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:40], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
print({g: round(sorted(v)[len(v) // 2], 1) for g, v in d.items()})
PY

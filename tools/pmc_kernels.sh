#!/bin/bash
# PMC passes of the bench command restricted to kernels matching a regex (rocprofv3 --kernel-include-regex: everything else
# runs unprofiled, ~10 s per pass).  Counters are collected in SEPARATE passes, each with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2: they cannot share a pass).
#   usage on the GPU box: tools/pmc_kernels.sh <tag> '<regex>' [workload]   -> gpurun_out/pmc_<tag>.json
# The JSON records the command, the workload and the hash of the library sources the counters belong to; bench.py only
# reports a traffic figure whose hash matches the library it has loaded.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
REGEX=$2
WL=${3:-metric}
mkdir -p $OUT
export DGS_NO_GRAPHS=1
CMD="python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 --workload $WL --no-roofline-legs $BENCH_ARGS"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"
P5="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --kernel-include-regex "$REGEX" --pmc $P --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1 || { echo "pass $i ($P) failed:"; tail -5 $OUT/p$i.log; }
done
HASH=$(cd $R && python -c "import sys; sys.path.insert(0,'dynamic-2dgs_amd'); from diff_surfel_rasterization import _C; print(_C.source_hash())" 2>/dev/null | tail -1)
python - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1","p2","p3","p4","p5"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % p)
    if not f: print(p, "no counter file"); continue
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        k = re.sub(r"<.*>$", "", k)   # template arguments (blend_bwd_kernel<false>) are not part of the key
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for k, d in sorted(acc.items()):
    e = {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}
    e["launches_averaged"] = max(len(v) for v in d.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_traffic_bytes_per_launch"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    kern[k] = e
out = {"command": "DGS_NO_GRAPHS=1 $CMD  under  rocprofv3 --kernel-trace --kernel-include-regex '$REGEX' --pmc <one pass at a time>",
       "workload": "$WL", "library_source_hash": "$HASH",
       "correction": "FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 rocprofv3 reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section): "
                     "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; Infinity-Cache hits are counted, not excluded",
       "passes": ["$P1", "$P2", "$P3", "$P4", "$P5"], "kernels": kern}
json.dump(out, open("$R/gpurun_out/pmc_$1.json", "w"), indent=1)
for k, d in kern.items():
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print("%-30s VALU insts %.1fM  LDS %.1fM  SALU %.1fM | wait_any/wave_cycles %3.0f%% | FETCH %s KiB WRITE %s KiB -> traffic %.1f MB | TCC hit %.0f%%" % (
        k[:30], d.get("SQ_INSTS_VALU", 0) / 1e6, d.get("SQ_INSTS_LDS", 0) / 1e6, d.get("SQ_INSTS_SALU", 0) / 1e6,
        100 * d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), d.get("FETCH_SIZE"), d.get("WRITE_SIZE"),
        d.get("hbm_traffic_bytes_per_launch", 0) / 1e6, 100 * d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)))
PY
rm -rf $OUT

#!/bin/bash
# PMC passes restricted to kernels matching a regex (rocprofv3 --kernel-include-regex: everything else runs unprofiled,
# so this stays at ~10 s per pass).  usage on the GPU box: tools/pmc_kernels.sh <tag> '<regex>'  -> gpurun_out/pmc_<tag>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
REGEX=$2
mkdir -p $OUT
export DGS_NO_GRAPHS=1
CMD="python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE"
P4="SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --kernel-include-regex "$REGEX" --pmc $P --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1","p2","p3","p4"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % p)
    if not f: print(p, "no counter file"); continue
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in sorted(d.items())} for k, d in sorted(acc.items())}
json.dump(out, open("$R/gpurun_out/pmc_$1.json", "w"), indent=1)
for k, d in out.items():
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print("%-44s active %.1f us | VALU %3.0f%% LDS-inst %3.0f%% | wait_any/wave_cycles %3.0f%% wait_lds %3.0f%% wait_vmem %3.0f%% | insts VALU %.1fM LDS %.1fM VMEM %.2fM SALU %.1fM | bank conflict cycles %.2fM" % (
        k[:44], g / 2400.0, 100 * d.get("SQ_INSTS_VALU", 0) * 4 / 1024 / g, 100 * d.get("SQ_ACTIVE_INST_LDS", 0) / 1024 / g * 1.0,
        100 * d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), 100 * d.get("SQ_WAIT_INST_LDS", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
        100 * d.get("SQ_WAIT_INST_VMEM", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
        d.get("SQ_INSTS_VALU", 0) / 1e6, d.get("SQ_INSTS_LDS", 0) / 1e6, d.get("SQ_INSTS_VMEM", 0) / 1e6, d.get("SQ_INSTS_SALU", 0) / 1e6, d.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6))
PY
rm -rf $OUT

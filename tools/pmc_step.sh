#!/bin/bash
# PMC passes over the eager train step (development aid): tools/pmc_step.sh <tag> <kernel-name-regex>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
PAT=${2:-knn_refine|lbs_bwd|lbs_fwd|surfel_bwd|sort_tiles|adam|ssim_fwd|mlp_}
mkdir -p $OUT
export DGS_NO_GRAPHS=1
CMD="python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -o p -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections, re
pat = re.compile(r"$PAT")
for p in ("p1","p2","p3","p4"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % p)
    if not f: print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not pat.search(k): continue
        k = re.sub(r"\(anonymous namespace\)::|void |dgs::|mlp::", "", k).split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(p, k[:40], {c: round(sum(v)/len(v)) for c, v in acc[k].items()})
PY

#!/bin/bash
# PMC passes over the raster-only loop.  Separate passes (MI355X_MICROARCH.md): SQ issue mix, SQ waits, HBM traffic.
# usage (on the GPU box): tools/pmc_blend.sh <tag>   -> gpurun_out/pmc_<tag>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
CMD="python $R/tools/quick_timing.py --iters 3 --views 4"
INC="--kernel-include-regex dgs::"
rocprofv3 --kernel-trace $INC --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace $INC --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace $INC --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace $INC --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -o p -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1","p2","p3","p4"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % p)
    if not f: print(p, "no counter file"); continue
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "dgs::" not in k: continue
        acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"correction": "FETCH_SIZE and WRITE_SIZE are KiB; gfx950 rocprofv3 reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section): traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "command": "$CMD (rasterizer forward + backward only, 200k surfels, 800x800; four separate --pmc passes)", "kernels": {}}
for k in sorted(acc):
    d = {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    out["kernels"][k] = d
json.dump(out, open("$R/gpurun_out/pmc_$1.json", "w"), indent=1)
for k in ("dgs::blend_fwd_kernel", "dgs::blend_bwd_kernel"):
    d = out["kernels"].get(k, {})
    if d: print(k, "VALU busy %.0f%% of GRBM_GUI_ACTIVE" % (100 * d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / max(d.get("GRBM_GUI_ACTIVE", 1), 1)), "traffic MB", d.get("hbm_traffic_bytes_per_launch", 0) / 1e6)
PY
rm -rf $OUT

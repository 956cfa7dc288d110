#!/bin/bash
# On the GPU box: headline ms/step of bench.py under every A/B build of the rasterizer library (csrc/ab_*.so, tools/ab_variants.sh build), alternating.
#   tools/diag/ab_lib.sh [repeats=3]
R=$(cd "$(dirname "$0")/../.." && pwd)
for i in $(seq ${1:-3}); do for so in $R/dynamic-2dgs_amd/csrc/ab_*.so; do printf "%-10s " $(basename $so .so); DGS_SURFEL_LIB=$so python $R/bench.py --no-cpu-baseline --no-roofline-legs --drift-gap 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done

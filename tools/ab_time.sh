#!/bin/bash
# On the GPU box: timing only of every A/B build (no parity run -- for diagnostic builds whose results are wrong by design).
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/dynamic-2dgs_amd/csrc
for so in $D/ab_*.so; do
  printf "%-14s " $(basename $so .so)
  DGS_SURFEL_LIB=$so timeout 300 python $R/tools/quick_timing.py "$@" 2>&1 | tail -3 | tr '\n' ' '
  echo
done

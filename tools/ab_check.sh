#!/bin/bash
# On the GPU box: parity of every A/B build (tests/test_gpu_parity.py through DGS_SURFEL_LIB), then its timing on one lease.
#   tools/ab_check.sh [quick_timing args]
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/dynamic-2dgs_amd/csrc
for so in $D/ab_*.so; do
  n=$(basename $so .so)
  printf "%-14s parity: " $n
  DGS_SURFEL_LIB=$so timeout 600 python -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
done
for rep in 1 2; do
for so in $D/ab_*.so; do
  printf "%-14s " $(basename $so .so)
  DGS_SURFEL_LIB=$so timeout 300 python $R/tools/quick_timing.py "$@" 2>&1 | tail -3 | tr '\n' ' '
  echo
done
done

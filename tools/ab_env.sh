#!/bin/bash
# On the GPU box: bench.py headline (no legs) under several environment settings, interleaved, N rounds; then one kernel timeline.
#   tools/ab_env.sh ROUNDS "NAME=VALUE ..." "NAME=VALUE ..." ...
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
for r in $(seq 1 $N); do
  for e in "$@"; do
    printf "%-40s " "$e"
    env $e python $R/bench.py --no-cpu-baseline --no-roofline-legs 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
  done
done

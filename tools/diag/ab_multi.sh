#!/bin/bash
# On the GPU box: headline ms/step of bench.py under several environment settings, round-robin.  usage: tools/diag/ab_multi.sh REPEATS "A=0" "A=1 B=2" ...
R=$1; shift
for i in $(seq $R); do for e in "$@"; do echo -n "$e: "; env $e python bench.py --no-cpu-baseline --no-roofline-legs --drift-gap 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done

#!/bin/bash
# On the GPU box: headline ms/step of bench.py under environment settings, alternating.  usage: tools/diag/ab_env.sh "A=0" "A=1" [repeats=2]
R=${3:-2}
for i in $(seq $R); do for e in "$1" "$2"; do echo -n "$e: "; env $e python bench.py --no-cpu-baseline --no-roofline-legs --drift-gap 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done

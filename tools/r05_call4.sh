#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== trained, deterministic pre-fit, three times"
for i in 1 2 3; do timeout 900 python bench.py --workload trained --no-cpu-baseline --no-roofline-legs 2> $O/trained_$i.err | tee $O/trained_$i.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); c = d['config']
print(d['value'], d['ms_per_step'], c.get('pre_fit'), c.get('pre_fit_fingerprint'))"; done

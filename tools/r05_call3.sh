#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== trained, deterministic pre-fit, twice"
for i in 1 2; do S=$(date +%s); timeout 900 python bench.py --workload trained --no-cpu-baseline --no-roofline-legs 2> $O/trained_$i.err | tee $O/trained_$i.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); c = d['config']
print(d['value'], d['ms_per_step'], c.get('pre_fit'), c.get('pre_fit_fingerprint'))"; echo "wall $(( $(date +%s) - S )) s"; tail -2 $O/trained_$i.err | cut -c1-300; done
echo "== pytest -m gpu"; timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log

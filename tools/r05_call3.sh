#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest -m gpu"; timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
echo "== bench: packet capture knob 0 / 1"
bash tools/ab_env.sh 3 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
echo "== trained, deterministic pre-fit, twice"
for i in 1 2; do /usr/bin/time -f "%e s" timeout 600 python bench.py --workload trained --no-cpu-baseline --no-roofline-legs 2> $O/trained_$i.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); c = d['config']
print(d['value'], d['ms_per_step'], c.get('pre_fit'), c.get('pre_fit_fingerprint'))"; tail -1 $O/trained_$i.err; done
echo "== timeline"
bash tools/timeline.sh r05a > /dev/null 2>&1; tail -30 $O/r05a_timeline.txt

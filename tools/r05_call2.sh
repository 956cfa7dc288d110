#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== bench: MLP backward split on / off"
bash tools/ab_env.sh 3 "DGS_MLP_SPLIT=1" "DGS_MLP_SPLIT=0"
echo "== pytest -m gpu"; timeout 1200 python -X faulthandler -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
echo "== timeline"
bash tools/timeline.sh r05a > /dev/null 2>&1; tail -40 $O/r05a_timeline.txt
echo "== graph knob probe"
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DGS_ALLOW_GRAPH_PACKET_CAPTURE=1 timeout 400 python tools/diag/graph_knob_probe.py 2000 2>&1 | tail -1 | tee $O/knob_on.json
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 400 python tools/diag/graph_knob_probe.py 2000 2>&1 | tail -1 | tee $O/knob_off.json

#!/bin/bash
# Round 5, first lease: the GPU suite on the restructured sources, the backward-blend A/B builds, the node MLP backward split on / off,
# a step timeline, and the graph-replay knob probe.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== A/B timing (rasterizer alone, 200k / 800x800)"
for rep in 1 2; do bash tools/ab_time.sh --iters 40; done
echo "== parity of the one-reciprocal build"
DGS_SURFEL_LIB=$R/dynamic-2dgs_amd/csrc/ab_2onercp.so DGS_PARITY_FILE=parity_onercp.json timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
echo "== bench: MLP backward split on / off"
bash tools/ab_env.sh 2 "DGS_MLP_SPLIT=1" "DGS_MLP_SPLIT=0"
echo "== timeline"
bash tools/timeline.sh r05a > /dev/null 2>&1; cat $O/r05a_timeline.txt | tail -40
echo "== graph knob probe"
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DGS_ALLOW_GRAPH_PACKET_CAPTURE=1 timeout 400 python tools/diag/graph_knob_probe.py 2000 2>&1 | tail -1 | tee $O/knob_on.json
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 400 python tools/diag/graph_knob_probe.py 2000 2>&1 | tail -1 | tee $O/knob_off.json

#!/bin/bash
# Everything the round's measurement claims rest on, in one bounded run on the GPU box:
#   usage (via gpurun, from the repo root): tools/round_artifacts.sh <tag>     -> gpurun_out/<tag>_*
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench (with cpu baseline)"; timeout 400 python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; tail -c 300 $O/${TAG}_bench_line.json
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel trace of the bench command"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o k -- python $R/bench.py --no-cpu-baseline --no-roofline-legs --steps 10 > $O/${TAG}_prof.log 2>&1
ls $O/${TAG}_prof | head
F=$(ls $O/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1); T=$(ls $O/${TAG}_prof/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$F" ] && cp $F $O/${TAG}_bench_graph_200k_800_kernel_stats.csv
[ -n "$T" ] && python $R/tools/graph_step_profile.py $T 10 34 0 > $O/${TAG}_graph_step_summary.txt 2>&1 && head -12 $O/${TAG}_graph_step_summary.txt
[ -n "$T" ] && python $R/tools/graph_step_timeline.py $T 10 0 5 > $O/${TAG}_graph_step_timeline.txt 2>&1
rm -rf $O/${TAG}_prof
echo "== PMC passes on the blend kernels"
cd $R && bash tools/pmc_kernels.sh ${TAG} 'dgs::blend' 2>&1 | tail -4

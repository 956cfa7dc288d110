#!/bin/bash
# On the GPU box: kernel timeline of one replayed step of bench.py (rocprofv3 --kernel-trace) -> gpurun_out/<tag>_timeline.txt
#   tools/timeline.sh TAG [ENV=VALUE ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o k -- python $R/bench.py --no-cpu-baseline --no-roofline-legs --steps 10 $BENCH_ARGS > $O/${TAG}_prof.log 2>&1
T=$(ls $O/${TAG}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/graph_step_timeline.py $T 10 0 5 > $O/${TAG}_timeline.txt 2>&1
python $R/tools/graph_step_profile.py $T 10 40 0 > $O/${TAG}_summary.txt 2>&1
rm -rf $O/${TAG}_prof
cat $O/${TAG}_timeline.txt

#!/bin/bash
# On the GPU box: tools/diag/concurrent_trace.sh TAG [k] [concurrent] -> gpurun_out/<TAG>_conc_timeline.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; K=${2:-2}; C=${3:-1}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_prof -o k -- python $R/tools/diag/concurrent_run.py 12 $K $C > $O/${TAG}_conc.log 2>&1
T=$(ls $O/${TAG}_prof/*kernel_trace.csv $O/${TAG}_prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/diag/concurrent_timeline.py $T 3 > $O/${TAG}_conc_timeline.txt 2>&1
rm -rf $O/${TAG}_prof
tail -3 $O/${TAG}_conc.log
cat $O/${TAG}_conc_timeline.txt

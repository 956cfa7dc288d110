#!/bin/bash
# On the GPU box: tools/diag/dp_trace.sh TAG [shard] -> gpurun_out/<TAG>_dp_timeline.txt (one step of the data-parallel structure, single rank)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; S=${2:-1}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_prof -o k -- python $R/tools/diag/dp_step_trace.py 12 $S > $O/${TAG}_dp.log 2>&1
T=$(ls $O/${TAG}_prof/*kernel_trace.csv $O/${TAG}_prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/graph_step_timeline.py $T 12 0 5 > $O/${TAG}_dp_timeline.txt 2>&1
rm -rf $O/${TAG}_prof
tail -2 $O/${TAG}_dp.log
cat $O/${TAG}_dp_timeline.txt

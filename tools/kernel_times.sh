# Dev tool (GPU box): per-kernel timeline of one graph-replayed step of the metric workload.  usage: bash tools/kernel_times.sh [workload]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; W=${1:-metric}
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_prof -o k -- python $R/bench.py --workload $W --no-cpu-baseline --no-roofline-legs --steps 10 > $O/kt_prof.log 2>&1
T=$(ls $O/kt_prof/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/graph_step_profile.py $T 10 0 0 2>&1 | head -3
python $R/tools/graph_step_timeline.py $T 10 0 5 2>&1 | tee $O/kt_timeline.txt | tail -40
grep -o '"value": [0-9.]*' $O/kt_prof.log | head -1
rm -rf $O/kt_prof

# Bench lines (metric with CPU baseline, C2-C4), kernel stats, per-step summary and timeline of the final code -> gpurun_out/r02_c_*
# (PMC=1: also the counter passes of the C5 workload; the metric's are taken by tools/round_artifacts.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r02_c
cd $R
[ "$PMC" = "1" ] && bash tools/pmc_kernels.sh r02_blend_c5 "dgs::blend" c5 2>&1 | tail -2   # only when the rasterizer library changed
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; tail -c 200 $O/${TAG}_bench_line.json; echo
for w in c2 c3 c4; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_line_$w.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o k -- python $R/bench.py --no-cpu-baseline --no-roofline-legs --steps 10 > $O/${TAG}_prof.log 2>&1
F=$(ls $O/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1); T=$(ls $O/${TAG}_prof/*kernel_trace.csv 2>/dev/null | head -1)
cp $F $O/${TAG}_bench_graph_200k_800_kernel_stats.csv
python $R/tools/graph_step_profile.py $T 10 34 0 > $O/${TAG}_graph_step_summary.txt 2>&1; head -8 $O/${TAG}_graph_step_summary.txt
python $R/tools/graph_step_timeline.py $T 10 0 5 > $O/${TAG}_graph_step_timeline.txt 2>&1; tail -3 $O/${TAG}_graph_step_timeline.txt
rm -rf $O/${TAG}_prof

# Round artefacts of the final code, two phases on the GPU box (the bench lines read the PMC files of phase 1 from profiles/):
#   phase 1:  tools/final_artifacts.sh pmc      -> gpurun_out/pmc_r06_metric.json, pmc_r06_c5.json  (copy to profiles/r06_pmc_<workload>.json)
#   phase 2:  tools/final_artifacts.sh lines    -> gpurun_out/r04_bench_line*.json, kernel stats, per-step summary, timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r06
cd $R
if [ "$1" = "det" ]; then   # what the deterministic mode costs per step (metric workload, captured step, same lease)
  python - <<'PY'
import os, sys, time
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"], "dynamic-2dgs_amd")]
import torch, bench
tr = bench.build_trainer(200_000, 800, 800, torch.device("cuda", 0))
tr.enable_graph(capacity=24 * 200_000)
def timed(n=60):
    for _ in range(10): tr.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): tr.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
snap = tr._snapshot(); it0 = tr.iteration
for rep in range(3):
    a = timed(); tr._restore(snap); tr.iteration = it0
    tr.set_deterministic(True); b = timed(); tr._restore(snap); tr.iteration = it0
    tr.set_deterministic(False)
    print("ms per step: float atomics %.4f, deterministic (fixed-point sums, serial walks) %.4f  (+%.1f %%)" % (a, b, 100 * (b / a - 1)))
PY
  exit 0
fi
if [ "$1" = "pmc" ]; then
  bash tools/pmc_kernels.sh r06_metric "dgs::" metric 2>&1 | tail -12
  bash tools/pmc_kernels.sh r06_c5 "dgs::" c5 2>&1 | tail -12
  exit 0
fi
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; tail -c 200 $O/${TAG}_bench_line.json; echo
for w in c2 c3 c4 c5; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_line_$w.json 2>/dev/null; done
for i in "" _2 _3; do timeout 400 python bench.py --workload trained --no-cpu-baseline > $O/${TAG}_bench_line_trained$i.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o k -- python $R/bench.py --no-cpu-baseline --no-roofline-legs --steps 10 > $O/${TAG}_prof.log 2>&1
F=$(ls $O/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1); T=$(ls $O/${TAG}_prof/*kernel_trace.csv 2>/dev/null | head -1)
cp $F $O/${TAG}_bench_graph_200k_800_kernel_stats.csv
python $R/tools/graph_step_profile.py $T 10 40 0 > $O/${TAG}_graph_step_summary.txt 2>&1; head -8 $O/${TAG}_graph_step_summary.txt
python $R/tools/graph_step_timeline.py $T 10 0 5 > $O/${TAG}_graph_step_timeline.txt 2>&1; tail -3 $O/${TAG}_graph_step_timeline.txt
rm -rf $O/${TAG}_prof

# A/B of the blend tile orders 3 (longest first) and 4 (XCD-local groups): step time, kernel times, ordering kernels, HBM traffic
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_order" 2>&1 | tail -1
for m in 3 4 3 4 3 4; do python bench.py --no-cpu-baseline --steps 40 --tile-order $m --no-roofline-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('order $m:', d['value'], 'views/s', d['ms_per_step'], 'ms')"; done
cd /tmp && export TMPDIR=/tmp
for m in 3 4; do timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k$m -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline-legs --steps 10 --tile-order $m > /tmp/log 2>&1; T=$(ls /tmp/k$m/*kernel_trace.csv | head -1); echo "order $m"; python $GRAFT_REPO_ROOT/tools/graph_step_timeline.py $T 10 0 5 | grep -i "scan_tiles\|tile_order\|blend\|step:"; done

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== bench, split on"; timeout 300 python -X faulthandler bench.py --no-cpu-baseline --no-roofline-legs --steps 5 > $O/crash_bench.log 2>&1; grep -n "Fatal\|fault\|Fault\|rror" $O/crash_bench.log | head -20; grep -n "Fatal" -A30 $O/crash_bench.log | head -60; tail -3 $O/crash_bench.log | cut -c1-300
echo "== bench tiny, split on, no graphs"; DGS_NO_GRAPHS=1 timeout 300 python -X faulthandler bench.py --workload tiny --no-cpu-baseline --no-roofline-legs --steps 5 > $O/crash_bench2.log 2>&1; grep -n "Fatal" -A30 $O/crash_bench2.log | head -50; tail -2 $O/crash_bench2.log | cut -c1-300
echo "== uhd test"; timeout 600 python -m pytest tests/test_gpu_parity.py -k "uhd" -x -q > $O/uhd.log 2>&1; tail -5 $O/uhd.log

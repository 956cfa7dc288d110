"""-m gpu: bench.py's OWN multi-rank path (process-group set-up, barriers, the MIN / MAX all-reduces around the timed region,
the rank-0 JSON line), run as two processes.  The test box has one GPU and RCCL refuses two ranks on one device, so the
collectives go over gloo (DGS_DIST_BACKEND) with both ranks on cuda:0 -- everything else is the code the driver's
`torch.distributed.run --nproc-per-node N bench.py --gpus N` executes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_prints_one_valid_line():
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DGS_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
                                       "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")], "exactly one JSON line, from rank 0"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 2 and r["scaling"] == "weak" and r["unit"] == "views/s"
    assert r["config"]["views_per_step"] == 2 and r["config"]["parallelism"].startswith("dp2")
    assert r["value"] > 0 and abs(r["value"] - 2 * 1e3 / r["ms_per_step"]) <= 1e-2 * r["value"]   # whole-job rate = world views per step
    assert r["roofline"] and r["roofline"]["kernel"] == "blend_bwd_kernel" and 0 < r["roofline"]["frac"] < 1
    assert "cpu_baseline" not in r   # rank 0 at N = 1 only
    # the communication object (N > 1): every slice timed alone, the step without collectives, what stays exposed, the 2 / 3 piece choice
    c = r["comm"]
    assert c["world"] == 2 and c["backend"] == "gloo" and c["wire_bytes_per_step"]["total"] > 50e6
    assert set(c["slices"]) >= {"sh", "rest", "radii"} and all(v["ms"] > 0 and v["bus_GBs"] > 0 for v in c["slices"].values())
    assert 0 < c["ms_per_step_no_collectives"] < r["ms_per_step"] and abs(c["exposed_ms_per_step"] - (r["ms_per_step"] - c["ms_per_step_no_collectives"])) < 1e-3
    assert c["split3"]["chosen"] in ("two", "three") and c["split3"]["ms_per_step_two_pieces"] > 0 and c["split3"]["ms_per_step_three_pieces"] > 0


def test_bench_two_ranks_two_views_per_rank():
    """The same launch with --views-per-rank 2 (Trainer.views_per_rank: two views per rank added before the one exchange of a step):
    four views per step, the rate counts views, one exchange per step (no split: the SH slice is part of `rest`), and the
    communication object reports what stays exposed per VIEW."""
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DGS_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
                                       "--no-cpu-baseline", "--no-roofline-legs", "--views-per-rank", "2"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["views_per_step"] == 4 and r["config"]["views_per_rank"] == 2
    assert r["value"] > 0 and abs(r["value"] - 4 * 1e3 / r["ms_per_step"]) <= 1e-2 * r["value"]
    c = r["comm"]
    assert c["views_per_rank"] == 2 and c["split3"] is None and "sh" not in c["slices"] and set(c["slices"]) >= {"rest", "radii"}
    assert abs(c["exposed_ms_per_view"] - c["exposed_ms_per_step"] / 2) < 1e-3

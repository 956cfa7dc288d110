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
    assert set(c["slices"]) >= {"sh_reduce_scatter", "sh_all_gather", "rest", "radii"} and all(v["ms"] > 0 and v["bus_GBs"] > 0 for v in c["slices"].values())
    assert c["sh_collective"].startswith("reduce_scatter") and c["levers"]["views_per_rank_2"]["views_per_step"] == 4 and c["levers"]["wire_bf16_on"]["ms_per_view"] > 0
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


def test_bench_eight_ranks_dry_run():
    """The driver's 8-GPU command, as eight processes sharing the one GPU of the test box over gloo: everything that is first-contact
    code at world = 8 -- eight loaders on one library, the view schedule, the MAX-over-ranks votes, eight slices per collective, the
    sharded SH update with eight owners, the levers of the communication object -- runs here first."""
    world, port = 8, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DGS_DIST_BACKEND="gloo", DGS_LEVER_CONCURRENT="1")   # (the concurrent-lanes lever is opt-in at N > 1: covered here)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2", "--drift-gap", "10"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not any(l.startswith("{") for so, _ in outs[1:] for l in so.splitlines()), "exactly one JSON line, from rank 0"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 4 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["config"]["views_per_step"] == 8 and r["config"]["parallelism"].startswith("dp8")
    assert sorted(r["config"]["views_first_step"]) == list(range(8))          # every rank its own view of the step
    assert r["value"] > 0 and abs(r["value"] - 8 * 1e3 / r["ms_per_step"]) <= 1e-2 * r["value"]
    assert "cpu_baseline" not in r and r["twin"] is not None
    c = r["comm"]
    assert c["world"] == 8 and c["sh_collective"].startswith("reduce_scatter")
    assert set(c["slices"]) >= {"sh_reduce_scatter", "sh_all_gather", "rest", "radii"} and all(v["ms"] > 0 for v in c["slices"].values())
    assert c["wire_bytes_per_step"]["sh"] == 4 * 48 * 200_000
    lv = c["levers"]
    assert lv["wire_bf16_on"]["ms_per_view"] > 0 and lv["wire_bf16_on"]["wire_bytes_per_step"] < c["wire_bytes_per_step"]["total"]
    assert lv["views_per_rank_2"]["views_per_step"] == 16 and lv["views_per_rank_2"]["ms_per_view"] > 0
    assert lv["views_per_rank_2_concurrent"]["views_per_step"] == 16 and lv["views_per_rank_2_concurrent"]["ms_per_view"] > 0


def test_bench_watchdog_prints_the_headline_alone():
    """bench.py's watchdog: when the measurements behind the timed region do not finish in time (here: no time at all), rank 0 still
    prints ONE line -- metric, value, steps, config -- and the process leaves with status 0."""
    env = dict(os.environ, DGS_BENCH_WATCHDOG_S="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "4", "--warmup", "2"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert "headline only" in r["note"] and r["value"] > 0 and r["n_gpus"] == 1 and r["steps"] == 4 and r["unit"] == "views/s"
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) <= 1e-2 * r["value"] and "roofline" not in r

"""N ranks of one launch may all find a native library stale (the .srchash files are not tracked, a checkout changes the sources):
`_dgs_build.build` has to let exactly one of them compile, never expose a half-written file at the library's path, and leave the
others with the finished binary ("first contact" of the multi-GPU path)."""
import multiprocessing as mp
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))

PAYLOAD = 1 << 20
# a slow "compiler": writes its output in pieces (a reader that opened the path meanwhile would see a short file) and logs the run
FAKE_CC = r"""
import sys, time, os
out, log = sys.argv[1], sys.argv[2]
with open(log, "a") as f:
    f.write("compile " + str(os.getpid()) + "\n")
with open(out, "wb") as f:
    for _ in range(8):
        f.write(b"x" * (%d // 8)); f.flush(); time.sleep(0.05)
""" % PAYLOAD


def _rank(args):
    lib, log, dep, start_at = args
    import _dgs_build
    while time.time() < start_at:
        pass
    seen_short = False
    cmd = [sys.executable, "-c", FAKE_CC, lib, log]
    _, how = _dgs_build.build(lib, cmd, [dep], ["-O3"], os.path.dirname(lib))
    size = os.path.getsize(lib)
    return how, size, seen_short


def _watch(args):
    lib, stop_at = args
    sizes = set()
    while time.time() < stop_at:
        try:
            sizes.add(os.path.getsize(lib))
        except OSError:
            sizes.add(-1)
    return sizes


@pytest.mark.parametrize("state", ["stale_hash", "no_hash", "no_library"])
def test_four_processes_one_compile(tmp_path, state):
    import _dgs_build
    lib, log, dep = str(tmp_path / "libfake.so"), str(tmp_path / "log.txt"), str(tmp_path / "src.hip")
    with open(dep, "w") as f:
        f.write("// source\n")
    old = b"o" * PAYLOAD
    if state != "no_library":
        with open(lib, "wb") as f:
            f.write(old)
    if state == "stale_hash":
        with open(lib + ".srchash", "w") as f:
            f.write("0123456789abcdef\n")
    ctx = mp.get_context("spawn")
    start = time.time() + 1.0
    with ctx.Pool(5) as pool:
        watcher = pool.apply_async(_watch, ((lib, start + 2.5),))
        res = pool.map(_rank, [(lib, log, dep, start)] * 4)
        sizes = watcher.get()
    hows = sorted(r[0] for r in res)
    assert hows == ["compiled", "reused", "reused", "reused"], hows
    assert all(r[1] == PAYLOAD for r in res)
    with open(log) as f:
        assert len(f.read().strip().split("\n")) == 1          # one compiler run
    # the path never showed a partial file: only "absent" (no_library, before the rename) or a complete old / new binary
    assert sizes <= {-1, PAYLOAD}, sizes
    if state != "no_library":
        assert -1 not in sizes
    with open(lib, "rb") as f:
        assert f.read() == b"x" * PAYLOAD
    assert _dgs_build.recorded_hash(lib) == _dgs_build.source_hash([dep], ["-O3"])
    left = [n for n in os.listdir(tmp_path) if ".tmp" in n]
    assert not left, left


def test_failed_compile_keeps_the_old_binary(tmp_path):
    import subprocess
    import _dgs_build
    lib, dep = str(tmp_path / "libfake.so"), str(tmp_path / "src.hip")
    with open(dep, "w") as f:
        f.write("// source\n")
    with open(lib, "wb") as f:
        f.write(b"old")
    cmd = [sys.executable, "-c", "import sys; open(sys.argv[1], 'wb').write(b'half'); sys.exit(3)", lib]
    with pytest.raises(subprocess.CalledProcessError):
        _dgs_build.build(lib, cmd, [dep], [], str(tmp_path))
    with open(lib, "rb") as f:
        assert f.read() == b"old"
    assert _dgs_build.recorded_hash(lib) is None
    assert not [n for n in os.listdir(tmp_path) if ".tmp" in n]

import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # see Trainer.enable_graph (must precede torch's HIP init)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-2dgs_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions: what the parity comparisons measured (tests/gpu_utils.py PARITY_LOG) -> gpurun_out/parity_r06.json."""
    try:
        import gpu_utils
    except Exception:
        return
    log = getattr(gpu_utils, "PARITY_LOG", None)
    if not log:
        return
    import json
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, os.environ.get("DGS_PARITY_FILE", "parity_r06.json")), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "records": log}, f, indent=0)
    except OSError:
        pass

"""Dev tool: ONE fitted `trained` bench scene, kept as a checkpoint, so that A/B runs on a densified scene compare kernels and not fits
(the fit's outcome varies from run to run: 65-115 k live surfels).
    on the GPU box:   python tools/diag/trained_cache.py build     -> gpurun_out/trained_ckpt/ (point_cloud.ply + deform.pth, ~30 MB)
    here:             mv gpurun_out/trained_ckpt tools/diag/cache/   (git-ignored, travels with the snapshot)
    in a probe:       from trained_cache import load; tr = load(device)   (dataset regenerated from its seed, no fit)"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
CACHE = os.path.join(ROOT, "tools", "diag", "cache", "trained_ckpt")


def _dataset(tmp, H, W, device):
    from dgs_amd.synthetic import DynamicTruth, write_dynamic_dnerf
    write_dynamic_dnerf(os.path.join(tmp, "scene"), n_train=48, n_test=2, H=H, W=W, device=device, truth=DynamicTruth(24000, 16000, detail=0.3))
    return os.path.join(tmp, "scene")


def load(device, H=800, W=800, slots=125000):
    """Trainer on the cached scene: late regime, fresh Adam moments, sorted storage order; not captured yet."""
    import torch
    from dgs_amd import io as dio
    from dgs_amd.fit import restore
    from dgs_amd.train import Trainer
    if not os.path.isdir(CACHE):
        raise FileNotFoundError("no cached scene under %s: run `trained_cache.py build` on the GPU box first" % CACHE)
    tmp = tempfile.mkdtemp(prefix="dgs_trained_")
    try:
        data = dio.load_dnerf(_dataset(tmp, H, W, device), white_background=False, num_pts=1000, seed=0)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    surfels, deform = restore(CACHE, device=device, packed_sh=True, slots=slots)
    cams = [f.camera.to(device) for f in data["train"]]
    targets = [f.image.to(device).contiguous() for f in data["train"]]
    tr = Trainer(surfels, deform, cams, targets, torch.zeros(3, device=device), lr_schedule=True)
    tr.sort_surfels()
    tr.refresh_knn_mode()
    tr.set_regime(warmup=False, lambda_normal=0.02, lambda_dist=1000.0)
    return tr


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "build":
    import torch
    import bench
    from dgs_amd.fit import save
    tr, _ = bench.trained_trainer(100_000, 800, 800, torch.device("cuda:0"), 10000)
    out = os.path.join(ROOT, "gpurun_out", "trained_ckpt")
    shutil.rmtree(out, ignore_errors=True)
    save(tr, out, 10000)
    print("live surfels", tr.surfels.num_surfels, "->", out)

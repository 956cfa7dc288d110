"""BASELINE.json configs C3 (150 k surfels, 800x800), C4 (300 k, 800x800) and C5 (1 M, 1600x1600) on the GPU box, single GPU:
the HIP path through the operator surface against the OpenMP oracle (image, radii, gradients) plus the size-independent
properties, and one graph-replayed train step against the eager step at C5.  These sizes exercise what the small cases do
not: 10 000 tiles (40 KB LDS histograms), int offsets at ~10 M list entries, 4 MB ray tables, MB-sized SH staging.
(The reference shapes: SURVEY.md section 8 header, rasterizer_impl.cu:198-342.  Real D-NeRF data is not in the container:
C3/C4 use the synthetic generator of SURVEY 8d at the named sizes.)"""
import numpy as np
import pytest
import torch

from scene_utils import oracle_from_case, small_case

pytestmark = pytest.mark.gpu

CONFIGS = {
    "c3": dict(P=150_000, H=800, W=800, view=9),
    "c4": dict(P=300_000, H=800, W=800, view=23),
    "c5": dict(P=1_000_000, H=1600, W=1600, view=41),
}


def _cot(H, W, seed=5):
    g = np.random.default_rng(seed)
    return g.standard_normal((3, H, W)).astype(np.float32), g.standard_normal((8, H, W)).astype(np.float32)


@pytest.mark.parametrize("name", ["c3", "c4", "c5"])
def test_config_matches_oracle_and_properties(name):
    from gpu_utils import frac_close, img_close, grad_close, median_flips, rel_l2, run_hip, hip_median_contrib
    cfg = CONFIGS[name]
    case = small_case(P=cfg["P"], H=cfg["H"], W=cfg["W"], seed=0, view=cfg["view"], n_views=64)
    gc, go = _cot(cfg["H"], cfg["W"])
    a = run_hip(case, gc, go, debug=False)
    # ---- size-independent properties
    b = run_hip(case, 2.0 * gc, 2.0 * go, debug=False)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["allmap"], b["allmap"]) and np.array_equal(a["radii"], b["radii"])  # forward is bitwise deterministic
    alpha = a["allmap"][1]
    assert alpha.min() >= 0.0 and alpha.max() <= 1.0 and np.isfinite(a["allmap"]).all() and np.isfinite(a["color"]).all()
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh", "dL_dmeans2D"):
        assert np.isfinite(a[k]).all(), k
        assert rel_l2(b[k], 2.0 * a[k]) <= 1e-6, k                       # backward is linear in the cotangent
    w = run_hip(dict(case, bg=torch.tensor([1.0, 1.0, 1.0])), debug=False)
    assert np.abs((w["color"] - a["color"]) - (1.0 - alpha)[None]).max() <= 1e-6   # colour = C + T * bg
    vis = a["radii"] > 0
    assert vis.sum() > 0.3 * cfg["P"]
    for k in ("dL_dmeans3D", "dL_dsh"):
        assert not a[k][~vis].any(), k                                   # culled surfels receive no gradient
    # ---- against the oracle (OpenMP, same inputs)
    orc = oracle_from_case(case)
    assert float((a["radii"] != orc.radii).mean()) <= 1e-4
    img_close(a["color"], orc.color, "color", max_bad_frac=6e-4, hard=6e-3)   # observed (C3 / C4 / C5) <= 1.9e-4 of the pixels, max 1.7e-3
    # Channels 5 and 7 (median depth, median weight) are the depth / weight of ONE contributor, the last one blended while
    # T > 0.5 (forward.cu:416-420): a pixel whose T sits on 0.5 to rounding picks a neighbouring contributor and jumps by the
    # difference between the two (seen at 1 M / 1600x1600: weight 0.500001 vs 0.028).  median_flips proves, pixel by pixel
    # from the oracle's own trace, that every pixel whose pick differs reports an entry the oracle blends too and differs
    # from the oracle's pick only across entries blended at T = 0.5 +- 2e-4; all other pixels are held to the tolerance of
    # the summed channels.
    sums, medians = [0, 1, 2, 3, 4, 6], [5, 7]
    frac_close(a["allmap"][sums], orc.allmap[sums], 1e-4, 5e-5, 8e-5, 1.5e-2, "allmap")   # observed 2.7e-5 of the entries, max 5.0e-3; C5 sums 1-3 k terms: 1.2e-3 of them beyond 1e-5 max(1, |x|)
    flips = median_flips(hip_median_contrib(case), orc)
    frac_close(a["allmap"][medians][:, ~flips], orc.allmap[medians][:, ~flips], 1e-4, 5e-5, 4e-5, 8e-3, "median depth / weight")   # observed 1.1e-5, max 2.7e-3
    mse = float(((a["color"] - orc.color) ** 2).mean())
    assert mse < 1e-9                                                     # PSNR > 90 dB for a [0, 1] image
    og = orc.backward(gc, go)
    # Gradients: rel-L2 <= 2e-4 once the 1e-3 worst-conditioned surfels are set aside (near edge-on splats, contributors on
    # the 1/255 and 1e-4 thresholds).  What those few carry is bounded by the yardstick the arithmetic itself offers: the
    # fp32 and fp64 builds of the ORACLE differ on the same surfels, and the HIP path must not be further from the fp64
    # oracle than a small multiple of the fp32 oracle's own distance.
    o64 = oracle_from_case(case, dtype=np.float64)
    og64 = o64.backward(gc.astype(np.float64), go.astype(np.float64))
    # untrimmed bound per gradient: <= 3x the worst value observed over C3 / C4 / C5 (profiles/r05_parity_margins.md: 8.3e-3, 7.3e-3, 1.4e-3,
    # 8.8e-3, 1.5e-3, 3.3e-5) -- rounds 3-5 asserted max(5e-3, 4 x the fp32 oracle's own distance from the fp64 one), which came to 0.15-0.20
    TOL_ALL = {"dL_dmeans3D": 2.5e-2, "dL_dmeans2D": 2.2e-2, "dL_dscales": 4.2e-3, "dL_drotations": 2.6e-2, "dL_dopacity": 4.5e-3, "dL_dsh": 1e-4}
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"):
        own = rel_l2(og[k], og64[k])
        grad_close(a[k], og[k], k, tol_trim=5e-5, tol_all=TOL_ALL[k])   # trimmed: observed <= 1.6e-5
        # ... and the HIP path is no further from the fp64 oracle than the fp32 oracle itself is (observed ratio ~1.0), plus its own distance from the fp32 one
        assert rel_l2(a[k], og64[k]) <= 1.5 * own + TOL_ALL[k], "%s: HIP vs fp64 oracle %.3e, fp32 oracle vs fp64 oracle %.3e" % (
            k, rel_l2(a[k], og64[k]), own)


def test_c5_graph_replayed_train_step_matches_eager():
    """One full train step (deformation, rasterizer fwd/bwd at 1 M surfels / 1600x1600, losses, Adam) replayed from the
    captured graph against the same step launched eagerly: same loss, same update."""
    import bench
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    P, H, W = bench.WORKLOADS["c5"]
    res = {}
    for graph in (False, True):
        tr = bench.build_trainer(P, H, W, dev, n_views=4, n_targets=1)
        try:
            if graph:
                tr.enable_graph(capacity=24 * P)
            losses = [float(tr.step()) for _ in range(2)]
            torch.cuda.synchronize()
            assert not _C.read_overflow()
        finally:
            _C.set_capacity(0)
        res[graph] = (losses, tr.surfels._xyz.detach().cpu().clone(), tr.surfels._opacity.detach().cpu().clone())
        del tr
        torch.cuda.empty_cache()
    (le, xe, oe), (lg, xg, og) = res[False], res[True]
    for a, b in zip(le, lg):
        assert np.isfinite(a) and abs(a - b) <= 1e-4 * abs(a), (le, lg)
    assert torch.isfinite(xg).all() and float((xe - xg).abs().median()) < 1e-6 and float((oe - og).abs().median()) < 1e-6

"""Contexts of the C ABI (include/dgs_surfel_rasterizer.h): the reference's entry points are stateless and re-entrant
(SURVEY.md 8b); whatever this library adds lives in a dgs_context.  Two threads with their own contexts, options and
streams interleave forward / backward calls on one device and must reproduce their single-threaded results -- the forward
(and the private lists) bit for bit; the backward accumulates with fp32 atomics whose order is not defined even within one
call, so its gradients are compared at the parity tolerance."""
import threading

import numpy as np
import pytest
import torch

from scene_utils import small_case

pytestmark = pytest.mark.gpu


def _run(case, ctx, gc, go, dev, iters):
    from diff_surfel_rasterization import _C
    e = torch.empty(0, device=dev)
    t = lambda k: case[k].to(dev).contiguous()
    H, W = case["image_height"], case["image_width"]
    out = []
    for _ in range(iters):
        R, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
            t("bg"), t("means3D"), e, t("opacities"), t("scales"), t("rotations"), 1.0, e, t("viewmatrix"), t("projmatrix"),
            case["tanfovx"], case["tanfovy"], H, W, t("shs"), case["sh_degree"], t("campos"), False, False, context=ctx)
        grads = _C.rasterize_gaussians_backward(
            t("bg"), t("means3D"), radii, e, t("scales"), t("rotations"), 1.0, e, t("viewmatrix"), t("projmatrix"), case["tanfovx"],
            case["tanfovy"], gc, go, t("shs"), case["sh_degree"], t("campos"), geom, R, binning, img, False, context=ctx)
        out.append((R, color.cpu().numpy(), allmap.cpu().numpy(), radii.cpu().numpy(), [g.cpu().numpy() for g in grads]))
    return out


def test_two_threads_two_contexts_interleaved():
    from diff_surfel_rasterization import _C
    from gpu_utils import rel_l2
    dev = torch.device("cuda:0")
    cases = [small_case(P=6000, H=160, W=144, seed=3, view=2, scale_mul=1.5),
             small_case(P=9000, H=112, W=208, seed=4, view=5, scale_mul=1.2, sh_degree=1)]
    cots = []
    for c in cases:
        g = torch.Generator().manual_seed(7)
        cots.append((torch.randn(3, c["image_height"], c["image_width"], generator=g).to(dev),
                     torch.randn(8, c["image_height"], c["image_width"], generator=g).to(dev)))
    _C.set_capacity(0)            # (a test that ran before may have left the device's DEFAULT context in capacity mode: num_rendered would be the capacity)
    ctxs = [_C.Context(dev), _C.Context(dev)]
    ctxs[1].set_option(0, 0)      # context 1: the reference's square rectangles -> different private lists / num_rendered
    ctxs[1].set_option(1, 0)      # ... and row-major tile order
    try:
        base = [_run(cases[i], ctxs[i], cots[i][0], cots[i][1], dev, 1)[0] for i in range(2)]
        assert base[0][0] > 0 and base[1][0] > 0
        # the options really are per context: the default context (tight rectangles) lists fewer entries for case 1
        assert _run(cases[1], None, cots[1][0], cots[1][1], dev, 1)[0][0] < base[1][0]
        results, errors = [None, None], []

        def worker(i):
            try:
                with torch.cuda.stream(torch.cuda.Stream(dev)):
                    results[i] = _run(cases[i], ctxs[i], cots[i][0], cots[i][1], dev, 12)
                    torch.cuda.current_stream().synchronize()
            except Exception as ex:   # surfaces in the main thread
                errors.append(ex)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for i in range(2):
            R0, c0, a0, r0, g0 = base[i]
            for R, c, a, r, g in results[i]:
                assert R == R0 and np.array_equal(c, c0) and np.array_equal(a, a0) and np.array_equal(r, r0)
                for x, y in zip(g, g0):
                    if y.size and np.abs(y).max() > 0:
                        assert rel_l2(x, y) <= 1e-6
    finally:
        for c in ctxs:
            c.close()


def test_context_overflow_flags_are_separate():
    """Capacity mode on one context does not leak into another: each has its own capacity and overflow flag."""
    from diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    case = small_case(P=3000, H=96, W=80, seed=9, view=4)
    gc, go = torch.zeros(3, 96, 80, device=dev), torch.zeros(8, 96, 80, device=dev)
    small, exact = _C.Context(dev), _C.Context(dev)
    try:
        small.set_option(2, 100)   # far too small: renders background, raises ITS flag
        a = _run(case, small, gc, go, dev, 1)[0]
        b = _run(case, exact, gc, go, dev, 1)[0]
        torch.cuda.synchronize()
        assert small.read_overflow() and not exact.read_overflow() and not small.read_overflow()
        assert np.abs(a[1] - case["bg"].numpy()[:, None, None]).max() == 0.0
        assert b[0] > 100 and np.abs(b[1]).max() > 0
    finally:
        small.close()
        exact.close()

"""-m gpu: the fused SSIM and KNN kernels (csrc/train_ops.hip) against the PyTorch formulations they replace
(which tests/ verify on the CPU against the reference's formula / brute force)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 64, 80), (3, 37, 53), (1, 16, 16), (3, 800, 800)])
def test_fused_ssim_matches_torch(shape):
    from dgs_amd.losses import ssim, ssim_torch
    g = torch.Generator().manual_seed(0)
    a = torch.rand(*shape, generator=g).cuda().requires_grad_(True)
    b = (a.detach() * 0.7 + 0.3 * torch.rand(*shape, generator=g).cuda()).clamp(0, 1)
    v = ssim(a, b)
    (3.0 * v).backward()
    ga = a.grad.clone()
    a.grad = None
    vr = ssim_torch(a, b)
    (3.0 * vr).backward()
    assert abs(float(v) - float(vr)) < 2e-6
    assert torch.allclose(ga, a.grad, rtol=2e-4, atol=2e-9 + 1e-4 * float(a.grad.abs().max()))


@pytest.mark.parametrize("N,M,D,K", [(1000, 64, 11, 3), (200000, 1024, 11, 3), (513, 1500, 3, 4), (77, 5, 16, 1)])
def test_knn_kernel_matches_bruteforce(N, M, D, K):
    from dgs_amd import _ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, D, generator=g).cuda()
    n = torch.randn(M, D, generator=g).cuda()
    idx = _ops.knn_indices(x, n, K)
    dsel = ((x[:, None, :] - n[idx]) ** 2).sum(-1)
    ref = torch.cat([torch.topk(((x[s:s + 8192, None, :] - n[None]) ** 2).sum(-1), K, dim=-1, largest=False).values
                     for s in range(0, N, 8192)])
    assert torch.allclose(dsel, ref, rtol=1e-5, atol=1e-6)          # same K nearest distances, ascending
    assert bool((dsel[:, 1:] >= dsel[:, :-1] - 1e-6).all())

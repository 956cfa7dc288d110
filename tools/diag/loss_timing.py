"""Dev tool (GPU box): the loss kernels alone, eager, HIP events around 200 back-to-back launches each.
usage: [DGS_TRAIN_OPS_LIB=variant.so] python tools/diag/loss_timing.py [H]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-2dgs_amd"))
import torch

from dgs_amd import _ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dev = torch.device("cuda:0")
lib = _ops.load()
g = torch.Generator(device=dev).manual_seed(0)
img = torch.rand(3, H, H, device=dev, generator=g)
gt = (img + 0.1 * torch.randn(3, H, H, device=dev, generator=g)).clamp(0, 1)
allmap = torch.rand(8, H, H, device=dev, generator=g) + 0.5
rays_d = torch.randn(H, H, 3, device=dev, generator=g)
rays_o = torch.randn(3, device=dev, generator=g)
wvt = torch.eye(4, device=dev)
nb, nr = int(lib.dgs_photo_blocks(3, H, H)), int(lib.dgs_regloss_blocks(H, H))
part = torch.empty(2 * nb + nr, device=dev)
part2 = torch.empty(int(lib.dgs_regloss_fused_blocks(H, H)), device=dev)
maps = torch.empty(3, 3, H, H, device=dev)
loss = torch.empty(1, device=dev)
unit = torch.ones(1, device=dev)
g_img = torch.empty_like(img)
g_all = torch.zeros_like(allmap)
st = _ops._stream(dev)
calls = {
    "ssim_fwd": lambda: lib.dgs_photo_forward(3, H, H, img.data_ptr(), gt.data_ptr(), part.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), None, st),
    "regloss_fwd": lambda: lib.dgs_regloss_forward_partials_z(H, H, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(), 0.02, 1000.0, part.data_ptr() + 8 * nb, None, g_all[5].data_ptr(), st),
    "combine": lambda: lib.dgs_loss_combine(part.data_ptr(), nb, part.data_ptr() + 8 * nb, nr, 3 * H * H, 0.2, loss.data_ptr(), st),
    "ssim_bwd": lambda: lib.dgs_photo_backward(3, H, H, img.data_ptr(), gt.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), 0.2, unit.data_ptr(), g_img.data_ptr(), None, st),
    "regloss_fused": lambda: lib.dgs_regloss_fused(H, H, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(), 0.02, 1000.0, part2.data_ptr(), g_all.data_ptr(), None, st),
    "regloss_bwd": lambda: lib.dgs_regloss_backward_slot(H, H, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(), wvt.data_ptr(), 0.02, 1000.0, unit.data_ptr(), g_all.data_ptr(), None, 1, st),
}
out = []
for name, fn in calls.items():
    for _ in range(20):
        assert fn() == 0, lib.dgs_train_ops_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out.append("%s %.1f" % (name, e0.elapsed_time(e1) * 1000 / 200))
print("us/launch:", "  ".join(out), " loss", float(loss))

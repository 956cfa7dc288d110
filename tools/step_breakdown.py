"""Phase-level GPU time breakdown of one train step (development aid, GPU only)."""
import os, sys, time
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dgs_amd.losses import training_loss
from dgs_amd.render import render
dev = torch.device("cuda:0")
P, H, W = bench.WORKLOADS[os.environ.get("WL", "metric")]
tr = bench.build_trainer(P, H, W, dev)
if os.environ.get('MLP_GRAPH', '1') == '1':
    tr.deform.enable_graphs(tr.deform.expand_time(tr.cameras[0].fid))
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
names = ["zero", "deform_fwd", "render_fwd", "loss_fwd", "backward", "stats+allreduce", "adam"]
acc = {n: 0.0 for n in names}
wall = 0.0
N = 10
for it in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    s, d = tr.surfels, tr.deform
    cam = tr.cameras[tr.view_for(tr.iteration)]
    gt = tr.targets[0]
    t0 = time.perf_counter()
    ev[0].record(); tr.bucket.zero()
    ev[1].record(); dv = d(s.get_xyz.detach(), d.expand_time(cam.fid), s.feature, s.motion_mask)
    ev[2].record(); pkg = render(cam, s, tr.bg, dv['d_xyz'], dv['d_rotation'], dv['d_scaling'])
    ev[3].record(); loss = training_loss(pkg, gt)
    ev[4].record(); loss.backward()
    ev[5].record()
    with torch.no_grad():
        vis = pkg["visibility_filter"]
        g2 = pkg["viewspace_points"].grad[:, :2].norm(dim=-1)
        tr.bucket.extra[:tr.P].copy_(torch.where(vis, g2, torch.zeros_like(g2)))
        tr.bucket.extra[tr.P:].copy_(vis.to(torch.float32))
        tr.bucket.all_reduce_mean()
    ev[6].record()
    tr.opt_surfels.step(); (tr.opt_deform.step() if tr.opt_deform is not None else None)
    ev[7].record()
    torch.cuda.synchronize()
    wall += time.perf_counter() - t0
    tr.iteration += 1
    for i, n in enumerate(names):
        acc[n] += ev[i].elapsed_time(ev[i + 1])
print("wall ms/step %.2f" % (wall / N * 1e3))
for n in names:
    print("%-18s %.3f ms" % (n, acc[n] / N))

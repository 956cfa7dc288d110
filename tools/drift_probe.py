"""Why a long run of the bench step is slower per step than a short one: the scene drifts.  Adam moves every parameter by ~lr per
step whatever the gradient (noise targets), log-scales at 0.005 per step: after 100 steps the splats are up to e^0.5 larger and the
tile lists longer.  Prints ms/step per block of 30 steps next to the mean screen radius and the visible count."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
P, H, W = bench.WORKLOADS["metric"]
tr = bench.build_trainer(P, H, W, dev)
tr.enable_graph(capacity=24 * P)
for _ in range(5):
    tr.step()
for blk in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30 * 1e3
    r = tr._radii[:tr.P].float()
    s = torch.exp(tr.surfels._scaling.detach()).mean()
    print("steps %3d-%3d: %.4f ms/step   mean screen radius of the last view %.2f px (visible %d)   mean scale %.5f" % (
        5 + 30 * blk, 35 + 30 * blk, dt, float(r[r > 0].mean()), int((r > 0).sum()), float(s)), flush=True)

"""Chase the intermittent 47 ms/step + empty-render state of the graph-mode bench (development aid, GPU only)."""
import os, sys, time
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from diff_surfel_rasterization import _C
dev = torch.device("cuda:0")
import dgs_amd.losses as L
L.DEBUG_TERMS = {} if os.environ.get('TERMS', '0') == '1' else None
P, H, W = bench.WORKLOADS["metric"]
tr = bench.build_trainer(P, H, W, dev)
def stats(tag):
    s = tr.surfels
    fl = torch.cat([p.detach().reshape(-1) for p in tr.bucket.params])
    print(tag, "finite", bool(torch.isfinite(fl).all()), "absmax %.3e" % float(fl.abs().max()), "scale mean %.4f max %.3e" % (float(s.get_scaling.mean()), float(s.get_scaling.max())),
          "grad absmax %.3e finite %s" % (float(tr.bucket.flat.abs().max()), bool(torch.isfinite(tr.bucket.flat).all())), "overflow", _C.read_overflow(reset=False), flush=True)
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "feature"] + ["d%d" % i for i in range(len(tr.bucket.params) - 7)]
    big = [(n, float(torch.nan_to_num(p.grad, 0, 0, 0).abs().max()), int((~torch.isfinite(p.grad)).sum()), int((p.grad.abs() > 1e3).sum())) for n, p in zip(names, tr.bucket.params)]
    print("   ", " ".join("%s:%.1e/%d/%d" % b for b in big if b[1] > 1e2 or b[2] > 0), flush=True)
    ex = tr.bucket.extra
    print("    extra max %.2e nonfinite %d" % (float(torch.nan_to_num(ex, 0, 0, 0).abs().max()), int((~torch.isfinite(ex)).sum())), flush=True)
stats("init")
if os.environ.get("DGS_NO_GRAPHS", "0") != "1":
    tr.enable_graph(capacity=24 * P)
stats("after enable_graph")
for i in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    l = tr.step()
    torch.cuda.synchronize()
    print("step", i, "loss %.6f" % float(l), "ms %.2f" % ((time.perf_counter() - t) * 1e3), " ".join("%s=%.5f" % (k, float(v)) for k, v in (L.DEBUG_TERMS or {}).items()), flush=True)
    if i in (0, 3, 7) or os.environ.get('EVERY', '0') == '1':
        stats("  after step %d" % i)

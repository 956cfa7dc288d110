import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
x = torch.randn(4096, 4096, device="cuda")
def work():
    y = x
    for _ in range(10):
        y = (y @ x).clamp_(-1, 1)
    return y
for ext in (False, True):
    try:
        kw = {"external": True} if ext else {}
        e1, e2 = torch.cuda.Event(enable_timing=True, **kw), torch.cuda.Event(enable_timing=True, **kw)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            work()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            e1.record(); out = work(); e2.record()
        for i in range(3):
            g.replay(); torch.cuda.synchronize()
            print("external", ext, "replay", i, "elapsed ms", e1.elapsed_time(e2))
    except Exception as ex:
        print("external", ext, "FAILED:", repr(ex)[:300])
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); work(); b.record(); torch.cuda.synchronize(); print("eager ms", a.elapsed_time(b))

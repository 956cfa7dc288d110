"""Is DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 still needed?  

ROCm 7.2 replays captured graphs through pre-recorded AQL packets by default; rounds 1-2 saw that path run a zero-fill node of the
train step out of order now and then (an L1 term of exactly 0, or 1e18 gradients), and every entry point has set the knob since.  The
replayed step no longer contains a memset / fill node.  This probe replays the metric step N times WITH WHATEVER THE ENVIRONMENT SAYS
(run it once with the knob unset and DGS_ALLOW_GRAPH_PACKET_CAPTURE=1, once with the knob 0) and looks for the two symptoms:
  * the first K replays against their eager twin from the same snapshot (tight: the trajectories have not separated yet),
  * over all N replays: every loss finite, and every step's loss within `jump` of the loss the same view had 64 steps earlier (the
    noise targets change a view's loss by ~1e-3 per revisit; a dropped term or exploded parameters move it by O(1)).
Prints one JSON line.   usage: python tools/diag/graph_knob_probe.py [N=2000] [K=40] [P=200000] [HW=800]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
import torch  # noqa: E402

import bench  # noqa: E402  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 by default: the caller's explicit setting wins)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
    HW = int(sys.argv[4]) if len(sys.argv) > 4 else 800
    from dgs_amd.train import Trainer
    Trainer.GUARD_RING = max(Trainer.GUARD_RING, 2 * N + 64)
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(P, HW, HW, dev)
    tr.enable_graph(capacity=24 * P)
    for _ in range(5):
        tr.step()
    snap, it0 = tr._snapshot(), tr.iteration
    rec0 = tr.overflow_recoveries
    for _ in range(N):
        tr.step()
    losses = tr.loss_history(N)
    recaptures = tr.overflow_recoveries - rec0
    finite = all(l == l and abs(l) < 1e6 for l in losses)
    jumps = [abs(losses[i] - losses[i - 64]) / max(abs(losses[i - 64]), 1e-12) for i in range(64, N)]
    # eager twin of the first K steps
    tr._restore(snap)
    tr.iteration = it0
    if getattr(tr, "_oflag", None) is not None:
        tr._oflag.zero_()
    from diff_surfel_rasterization import _C
    tr._graph = None
    _C.set_capacity(0)
    _C.set_option(6, 0)
    eager = [float(tr.step()) for _ in range(K)]
    rel = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses[:K], eager)]
    out = {"replays": N, "packet_capture_knob": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "unset"),
           "all_finite": finite, "max_same_view_jump": max(jumps) if jumps else 0.0,
           "steps_with_jump_over_5pct": sum(j > 0.05 for j in jumps), "recaptures": recaptures,
           "twin_steps": K, "twin_max_rel": max(rel), "twin_max_rel_first_3": max(rel[:3]),
           "loss_first_last": [losses[0], losses[-1]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

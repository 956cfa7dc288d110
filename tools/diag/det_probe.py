"""Where do two deterministic pre-fits of the `trained` workload part ways?  Runs fit(deterministic=True) twice with the bench's settings
and compares, every `every` iterations, bit-pattern checksums of the parameter tensors, the Adam moments, the densification statistics
and the loss.   usage: python tools/diag/det_probe.py [iterations=1200] [every=25] [P=100000] [HW=800]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-2dgs_amd")]
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402


def main():
    its = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    HW = int(sys.argv[4]) if len(sys.argv) > 4 else 800
    from dgs_amd.fit import fit
    from dgs_amd.synthetic import DynamicTruth, write_dynamic_dnerf
    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="dgs_det_")
    write_dynamic_dnerf(os.path.join(tmp, "scene"), n_train=48, n_test=2, H=HW, W=HW, device=dev, truth=DynamicTruth(24000, 16000, detail=0.3))
    cs = lambda t: int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())
    logs = []
    for run in range(2):
        log = []

        def hook(it, tr, log=log):
            if it % every:
                return
            s, d = tr.surfels, tr.deform
            m = tr.opt_surfels
            rec = {"it": it, "n": int(s.num_surfels), "xyz": cs(s._xyz), "sh": cs(s._features), "opacity": cs(s._opacity), "scaling": cs(s._scaling),
                   "rotation": cs(s._rotation), "feature": cs(s.feature), "nodes": cs(d.nodes), "net": sum(cs(p) for p in d.network.parameters()),
                   "grad_bucket": cs(tr.bucket.flat), "accum": cs(s.xyz_gradient_accum), "denom": cs(s.denom), "radii": cs(s.max_radii2D),
                   "exp_avg": cs(m.exp_avg), "recoveries": tr.overflow_recoveries, "knn": str(getattr(d, "knn_refine_mode", ""))}
            log.append(rec)
        warm, reg = 3 * its // 10, 8 * its // 10
        tr, losses = fit(os.path.join(tmp, "scene"), os.path.join(tmp, "model%d" % run), iterations=its, device=dev, num_pts=P, node_num=512, seed=0,
                         warm_up=warm, regularize_from=reg, node_densify_at=10 ** 9, deterministic=True, on_iteration=hook)
        tr.set_deterministic(False)
        logs.append((log, losses))
        del tr
        torch.cuda.empty_cache()
    shutil.rmtree(tmp, ignore_errors=True)
    (la, lossa), (lb, lossb) = logs
    first_loss = next((i for i, (x, y) in enumerate(zip(lossa, lossb)) if x != y), None)
    print("first differing loss at iteration", None if first_loss is None else first_loss + 1,
          "" if first_loss is None else (lossa[first_loss], lossb[first_loss]))
    for ra, rb in zip(la, lb):
        diff = [k for k in ra if ra[k] != rb[k]]
        if diff:
            print("first differing checkpoint: iteration", ra["it"], "fields", diff)
            print(" a:", {k: ra[k] for k in ("n", "recoveries", "knn")}, " b:", {k: rb[k] for k in ("n", "recoveries", "knn")})
            break
    else:
        print("all %d checkpoints identical" % len(la))


if __name__ == "__main__":
    main()

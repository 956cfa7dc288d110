import os, sys, subprocess, json
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, os.path.join(ROOT, "dynamic-2dgs_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from gpu_utils import run_hip
    from scene_utils import small_case
    case = small_case(P=2000, H=96, W=112, seed=2, view=3, scale_mul=1.5)
    g = np.random.default_rng(1)
    gc, go = g.standard_normal((3, 96, 112)).astype(np.float32), g.standard_normal((8, 96, 112)).astype(np.float32)
    hip = run_hip(case, gc, go, debug=False)
    np.savez(sys.argv[2], **{k: v for k, v in hip.items() if v is not None})
else:
    csrc = os.path.join(ROOT, "dynamic-2dgs_amd", "csrc")
    outs = {}
    for name in ("libdgs_surfel_rasterizer.so", "libdgs_surfel_rasterizer_precise.so"):
        f = "/tmp/fp_%s.npz" % name
        subprocess.check_call([sys.executable, __file__, "child", f], env=dict(os.environ, DGS_SURFEL_LIB=os.path.join(csrc, name)))
        outs[name] = np.load(f)
    a, b = outs["libdgs_surfel_rasterizer.so"], outs["libdgs_surfel_rasterizer_precise.so"]
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(k, "max %.3e  n>1e-5: %d  argmax %s  vals %s %s" % (d.max(), (d > 1e-5).sum(), np.unravel_index(d.argmax(), d.shape), a[k].flat[d.argmax()], b[k].flat[d.argmax()]))

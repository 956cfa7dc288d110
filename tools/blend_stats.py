"""How much of the blend kernels' traversal is useful (CPU, host-compiled surfel_math.h + the oracle's lists)?"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dynamic-2dgs_amd")]
from scene_utils import oracle_from_case, small_case
hmdir = os.path.join(ROOT, "tests", "hostmath")
subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(hmdir, "hostmath.cpp"), "-o", os.path.join(hmdir, "libhostmath.so")])
hm = ctypes.CDLL(os.path.join(hmdir, "libhostmath.so"))
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
f32 = lambda t: np.ascontiguousarray(t.numpy().astype(np.float32))
P, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
case = small_case(P=P, H=H, W=W, seed=0, view=0, scale_mul=float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
orc = oracle_from_case(case)
m3, sc, rot, op, sh = (f32(case[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs"))
vm, cp = f32(case["viewmatrix"]).reshape(-1), f32(case["campos"])
radii = np.zeros(P, np.int32); rec = np.zeros((P, 24), np.float32); tiles = np.zeros(P, np.int32); rects = np.zeros((P, 2), np.uint32)
hm.hm_preprocess(P, case["sh_degree"], sh.shape[1], p(m3), p(sc), p(rot), p(op), p(sh), None, p(vm), p(cp), W, H,
                 ctypes.c_float(case["tanfovx"]), ctypes.c_float(case["tanfovy"]), p(radii), p(rec), p(tiles), p(rects), 1)
ranges, plist = orc.field("ranges"), orc.field("point_list")
print("mean radius %.1f px, R(ref lists) %d" % (radii[radii > 0].mean(), orc.num_rendered))
for shape, name in ((0, "16x4 strips, box"), (2, "8x8 quadrants, box"), (1, "8x8 quadrants, box + conic (the kernels)")):
  hm.hm_set_shape(shape)
  out = np.zeros(16)
  hm.hm_blend_stats(W, H, p(np.ascontiguousarray(ranges)), p(np.ascontiguousarray(plist)), p(rec), p(out))
  S, sp, sa, pp, ppass, pb, sba, vis, visany, visb, vispb, missed = out[:12]
  print("---- wave shape", name, " S", S)
  print("strip pairs (alive) %d ; visited by mask %d (%.1f%%) ; of visited: any pixel passes %.1f%%, some live pixel blends %.1f%% ; missed %d"
      % (sp, vis, 100 * vis / sp, 100 * visany / vis, 100 * visb / vis, missed))
  print("lanes blending per blending (strip,entry): %.1f of 64 ; blending (strip,entry) pairs %d" % (vispb / max(visb, 1), visb))
  if g_shape_is_quadrant := (shape >= 1):
      print("of the 4 4x4 sub-blocks of a visited quadrant: %.2f hold a blending pixel (%.2f a passing one); of the 2 8x4 halves: %.2f"
            % (out[12] / max(vis, 1), out[14] / max(vis, 1), out[13] / max(vis, 1)))
# ---- row-per-block kernels (round 4): iterations a wave would run if each of its four 16-lane rows walked its own 4x4 block's list
for chunk, linear in ((64, 0), (64, 1), (48, 1)):
    out = np.zeros(8)
    hm.hm_row_stats(W, H, chunk, linear, p(np.ascontiguousarray(ranges)), p(np.ascontiguousarray(plist)), p(rec), p(out))
    v, s_, f, b, px, pairs, missed = out[:7]
    print("---- rows, %d entries per chunk, %s block test: visits of the quadrant kernel %d ; row iterations %d (%.3f of the visits), with free-running rows %.3f, "
          "perfectly balanced %.3f ; %.2f blocks per visit ; %.1f of 16 lanes blend per (entry, block) ; dropped pairs with a passing pixel: %d"
          % (chunk, "tangent-plane" if linear else "exact", v, s_, s_ / v, f / v, b / v, pairs / v, px / pairs, missed))


"""Generates tests/golden/node_pretrain_golden.npz by running the REFERENCE's node pre-training step -- the body of
GUI.train_node_rendering_step (train_gui.py:441-599), compiled from the reference's file in the build container -- on the CPU for a
complete (shortened) stage: warm-up, densification, opacity reset, the regularisers, the 'samp_hyper' node sampling, the final hand-over
of the node positions.

What is the reference's own code here: the step itself, ControlNodeWarp / DeformNetwork (utils/time_utils.py), DeformModel
(scene/deform_model.py), GaussianModel / StandardGaussianModel incl. their density control and optimiser surgery
(scene/gaussian_model.py), render() (gaussian_renderer/__init__.py), l1_loss / ssim (utils/loss_utils.py), the learning-rate schedules.
Stand-ins (absent packages, as in the other make_* scripts): pytorch3d.ops.knn_points (published semantics), simple_knn distCUDA2
(brute force), the rasterizer = the repo's CPU oracle operator; `.cuda()` / device="cuda" are redirected to the CPU.  `train_gui.py`
as a module needs a GUI toolkit, so only the one method is compiled from it (its source text is executed here, never stored).

Every random draw the reference makes (random.randint for the view, torch.rand in the regularisers, torch.normal in the split,
torch.randint in the farthest-point sampling) is recorded in call order: the test replays them through dgs_amd.node_pretrain.Draws'
interface, so a different ORDER or SHAPE of draws in the restatement fails the test.
Run from the repo root:  python tests/golden/make_node_pretrain_golden.py
"""
import ast
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dynamic-2dgs_amd")):
    sys.path.insert(0, p)

from make_deform_golden import fill_params, import_reference  # noqa: E402
from make_densify_golden import cuda_to_cpu  # noqa: E402
from make_init_golden import dist2_bruteforce  # noqa: E402

# the shortened schedule (the reference's defaults: 2000 / 7500 / 10000, interval 100, reset 3000)
# rotation_lr = 0 in the two tight cases: the node surfels are isotropic, so a rotation about their own normal changes nothing that is
# rendered -- its gradient is rounding noise, Adam turns noise into +-lr steps, and the next split places its children along the
# rotated axes: two correct implementations then differ by percents (so would two runs of the reference with its float atomics).
# With the rotations frozen every quantity of the stage is well conditioned and is compared tightly; "default" keeps the reference's
# rate and is compared loosely.
CASES = {
    "split": dict(node_num=24, n_points=160, views=6, S=40, node_warm_up=4, sampling_at=10, iterations=15, densify_interval=3,
                  opacity_reset_interval=5, densify_grad_threshold=0.0095, extent=4.0, seed=3, width=32, rotation_lr=0.0, opacity_lr=0.05),
    "clone": dict(node_num=16, n_points=120, views=5, S=36, node_warm_up=3, sampling_at=9, iterations=13, densify_interval=4,
                  opacity_reset_interval=6, densify_grad_threshold=0.008, extent=60.0, seed=8, width=32, rotation_lr=0.0, opacity_lr=1.0),
    "masks": dict(node_num=16, n_points=120, views=5, S=36, node_warm_up=3, sampling_at=7, iterations=10, densify_interval=4,
                  opacity_reset_interval=6, densify_grad_threshold=0.002, extent=60.0, seed=8, width=32, rotation_lr=0.0, opacity_lr=0.05, masks=True),
    "default": dict(node_num=24, n_points=160, views=6, S=40, node_warm_up=4, sampling_at=10, iterations=15, densify_interval=3,
                    opacity_reset_interval=5, densify_grad_threshold=0.0095, extent=4.0, seed=3, width=32, rotation_lr=0.002, opacity_lr=0.05),
}


def scene_inputs(case):
    """Cameras, target images and the initial point cloud -- deterministic formulas shared with the test."""
    from dgs_amd.cameras import orbit_cameras
    c = CASES[case]
    cams = orbit_cameras(c["views"], c["S"], c["S"])
    g = torch.Generator().manual_seed(c["seed"])
    pts = (torch.rand(c["n_points"], 3, generator=g) * 2 - 1) * 0.7
    S = c["S"]
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    targets = []
    for k, cam in enumerate(cams):
        f = float(cam.fid)
        img = torch.zeros(3, S, S)
        for b in range(3):   # three blobs that move with time
            cx = S * (0.3 + 0.2 * b + 0.15 * f)
            cy = S * (0.65 - 0.18 * b + 0.1 * f * (b - 1))
            blob = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * (0.09 * S) ** 2))
            img += blob[None] * torch.tensor([0.9 - 0.3 * b, 0.3 + 0.25 * b, 0.5])[:, None, None]
        targets.append(img.clamp(0, 1))
    return cams, targets, pts


def reference_method(name):
    """One method of class GUI, compiled from train_gui.py with its line numbers (the module itself imports a GUI toolkit)."""
    path = os.path.join(REF, "train_gui.py")
    tree = ast.parse(open(path).read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "GUI":
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == name:
                    code = compile(ast.Module(body=[f], type_ignores=[]), path, "exec")
                    return code
    raise KeyError(name)


def load_by_file(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def import_reference_stack():
    import diff_surfel_rasterization as product
    import oracle_raster_op
    tu = import_reference()   # utils.time_utils with the pytorch3d stand-in; patches .cuda()
    # pytorch3d's knn_points returns a NAMED tuple: time_utils unpacks it, deform_utils reads .dists / .idx
    import collections
    import utils.deform_utils as du
    from make_deform_golden import knn_stub
    KNN = collections.namedtuple("KNN", "dists idx knn")
    named = lambda p1, p2, l1=None, l2=None, K=1, **kw: KNN(*knn_stub(p1, p2, K=K, return_nn=kw.get("return_nn", False)))
    sys.modules["pytorch3d.ops"].knn_points = named
    if hasattr(du, "knn_points"):
        du.knn_points = named
    fake = types.ModuleType("diff_surfel_rasterization")
    fake.GaussianRasterizationSettings = product.GaussianRasterizationSettings
    fake.GaussianRasterizer = oracle_raster_op.OracleRasterizer
    sys.modules["diff_surfel_rasterization"] = fake
    for nm in ("cv2", "matplotlib", "matplotlib.pyplot", "plyfile", "simple_knn", "simple_knn._C"):
        sys.modules.setdefault(nm, types.ModuleType(nm))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = dist2_bruteforce
    # the `scene` package pulls in the dataset readers (imageio ...): its two modules are loaded by file under their real names
    scene = types.ModuleType("scene")
    scene.__path__ = []
    sys.modules["scene"] = scene
    gm = load_by_file("scene.gaussian_model", "scene/gaussian_model.py")
    scene.gaussian_model = gm
    dm = load_by_file("scene.deform_model", "scene/deform_model.py")
    import gaussian_renderer as ref_renderer
    import utils.loss_utils as lu
    return tu, gm, dm, ref_renderer, lu


class Recorder:
    """Wraps the four random sources; every draw lands in `log` as (kind, array)."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.log = []
        self._rand, self._randn, self._randint = torch.rand, torch.randn, torch.randint

    def randint_py(self, a, b):
        assert a == 0
        v = int(self._randint(0, b + 1, (1,), generator=self.g))
        self.log.append(("pick", np.array([v, b + 1])))
        return v

    def rand(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            shape = tuple(shape[0])
        v = self._rand(*shape, generator=self.g) if shape else self._rand([], generator=self.g)
        self.log.append(("rand", v.numpy().copy()))
        return v

    def normal(self, mean, std):
        z = self._randn(mean.shape, generator=self.g)
        self.log.append(("randn", z.numpy().copy()))
        return mean + std * z

    def randint(self, low, high, size, **kw):
        assert low == 0 and tuple(size) == (1,)
        v = self._randint(0, high, (1,), generator=self.g)
        self.log.append(("start", np.array([int(v), high])))
        return v


def run(case):
    c = CASES[case]
    tu, gm, dm, ref_renderer, lu = import_reference_stack()
    cams, targets, pts = scene_inputs(case)
    masks = alpha_masks(targets) if c.get("masks") else [None] * len(cams)
    views = [SimpleNamespace(**cam._asdict(), original_image=targets[k], gt_alpha_mask=masks[k], image_name="v%d" % k, flow_dirs=[])
             for k, cam in enumerate(cams)]
    opt = SimpleNamespace(
        progressive_train_node=False, progressive_stage_steps=3000, progressive_stage_ratio=0.2, node_warm_up=c["node_warm_up"],
        iterations_node_sampling=c["sampling_at"], iterations_node_rendering=c["iterations"], lambda_dssim=0.2, no_arap_loss=False,
        gt_alpha_mask_as_scene_mask=bool(c.get("masks")), gt_alpha_mask_as_dynamic_mask=False, densification_interval=c["densify_interval"],
        opacity_reset_interval=c["opacity_reset_interval"], densify_grad_threshold=c["densify_grad_threshold"], densify_from_iter=500,
        node_max_num_ratio_during_init=16, deform_downsamp_strategy="samp_hyper", deform_downsamp_with_dynamic_mask=False,
        percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30_000,
        deform_lr_max_steps=40_000, feature_lr=0.004, opacity_lr=c["opacity_lr"], scaling_lr=0.002, rotation_lr=c["rotation_lr"], deform_lr_scale=1.0)
    rec = Recorder(c["seed"] + 100)
    saved = (torch.rand, torch.normal, torch.randint, torch.Tensor.to)
    rand_like = torch.rand_like
    torch.rand, torch.normal, torch.randint = rec.rand, rec.normal, rec.randint
    torch.rand_like = lambda t, **k: rec.rand(*t.shape)        # the random background of render() (gaussian_renderer/__init__.py:58)
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else saved[3](self, *a, **k)
    losses, counts = [], []
    backward = torch.Tensor.backward

    def recording_backward(self, *a, **k):
        losses.append(float(self.detach()))
        return backward(self, *a, **k)
    out = {}
    try:
        with cuda_to_cpu():
            torch.manual_seed(0)
            # the reference's own network class at a smaller width (a constructor argument ControlNodeWarp does not pass on): with
            # 256-wide layers a few of the 523 k weights see gradients of rounding-noise size, Adam turns each into a full +-lr step,
            # and two correct implementations drift apart by percents within ten iterations -- a comparison that pins nothing
            init = tu.DeformNetwork.__init__
            defaults = init.__defaults__
            assert init.__code__.co_varnames[1:3] == ("D", "W") and defaults[:2] == (8, 256)
            init.__defaults__ = (8, c["width"]) + defaults[2:]
            deform = dm.DeformModel(K=3, deform_type="node", is_blender=True, skinning=False, hyper_dim=8, node_num=c["node_num"],
                                    pred_opacity=False, pred_color=False, use_hash=False, hash_time=False, d_rot_as_res=True,
                                    local_frame=True, progressive_brand_time=False, with_arap_loss=True, max_d_scale=-1,
                                    enable_densify_prune=False, is_scene_static=False)
            init.__defaults__ = defaults
            assert deform.deform.network.W == c["width"]
            fill_params(deform.deform.network)
            with torch.no_grad():   # a visible motion (the default heads start at 1e-5)
                deform.deform.network.gaussian_warp.weight.mul_(4.0)
            out["net0"] = {k: v.detach().numpy().copy() for k, v in deform.deform.network.named_parameters()}
            deform.train_setting(opt)                                        # GUI.__init__, train_gui.py:148
            deform.deform.init(init_pcl=pts.clone(), force_init=True, opt=opt, as_gs_force_with_motion_mask=False, force_gs_keep_all=False)
            out["nodes0"] = deform.deform.nodes.detach().numpy().copy()
            out["gs_scaling0"] = deform.deform.as_gaussians._scaling.detach().numpy().copy()
            gui = SimpleNamespace(
                viewpoint_stack=None, opt=opt, iteration_node_rendering=1, deform=deform, pipe=SimpleNamespace(
                    debug=False, compute_cov3D_python=False, convert_SHs_python=False, depth_ratio=1.0),
                dataset=SimpleNamespace(load2gpu_on_the_fly=False, is_blender=True, white_background=False),
                scene=SimpleNamespace(getTrainCameras=lambda: list(views), cameras_extent=c["extent"]),
                background=torch.zeros(3), gaussians=SimpleNamespace(feature=torch.full((pts.shape[0], 8), -1e-2)), smooth_term=None)
            # what every density-control call did: rows added by the clone and by the split, rows removed by the final prune
            calls, cur = [], {}
            post, prune, dap = gm.GaussianModel.densification_postfix, gm.GaussianModel.prune_points, gm.GaussianModel.densify_and_prune

            def postfix(self, new_xyz, *a, **k):
                cur.setdefault("added", []).append(int(new_xyz.shape[0]))
                return post(self, new_xyz, *a, **k)

            def prune_points(self, mask):
                cur.setdefault("pruned", []).append(int(mask.sum()))
                return prune(self, mask)

            def densify_and_prune(self, *a, **k):
                cur.clear()
                r = dap(self, *a, **k)
                n_clone, n_children = cur["added"]
                parents, pruned = cur["pruned"]
                assert n_children == 2 * parents
                calls.append((gui.iteration_node_rendering, n_clone, parents, pruned, int(self.get_xyz.shape[0])))
                return r
            gm.GaussianModel.densification_postfix, gm.GaussianModel.prune_points, gm.GaussianModel.densify_and_prune = postfix, prune_points, densify_and_prune
            ns = dict(randint=rec.randint_py, torch=torch, np=np, render=ref_renderer.render, l1_loss=lu.l1_loss, ssim=lu.ssim,
                      GaussianModel=gm.GaussianModel, int=int, min=min, max=max, sorted=sorted, len=len)
            exec(reference_method("train_node_rendering_step"), ns)
            step = ns["train_node_rendering_step"]
            torch.Tensor.backward = recording_backward
            per_it = []
            while gui.iteration_node_rendering < opt.iterations_node_rendering:   # GUI.train, train_gui.py:207-213
                it = gui.iteration_node_rendering
                step(gui)
                gs = deform.deform.as_gaussians
                per_it.append((it, gs.get_xyz.shape[0], float(gs.get_scaling[0, 0]), float(gs.get_xyz.detach().abs().sum()),
                               float(deform.deform.nodes.detach()[:, :3].abs().sum())))
            torch.Tensor.backward = backward
            gm.GaussianModel.densification_postfix, gm.GaussianModel.prune_points, gm.GaussianModel.densify_and_prune = post, prune, dap
            out["calls"] = np.array(calls)
            gs = deform.deform.as_gaussians
            out["final"] = dict(
                nodes=deform.deform.nodes.detach().numpy().copy(), node_radius=deform.deform._node_radius.detach().numpy().copy(),
                node_weight=deform.deform._node_weight.detach().numpy().copy(), gs_xyz=gs._xyz.detach().numpy().copy(),
                gs_opacity=gs._opacity.detach().numpy().copy(), gs_scaling=gs._scaling.detach().numpy().copy(),
                gs_f_dc=gs._features_dc.detach().numpy().copy(), gs_rotation=gs._rotation.detach().numpy().copy(),
                warp_w=deform.deform.network.gaussian_warp.weight.detach().numpy().copy(),
                lin0_w=deform.deform.network.linear[0].weight.detach().numpy().copy(),
                lr_deform=np.array([g_["lr"] for g_ in deform.optimizer.param_groups]),
                lr_gs=np.array([g_["lr"] for g_ in gs.optimizer.param_groups]))
    finally:
        torch.rand, torch.normal, torch.randint, torch.Tensor.to = saved
        torch.rand_like = rand_like
        torch.Tensor.backward = backward
    return losses, per_it, rec.log, out


def alpha_masks(targets):
    """Ground-truth masks [1,H,W] of the views: where the target's blobs are, with a soft edge."""
    return [(t.sum(0, keepdim=True) * 4.0).clamp(0, 1) for t in targets]


def main():
    arrays = {}
    for case in CASES:
        losses, per_it, log, out = run(case)
        per_it = np.array(per_it, dtype=np.float64)
        print(case, "losses", ["%.5f" % v for v in losses])
        print(case, "node surfels per iteration", per_it[:, 1].astype(int).tolist())
        assert len(losses) == CASES[case]["iterations"] - 1
        assert per_it[:, 1].max() > CASES[case]["node_num"], "the density control never grew the node surfels"
        assert per_it[CASES[case]["sampling_at"] - 2, 1] >= CASES[case]["node_num"], "fewer node surfels than control nodes at the sampling"
        arrays[case + "_losses"] = np.array(losses)
        arrays[case + "_per_it"] = per_it
        arrays[case + "_draw_kinds"] = np.array([k for k, _ in log])
        for i, (_, v) in enumerate(log):
            arrays["%s_draw_%03d" % (case, i)] = np.asarray(v)
        for k, v in out["final"].items():
            arrays["%s_final_%s" % (case, k)] = v
        arrays[case + "_density_calls"] = out["calls"]
        print(case, "density control (iteration, cloned, split, pruned, rows):", out["calls"].tolist())
        arrays[case + "_nodes0"] = out["nodes0"]
        arrays[case + "_gs_scaling0"] = out["gs_scaling0"]
        kinds = [k for k, _ in log]
        print(case, "draws:", {k: kinds.count(k) for k in sorted(set(kinds))})
    np.savez_compressed(os.path.join(HERE, "node_pretrain_golden.npz"), **arrays)


if __name__ == "__main__":
    main()

"""Data formats of SURVEY 8 f4 (dgs_amd/io.py).  The D-NeRF reader is pinned by what the imported reference read from the
tiny dataset under tests/golden/dnerf_tiny (make_dnerf_golden.py), deform.pth by the reference's state_dict layout
(make_io_golden.py).  The reference writes PLY through `plyfile`, which is absent here: the PLY layout is restated from the
attribute list in gaussian_model.py:229-256 and checked as a format (header text, byte layout, round trips)."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from dgs_amd import io as dio
from dgs_amd.deform import ControlNodes
from dgs_amd.model import SurfelModel
from dgs_amd.synthetic import make_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_dnerf_reader_matches_reference():
    g = np.load(os.path.join(GOLD, "dnerf_golden.npz"))
    for tag, white in (("black", False), ("white", True)):
        for split in ("train", "test"):
            frames = dio.read_transforms(os.path.join(GOLD, "dnerf_tiny"), "transforms_%s.json" % split, white_background=white)
            k = "%s_%s_" % (tag, split)
            assert [f.name for f in frames] == list(g[k + "name"])
            np.testing.assert_allclose(np.stack([f.R for f in frames]), g[k + "R"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(np.stack([f.T for f in frames]), g[k + "T"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(np.array([[f.camera.FoVx, f.camera.FoVy] for f in frames]), g[k + "fov"], rtol=1e-12)
            assert np.array_equal(np.array([float(f.camera.fid) for f in frames], np.float32), g[k + "fid"])
            for name, mine in (("wvt", "world_view_transform"), ("full", "full_proj_transform"), ("center", "camera_center")):
                got = np.stack([getattr(f.camera, mine).numpy() for f in frames])
                np.testing.assert_allclose(got, g[k + name], rtol=1e-6, atol=1e-7)
            assert np.array_equal(np.stack([f.image.numpy() for f in frames]), g[k + "image"])     # 8-bit values / 255: exact
            assert np.array_equal(np.stack([f.alpha.numpy() for f in frames]), g[k + "alpha"])
            if split == "train":
                norm = dio.scene_normalization(frames)
                np.testing.assert_allclose(norm["radius"], g[k + "radius"], rtol=1e-6)
                np.testing.assert_allclose(norm["translate"], g[k + "translate"], rtol=1e-6, atol=1e-7)
    assert frames[0].camera.image_height == 10 and frames[0].image.shape == (3, 10, 10)


def test_load_dnerf_creates_the_initial_point_cloud(tmp_path):
    root = tmp_path / "scene"
    shutil.copytree(os.path.join(GOLD, "dnerf_tiny"), root)
    d = dio.load_dnerf(str(root), num_pts=500)
    assert len(d["train"]) == 4 and len(d["test"]) == 2 and os.path.exists(root / "points3d.ply")
    pc = d["point_cloud"]
    assert pc.points.shape == (500, 3) and np.abs(pc.points).max() <= 1.3 and pc.colors.min() >= 0.49 and pc.colors.max() <= 0.51
    d2 = dio.load_dnerf(str(root), eval=False)      # second call reads the stored cloud back; eval=False merges the splits
    assert np.array_equal(d2["point_cloud"].points, pc.points) and len(d2["train"]) == 6
    scene = dio.scene_from_point_cloud(pc.points, pc.colors)
    assert scene.xyz.shape == (500, 3) and scene.f_rest.shape == (500, 15, 3) and scene.log_scale.shape == (500, 2)
    # scale = sqrt(mean squared distance to the 3 nearest neighbours), brute force
    x = torch.tensor(pc.points, dtype=torch.float32)
    d2m = torch.cdist(x, x).pow(2)
    d2m.fill_diagonal_(float("inf"))
    want = d2m.topk(3, largest=False).values.mean(1)
    assert torch.allclose(torch.exp(scene.log_scale[:, 0]) ** 2, want, rtol=1e-3)
    assert torch.allclose(torch.sigmoid(scene.opacity_logit), torch.full((500, 1), 0.1))


def test_surfel_ply_layout_and_round_trip(tmp_path):
    scene = make_scene(37, seed=4)
    model = SurfelModel(scene, capacity=50)        # dead slots are not saved
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    dio.save_surfels(model, path)
    raw = open(path, "rb").read()
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] + ["opacity"] + \
            ["scale_0", "scale_1"] + ["rot_%d" % i for i in range(4)] + ["fea_%d" % i for i in range(8)]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 37\n" + "".join("property float %s\n" % n for n in names) + "end_header\n"
    assert raw.startswith(header.encode()) and len(raw) == len(header) + 37 * 4 * len(names)
    table = np.frombuffer(raw[len(header):], "<f4").reshape(37, len(names))
    assert np.array_equal(table[:, 0:3], scene.xyz.numpy()) and not table[:, 3:6].any()
    # SH blocks are channel-major: f_rest_k = coefficient (k % 15) + 1 of channel k // 15
    assert np.array_equal(table[:, 6:9], scene.f_dc[:, 0, :].numpy())
    assert np.array_equal(table[:, 9 + 15 * 1 + 4], scene.f_rest[:, 4, 1].numpy())
    assert np.array_equal(table[:, 54], scene.opacity_logit[:, 0].numpy()) and np.array_equal(table[:, 55:57], scene.log_scale.numpy())
    back = dio.load_surfels(path)
    for a, b in zip(back, scene):
        assert torch.equal(a, b)
    # the same file through the packed-SH model
    packed = SurfelModel(scene, packed_sh=True)
    dio.save_surfels(packed, str(tmp_path / "p.ply"))
    assert open(str(tmp_path / "p.ply"), "rb").read() == raw


def test_ply_reader_accepts_ascii_and_big_endian(tmp_path):
    v = np.zeros(3, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"], v["red"], v["green"], v["blue"] = [1.5, -2, 3], [0, 0.25, 7], [9, 8, -7.5], [0, 128, 255], [1, 2, 3], [255, 0, 9]
    head = "ply\nformat %s 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n" \
           "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n"
    with open(tmp_path / "a.ply", "w") as f:
        f.write(head % "ascii")
        for r in v:
            f.write(" ".join(str(x) for x in r) + "\n")
    with open(tmp_path / "b.ply", "wb") as f:
        f.write((head % "binary_big_endian").encode())
        f.write(v.astype([("x", ">f4"), ("y", ">f4"), ("z", ">f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")]).tobytes())
    for name in ("a.ply", "b.ply"):
        got = dio.read_ply(str(tmp_path / name))
        for n in v.dtype.names:
            assert np.array_equal(got[n], v[n]), (name, n)
    dio.store_point_cloud(str(tmp_path / "c.ply"), np.stack([v["x"], v["y"], v["z"]], 1), np.stack([v["red"], v["green"], v["blue"]], 1))
    pc = dio.fetch_point_cloud(str(tmp_path / "c.ply"))
    assert np.array_equal(pc.points[:, 2], v["z"]) and np.allclose(pc.colors[:, 0], v["red"] / 255.0) and not pc.normals.any()


def test_deform_weights_file_is_the_reference_layout(tmp_path):
    keys = json.load(open(os.path.join(GOLD, "deform_state_keys.json")))
    deform = ControlNodes(node_num=48, K=3, hyper_dim=8, local_frame=True)
    path = dio.save_deform(deform, str(tmp_path), 3000)
    assert path.endswith(os.path.join("deform", "iteration_3000", "deform.pth"))
    saved = torch.load(path, weights_only=True)
    assert sorted(saved) == sorted(k for k, _, _ in keys)
    for k, shape, dtype in keys:
        assert list(saved[k].shape) == shape and str(saved[k].dtype) == dtype, k
    # a file as the reference writes it (its key order, a different node count) loads; the newest iteration is found
    gen = torch.Generator().manual_seed(0)
    ref_state = {}
    for k, shape, dtype in keys:
        shape = [64 if s == 48 else s for s in shape]
        ref_state[k] = torch.tensor(True) if dtype == "torch.bool" else torch.randn(*shape, generator=gen)
    os.makedirs(tmp_path / "deform" / "iteration_40000")
    torch.save(ref_state, tmp_path / "deform" / "iteration_40000" / "deform.pth")
    assert dio.load_deform(deform, str(tmp_path)) is True
    assert deform.nodes.shape == (64, 11) and torch.equal(deform.network.linear[5].weight, ref_state["network.linear.5.weight"])
    assert dio.load_deform(deform, str(tmp_path / "nothing_here")) is False
    # a file from a real training run also carries the node Gaussians (`gs_` + GaussianModel.param_names(),
    # utils/time_utils.py:867-872, scene/gaussian_model.py:77-78): skipped like the reference's strict=False load does for
    # everything it does not own -- the owned tensors still load
    full = dict(ref_state)
    for name, shape in (("_xyz", (64, 3)), ("_features_dc", (64, 1, 3)), ("_features_rest", (64, 15, 3)), ("_scaling", (64, 2)),
                        ("_rotation", (64, 4)), ("_opacity", (64, 1)), ("max_radii2D", (64,)), ("xyz_gradient_accum", (64, 1))):
        full["gs_" + name] = torch.randn(*shape, generator=gen)
    full["network.linear.5.weight"] = full["network.linear.5.weight"] + 1.0
    os.makedirs(tmp_path / "deform" / "iteration_41000")
    torch.save(full, tmp_path / "deform" / "iteration_41000" / "deform.pth")
    assert dio.load_deform(deform, str(tmp_path)) is True
    assert torch.equal(deform.network.linear[5].weight, full["network.linear.5.weight"])
    assert sorted(dio.load_deform.skipped_keys) == sorted(k for k in full if k.startswith("gs_"))
    # an own key missing from the file is an error, not a silent partial load
    del full["network.linear.5.weight"]
    os.makedirs(tmp_path / "deform" / "iteration_42000")
    torch.save(full, tmp_path / "deform" / "iteration_42000" / "deform.pth")
    with pytest.raises(KeyError):
        dio.load_deform(deform, str(tmp_path))
    # ... and so is an entry neither this model nor the reference's writer knows (only the `gs_*` family is skipped)
    odd = dict(ref_state, surprise=torch.zeros(3))
    os.makedirs(tmp_path / "deform" / "iteration_43000")
    torch.save(odd, tmp_path / "deform" / "iteration_43000" / "deform.pth")
    with pytest.raises(KeyError):
        dio.load_deform(deform, str(tmp_path))


def test_padding_nodes_are_stripped_on_save_and_restored_on_load(tmp_path):
    """The fused MLP kernels need a node count that is a multiple of 64: padding nodes (far outside the scene) are no part of a
    checkpoint, and a loader that wants the fused path gets them back (load_deform(pad_to=64))."""
    deform = ControlNodes(node_num=50, K=3, hyper_dim=8, local_frame=True)
    deform.nodes.data[:, :3] = torch.randn(50, 3)
    assert deform.pad_nodes(64) == 14 and deform.nodes.shape[0] == 64 and int(deform.live_nodes.sum()) == 50
    live_before = deform.nodes.detach()[deform.live_nodes].clone()
    dio.save_deform(deform, str(tmp_path), 100)
    saved = torch.load(tmp_path / "deform" / "iteration_100" / "deform.pth", weights_only=True)
    assert saved["nodes"].shape[0] == 50                                   # the file holds the model, not the padding
    fresh = ControlNodes(node_num=8, K=3, hyper_dim=8, local_frame=True)
    assert dio.load_deform(fresh, str(tmp_path)) and fresh.nodes.shape[0] == 50
    fresh = ControlNodes(node_num=8, K=3, hyper_dim=8, local_frame=True)
    assert dio.load_deform(fresh, str(tmp_path), pad_to=64) and fresh.nodes.shape[0] == 64 and int(fresh.live_nodes.sum()) == 50
    assert torch.equal(fresh.nodes.detach()[fresh.live_nodes], live_before)


def test_fit_tiny_scene_end_to_end_and_restore(tmp_path, monkeypatch):
    """Reader -> initialisation -> steps with densification and opacity reset -> checkpoint -> restore, on the CPU with the
    oracle operator serving the rasterizer call."""
    import dgs_amd.render as render_mod
    from dgs_amd import fit as fit_mod
    from oracle_raster_op import OracleRasterizer
    monkeypatch.setattr(render_mod, "GaussianRasterizer", OracleRasterizer)
    root = tmp_path / "scene"
    shutil.copytree(os.path.join(GOLD, "dnerf_tiny"), root)
    logs = []
    tr, losses = fit_mod.fit(str(root), str(tmp_path / "out"), iterations=7, device="cpu", densify_from=2, densify_interval=2,
                             opacity_reset_interval=6, densify_grad_threshold=1e-9, slots=260, node_num=16, num_pts=200,
                             rasterizer_cls=OracleRasterizer, log=logs.append, oneup_sh_degree_step=3)
    assert tr.surfels.active_sh_degree == 2           # iterations 3 and 6
    assert len(losses) == 7 and all(l == l for l in losses) and len(logs) == 2
    assert tr.surfels.num_surfels != 200 and tr.P >= 260
    surfels, deform = fit_mod.restore(str(tmp_path / "out"), node_num=16)
    alive = tr.surfels.alive
    assert surfels.get_xyz.shape[0] == tr.surfels.num_surfels
    assert torch.equal(surfels._xyz.detach(), tr.surfels._xyz.detach()[alive])
    assert torch.equal(surfels._features_rest.detach(), tr.surfels._features_rest.detach()[alive])
    assert torch.equal(deform.nodes.detach(), tr.deform.nodes.detach())


def test_scene_from_point_cloud_matches_reference_create_from_pcd():
    """Against GaussianModel.create_from_pcd of the imported reference (make_init_golden.py; its CUDA-only distCUDA2 replaced
    by the published 3-NN semantics)."""
    g = np.load(os.path.join(GOLD, "init_golden.npz"))
    scene = dio.scene_from_point_cloud(g["points"], g["colors"])
    for mine, key, tol in ((scene.xyz, "xyz", 0), (scene.f_dc, "f_dc", 1e-6), (scene.f_rest, "f_rest", 0), (scene.rotation, "rotation", 0),
                           (scene.opacity_logit, "opacity", 1e-6), (scene.feature, "feature", 0), (scene.log_scale, "scaling", 2e-5)):
        assert mine.shape == g[key].shape, key
        np.testing.assert_allclose(mine.numpy(), g[key], rtol=tol, atol=tol, err_msg=key)


def test_point_cloud_ply_fields_match_what_the_reference_hands_to_plyfile(tmp_path):
    """Reference-side half of the PLY pin (tests/golden/make_ply_golden.py): the structured array GaussianModel.save_ply gives
    to plyfile's PlyElement.describe -- 69 little-endian float32 fields, their names and ORDER, the channel-major SH flattening, zero
    normals -- is what dgs_amd.io.save_surfels writes, field for field and value for value; and what GaussianModel.load_ply rebuilds
    from such an element is what load_surfels returns.  (The byte layout of the file itself rests on the PLY specification:
    `plyfile` is not in the image.)"""
    g = np.load(os.path.join(GOLD, "ply_golden.npz"))
    from dgs_amd.synthetic import SurfelScene
    t = lambda n: torch.from_numpy(g["in_" + n].copy())
    scene = SurfelScene(t("xyz"), t("scaling"), t("rotation"), t("opacity"), t("f_dc"), t("f_rest"), t("feature"))
    model = SurfelModel(scene)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    dio.save_surfels(model, path)
    v = dio.read_ply(path)
    assert list(v.dtype.names) == list(g["field_names"])                          # names and order
    assert [v.dtype[n].str for n in v.dtype.names] == list(g["field_dtypes"])     # '<f4' each
    mine = np.stack([np.asarray(v[n], np.float64) for n in v.dtype.names], axis=1)
    assert np.array_equal(mine, g["table"])
    # header text: what plyfile writes for PlyData([el]) with its defaults (binary, native = little endian), property per field
    with open(path, "rb") as f:
        head = f.read(4096).split(b"end_header\n")[0].decode("ascii").split("\n")
    assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0"
    assert [h for h in head if h.startswith("element")] == ["element vertex %d" % mine.shape[0]]
    assert [h.split()[1:] for h in head if h.startswith("property")] == [["float", n] for n in g["field_names"]]
    # the load side
    back = dio.load_surfels(path, sh_degree=3, fea_dim=8)
    for name, got in (("xyz", back.xyz), ("f_dc", back.f_dc), ("f_rest", back.f_rest), ("opacity", back.opacity_logit),
                      ("scaling", back.log_scale), ("rotation", back.rotation), ("feature", back.feature)):
        assert got.shape == g["loaded_" + name].shape, name
        assert np.array_equal(got.numpy(), g["loaded_" + name]), name
    assert int(g["loaded_active_sh_degree"]) == 3

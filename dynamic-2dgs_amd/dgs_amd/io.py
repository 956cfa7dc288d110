"""Data formats either side of the train step: the surfel checkpoint (PLY), the deformation weights
(deform.pth), the initial point cloud (points3d.ply) and the D-NeRF / Blender reader.

File layouts follow the reference so that checkpoints and datasets move between the two code bases:
  * point_cloud.ply: binary little-endian PLY, one float32 property per attribute in the order of
    GaussianModel.construct_list_of_attributes / save_ply (scene/gaussian_model.py:229-256): x y z nx ny nz f_dc_* f_rest_*
    opacity scale_* rot_* fea_*, with the SH blocks stored CHANNEL-major (`transpose(1, 2).flatten(1)`), values
    pre-activation; load_ply (:263-306) is the inverse.
  * deform/iteration_N/deform.pth: torch.save of the ControlNodeWarp state_dict (scene/deform_model.py:41-56); the key names
    of dgs_amd.deform.ControlNodes are the reference's, plus its boolean `inited` buffer.
  * points3d.ply: x y z nx ny nz (float32) red green blue (uint8) (scene/dataset_readers.py:173-198).
  * transforms_{train,test}.json + RGBA frames: readCamerasFromTransforms / readNerfSyntheticInfo (:272-403), loadCam
    (utils/camera_utils.py:22-63) and Camera (scene/cameras.py:18-59).
The reference reads and writes PLY through the `plyfile` package; this module carries its own reader/writer for the
subset of the format those files use (one `vertex` element, scalar properties).
"""
import json
import os
from typing import NamedTuple

import numpy as np
import torch

from .cameras import Camera, projection_matrix, world_to_view
from .synthetic import SurfelScene

# ---- PLY ----------------------------------------------------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
_PLY_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def write_ply(path, vertex):
    """vertex: numpy structured array (scalar fields) -> binary_little_endian PLY with one `vertex` element."""
    vertex = np.asarray(vertex)
    fields = [(n, vertex.dtype[n].str.lstrip("<|=")) for n in vertex.dtype.names]
    head = ["ply", "format binary_little_endian 1.0", "element vertex %d" % vertex.shape[0]]
    head += ["property %s %s" % (_PLY_NAMES[t], n) for n, t in fields]
    head.append("end_header")
    packed = np.empty(vertex.shape[0], dtype=[(n, "<" + t) for n, t in fields])
    for n, _ in fields:
        packed[n] = vertex[n]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(packed.tobytes())


def read_ply(path):
    """The `vertex` element of a PLY file (ascii, binary little- or big-endian; scalar properties) as a structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    if cur["name"] == "vertex" or cur is elements[0]:
                        raise ValueError("%s: list properties in the vertex element are not supported" % path)
                    continue
                cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements or elements[0]["name"] != "vertex":
            raise ValueError("%s: the first element must be `vertex`" % path)
        el = elements[0]
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=el["count"], ndmin=2, dtype=np.float64)
            out = np.empty(el["count"], dtype=[(n, t) for n, t in el["props"]])
            for i, (n, _) in enumerate(el["props"]):
                out[n] = rows[:, i]
            return out
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        dt = np.dtype([(n, order + t) for n, t in el["props"]])
        raw = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
        return raw.astype([(n, t) for n, t in el["props"]])


# ---- surfel checkpoint --------------------------------------------------------------------------------------------------
def surfel_attribute_names(n_dc=3, n_rest=45, n_scale=2, n_rot=4, fea_dim=8):
    """construct_list_of_attributes (gaussian_model.py:229-243)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += ["f_dc_%d" % i for i in range(n_dc)] + ["f_rest_%d" % i for i in range(n_rest)] + ["opacity"]
    names += ["scale_%d" % i for i in range(n_scale)] + ["rot_%d" % i for i in range(n_rot)] + ["fea_%d" % i for i in range(fea_dim)]
    return names


@torch.no_grad()
def save_surfels(model, path):
    """GaussianModel.save_ply (gaussian_model.py:245-256) for the live slots of a SurfelModel."""
    alive = model.alive.cpu() if hasattr(model, "alive") else slice(None)
    c = lambda t: t.detach().cpu()[alive]
    xyz = c(model._xyz).numpy()
    f_dc = c(model._features_dc).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    f_rest = c(model._features_rest).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, c(model._opacity).numpy(), c(model._scaling).numpy(), c(model._rotation).numpy(),
            c(model.feature).numpy()]
    table = np.concatenate(cols, axis=1).astype(np.float32)
    names = surfel_attribute_names(f_dc.shape[1], f_rest.shape[1], cols[5].shape[1], cols[6].shape[1], cols[7].shape[1])
    assert len(names) == table.shape[1]
    vertex = np.empty(table.shape[0], dtype=[(n, "f4") for n in names])
    for i, n in enumerate(names):
        vertex[n] = table[:, i]
    write_ply(path, vertex)


def load_surfels(path, sh_degree=3, fea_dim=None) -> SurfelScene:
    """GaussianModel.load_ply (gaussian_model.py:263-306): the pre-activation parameters as a SurfelScene
    (feed it to SurfelModel(scene, capacity=...)).  fea_dim: the model's feature width (the reference zero-fills up to it: 8 hyper
    coordinates, 9 with a motion mask); None = as many `fea_*` properties as the file holds, at least 8."""
    v = read_ply(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], np.float32)
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    f_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[:, :, None]          # [P,3,1]
    rest = [n for n in names if n.startswith("f_rest_")]
    if len(rest) != 3 * (sh_degree + 1) ** 2 - 3:
        raise ValueError("%s holds %d f_rest_* properties, SH degree %d needs %d" % (path, len(rest), sh_degree, 3 * (sh_degree + 1) ** 2 - 3))
    f_rest = np.stack([col(n) for n in rest], axis=1).reshape(xyz.shape[0], 3, (sh_degree + 1) ** 2 - 1)
    scales = np.stack([col(n) for n in names if n.startswith("scale_")], axis=1)
    rots = np.stack([col(n) for n in names if n.startswith("rot")], axis=1)
    fea_names = [n for n in names if n.startswith("fea")]
    if fea_dim is None:
        fea_dim = max(8, len(fea_names))
    feas = np.zeros((xyz.shape[0], fea_dim), np.float32)
    for i, n in enumerate(fea_names):
        feas[:, i] = col(n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
    return SurfelScene(t(xyz), t(scales), t(rots), t(col("opacity")[:, None]), t(f_dc).transpose(1, 2).contiguous(),
                       t(f_rest).transpose(1, 2).contiguous(), t(feas))


# ---- deformation weights ------------------------------------------------------------------------------------------------
def search_for_max_iteration(folder):
    """utils/system_utils.py:29-33."""
    if not os.path.exists(folder):
        return None
    iters = [int(name.split("_")[-1]) for name in os.listdir(folder) if "_" in name]
    return max(iters) if iters else None


def save_deform(deform, model_path, iteration):
    """DeformModel.save_weights (scene/deform_model.py:41-44)."""
    out = os.path.join(model_path, "deform/iteration_{}".format(iteration))
    os.makedirs(out, exist_ok=True)
    state = {k: v.detach().cpu() for k, v in deform.state_dict().items()}
    live = getattr(deform, "live_nodes", None)
    if live is not None and not bool(live.all()):   # padding rows of the fused kernels are not part of the model
        keep = live.cpu()
        for k in ("nodes", "_node_radius", "_node_weight"):
            state[k] = state[k][keep].contiguous()
    state.setdefault("inited", torch.tensor(True))   # buffer of the reference's ControlNodeWarp
    torch.save(state, os.path.join(out, "deform.pth"))
    return os.path.join(out, "deform.pth")


def load_deform(deform, model_path, iteration=-1, pad_to=1):
    """DeformModel.load_weights (scene/deform_model.py:46-56): False if there is nothing to load.  pad_to = 64 re-creates the
    padding nodes save_deform strips (the fused MLP kernels need a node count that is a multiple of 64; without them a resumed
    run would silently fall back to the PyTorch formulation)."""
    it = search_for_max_iteration(os.path.join(model_path, "deform")) if iteration == -1 else iteration
    path = os.path.join(model_path, "deform/iteration_{}/deform.pth".format(it))
    if not os.path.exists(path):
        return False
    state = torch.load(path, map_location="cpu", weights_only=True)
    own = deform.state_dict()
    # The reference's ControlNodeWarp.load_state_dict (utils/time_utils.py:845-865) takes the node tensors by name, hands the
    # `gs_*` entries (the nodes' own GaussianModel, written by its state_dict :867-872 once train_setting has run -- every file
    # of a real training run has them) to `as_gaussians`, and loads the rest with strict=False.  The node Gaussians are only
    # used by the reference's node-rendering warm-up, which is not part of this path: they and any other key this model does
    # not own are skipped; an own key the file lacks is an error (a silently half-loaded network would be worse).
    skipped = [k for k in state if k not in own and k != "inited"]
    foreign = [k for k in skipped if not k.startswith("gs_")]
    if foreign:
        raise KeyError("deform.pth holds entries this model does not know and the reference does not write: %s" % foreign)
    missing = [k for k in own if k not in state]
    if missing:
        raise KeyError("deform.pth lacks %s (file has %d entries, %d of them not owned by this model)" % (missing, len(state), len(skipped)))
    if state["nodes"].shape != own["nodes"].shape:   # node densification changes the node count: adopt the file's
        with torch.no_grad():
            for k in ("nodes", "_node_radius", "_node_weight"):
                getattr(deform, k).data = torch.empty_like(state[k], device=own[k].device)
    deform.load_state_dict({k: v for k, v in state.items() if k in own})
    load_deform.skipped_keys = skipped
    if pad_to > 1 and hasattr(deform, "pad_nodes"):
        deform.pad_nodes(pad_to)
    return True


# ---- initial point cloud ------------------------------------------------------------------------------------------------
class PointCloud(NamedTuple):
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


def store_point_cloud(path, xyz, rgb):
    """storePly (scene/dataset_readers.py:183-198): rgb in 0..255."""
    xyz = np.asarray(xyz)
    v = np.empty(xyz.shape[0], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                                      ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["nx"] = v["ny"] = v["nz"] = 0
    rgb = np.asarray(rgb)
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    write_ply(path, v)


def fetch_point_cloud(path) -> PointCloud:
    """fetchPly (scene/dataset_readers.py:173-180)."""
    v = read_ply(path)
    return PointCloud(np.vstack([v["x"], v["y"], v["z"]]).T, np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0,
                      np.vstack([v["nx"], v["ny"], v["nz"]]).T)


# ---- D-NeRF / Blender reader --------------------------------------------------------------------------------------------
class Frame(NamedTuple):
    camera: Camera
    image: torch.Tensor        # [3,H,W] in [0,1], composited over the background
    alpha: torch.Tensor        # [1,H,W]
    name: str
    R: np.ndarray
    T: np.ndarray


def _resized_planes(rgba_u8, resolution):
    """PILtoTorch (utils/general_utils.py:23-37): colour and alpha resized separately, /255."""
    from PIL import Image
    rgb = np.asarray(Image.fromarray(rgba_u8[..., :3]).resize(resolution))
    a = np.asarray(Image.fromarray(rgba_u8[..., 3]).resize(resolution))
    return (torch.from_numpy(np.concatenate([rgb, a[..., None]], axis=-1)) / 255.0).permute(2, 0, 1)


def read_transforms(path, transforms_file, white_background=False, extension=".png", no_bg=True, resolution=1, znear=0.01, zfar=100.0):
    """readCamerasFromTransforms + loadCam + Camera for one split; frames sorted by the number in their file name."""
    from PIL import Image
    with open(os.path.join(path, transforms_file)) as f:
        contents = json.load(f)
    fovx = contents["camera_angle_x"]
    frames = sorted(contents["frames"], key=lambda fr: int(os.path.basename(fr["file_path"]).split(".")[0].split("_")[-1]))
    out = []
    for idx, fr in enumerate(frames):
        fp = fr["file_path"]
        cam_name = os.path.join(path, fp if fp.endswith(("jpg", "png")) else fp + extension)
        fid = fr["time"] if "time" in fr else idx / len(frames)
        rgba_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.join(path, fp))), "rgba")
        if os.path.exists(rgba_dir):
            cam_name = os.path.join(rgba_dir, os.path.basename(fp)).replace(".jpg", ".png")
        m = np.linalg.inv(np.array(fr["transform_matrix"]))
        R = -np.transpose(m[:3, :3])
        R[:, 0] = -R[:, 0]
        T = -m[:3, 3]
        norm = np.array(Image.open(cam_name).convert("RGBA")) / 255.0
        bg = np.array([1, 1, 1]) if white_background else np.array([0, 0, 0])
        if no_bg:
            norm[:, :, :3] = norm[:, :, 3:4] * norm[:, :, :3] + bg * (1 - norm[:, :, 3:4])
        # the reader re-quantises through an 8-bit image (dataset_readers.py:316): truncation, not rounding
        rgba = np.floor(norm * 255.0).astype(np.uint8)
        h, w = rgba.shape[:2]
        focal = w / (2 * np.tan(fovx / 2))
        fovy = 2 * np.arctan(h / (2 * focal))
        FovY, FovX = fovx, fovy     # (sic) the reference's naming; identical for square frames
        res = (round(w / resolution), round(h / resolution))
        planes = _resized_planes(rgba, res)
        image, alpha = planes[:3].clamp(0.0, 1.0).float(), planes[3:4].float()
        wvt = torch.tensor(world_to_view(R, T)).transpose(0, 1).contiguous()
        proj = projection_matrix(znear, zfar, FovX, FovY).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0).contiguous()
        cam = Camera(int(image.shape[1]), int(image.shape[2]), float(FovX), float(FovY), wvt, full, wvt.inverse()[3, :3].contiguous(),
                     torch.tensor([float(fid)], dtype=torch.float32))
        out.append(Frame(cam, image, alpha, os.path.splitext(os.path.basename(cam_name))[0], R, T))
    return out


def scene_normalization(frames):
    """getNerfppNorm (scene/dataset_readers.py:79-113, apply=False): translate = -mean camera centre, radius = largest
    distance of a camera from it (the `cameras_extent` the densification thresholds scale with)."""
    centers = []
    for fr in frames:
        c2w = np.linalg.inv(world_to_view(fr.R, fr.T))
        centers.append(c2w[:3, 3:4])
    centers = np.hstack(centers)
    center = np.mean(centers, axis=1, keepdims=True)
    radius = np.max(np.linalg.norm(centers - center, axis=0, keepdims=True))
    return {"translate": -center.flatten(), "radius": radius}


def load_dnerf(path, white_background=False, eval=True, extension=".png", resolution=1, num_pts=100_000, seed=0):
    """readNerfSyntheticInfo (scene/dataset_readers.py:325-403): train / test frames, normalisation, and the initial point
    cloud (points3d.ply; created with `num_pts` uniform points in [-1.3, 1.3]^3 when the dataset has none)."""
    train = read_transforms(path, "transforms_train.json", white_background, extension, resolution=resolution)
    test = []
    if os.path.exists(os.path.join(path, "transforms_test.json")):
        test = read_transforms(path, "transforms_test.json", white_background, extension, resolution=resolution)
    if not eval:
        train = train + test
    ply_path = os.path.join(path, "points3d.ply")
    if not os.path.exists(ply_path):
        rng = np.random.RandomState(seed)
        xyz = rng.random_sample((num_pts, 3)) * 2.6 - 1.3
        shs = rng.random_sample((num_pts, 3)) / 255.0
        store_point_cloud(ply_path, xyz, (shs * 0.28209479177387814 + 0.5) * 255)   # SH2RGB (utils/sh_utils.py)
    return {"train": train, "test": test, "normalization": scene_normalization(train), "point_cloud": fetch_point_cloud(ply_path),
            "ply_path": ply_path}


# ---- initialisation from a point cloud ------------------------------------------------------------------------------------
def mean_nn_dist2(points, k=3, chunk=1024):
    """simple_knn distCUDA2 (submodules/simple-knn/simple_knn.cu:148-183): mean SQUARED distance of every point to its k
    nearest other points (exact).  Init-time only.  Candidates come from the |q|^2 + |p|^2 - 2 q.p expansion (one [chunk, P]
    matrix, a GEMM); the k+8 best of them are re-evaluated with exact differences, so the expansion's cancellation error
    (~1e-7 |p|^2) can only matter if it reorders neighbours whose distances differ by less than that."""
    P = points.shape[0]
    kk = min(k, P - 1)
    cand = min(kk + 8, P - 1)
    out = torch.empty(P, dtype=points.dtype, device=points.device)
    n2 = (points * points).sum(-1)
    for a in range(0, P, chunk):
        q = points[a:a + chunk]
        d = n2[a:a + chunk, None] + n2[None, :] - 2.0 * (q @ points.t())
        d[torch.arange(q.shape[0], device=d.device), torch.arange(a, a + q.shape[0], device=d.device)] = float("inf")   # not itself
        idx = d.topk(cand, dim=1, largest=False).indices
        exact = (q[:, None, :] - points[idx]).pow(2).sum(-1)
        out[a:a + chunk] = exact.topk(kk, dim=1, largest=False).values.mean(dim=1)
    return out


def scene_from_point_cloud(points, colors, sh_degree=3, fea_dim=8, device=None) -> SurfelScene:
    """GaussianModel.create_from_pcd (scene/gaussian_model.py:143-179, with_motion_mask=False): DC colour from RGB, isotropic
    log-scales from the 3-NN spacing, identity rotations, opacity 0.1, hyper coordinates -1e-2.  device: where the neighbour search
    runs (the reference's distCUDA2 runs on the GPU as well; 100k points take ~40 s on the host, well under 1 s on the device); the
    scene comes back on the CPU either way."""
    pts = torch.as_tensor(np.asarray(points), dtype=torch.float32)
    rgb = torch.as_tensor(np.asarray(colors), dtype=torch.float32)
    P = pts.shape[0]
    f_dc = ((rgb - 0.5) / 0.28209479177387814)[:, None, :]             # RGB2SH
    f_rest = torch.zeros(P, (sh_degree + 1) ** 2 - 1, 3)
    dist2 = torch.clamp_min(mean_nn_dist2(pts.to(device) if device is not None else pts).cpu(), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 2)
    rots = torch.zeros(P, 4)
    rots[:, 0] = 1
    o = 0.1 * torch.ones(P, 1)
    return SurfelScene(pts, scales, rots, torch.log(o / (1 - o)), f_dc.contiguous(), f_rest, torch.full((P, fea_dim), -1e-2))

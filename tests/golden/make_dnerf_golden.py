"""Generates the tiny D-NeRF style dataset tests/golden/dnerf_tiny/ (two transforms_*.json + RGBA PNGs; data, made here
from seeded numpy) and tests/golden/dnerf_golden.npz = what the IMPORTED reference reads from it:
scene/dataset_readers.py readCamerasFromTransforms + getNerfppNorm, utils/camera_utils.py loadCam and scene/cameras.py
Camera (matrices, centre, fid, ground-truth image, alpha mask), for black and white backgrounds.

Absent packages that the exercised functions never call are stood in by empty modules (imageio, cv2, plyfile,
simple_knn); `scene/__init__.py` is bypassed by loading the files directly.
Run from the repo root:  python tests/golden/make_dnerf_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "dnerf_tiny")
W, H = 10, 10     # the reader's FovX/FovY naming swap only matters for non-square images; D-NeRF frames are square


def pose(theta, phi, radius):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "..", "dynamic-2dgs_amd"))
    from dgs_amd.cameras import pose_spherical
    return pose_spherical(theta, phi, radius)


def make_dataset():
    rng = np.random.default_rng(3)
    for split, n in (("train", 4), ("test", 2)):
        os.makedirs(os.path.join(DATA, split), exist_ok=True)
        frames = []
        order = list(range(n))[::-1]   # stored out of order: the reader sorts by the number in the file name
        for i in order:
            rgba = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
            rgba[:2, :, 3] = 0
            rgba[-2:, :, 3] = 255
            Image.fromarray(rgba, "RGBA").save(os.path.join(DATA, split, "r_%03d.png" % i))
            c2w = pose(-180.0 + 77.0 * i + (5.0 if split == "test" else 0.0), -30.0 + 3.0 * i, 4.0 - 0.1 * i)
            fr = {"file_path": "./%s/r_%03d" % (split, i), "rotation": 0.1, "transform_matrix": c2w.tolist()}
            if split == "train":
                fr["time"] = round(i / (n - 1), 6)     # the test split has no 'time': idx / len(frames) applies
            frames.append(fr)
        with open(os.path.join(DATA, "transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f, indent=1)


def import_reference():
    for name in ("imageio", "cv2", "plyfile", "simple_knn", "simple_knn._C"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.path.insert(0, REF)
    # a bare package object for `scene` so that its submodules load without scene/__init__.py (which needs a GPU stack)
    pkg = types.ModuleType("scene"); pkg.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = pkg
    import scene.dataset_readers as dr
    import utils.camera_utils as cu
    # The reader hands Image.fromarray an int8 array with an explicit mode (dataset_readers.py:316).  The Pillow releases
    # the reference was written against took the raw bytes in that case; Pillow 12 (this image) refuses the dtype.
    # Restore the raw-byte reading for the reference's call only.
    orig = Image.fromarray

    def fromarray(obj, mode=None):
        if getattr(obj, "dtype", None) == np.int8:
            obj = obj.view(np.uint8)
        return orig(obj, mode)
    dr.Image = types.SimpleNamespace(open=Image.open, fromarray=fromarray)
    return dr, cu


def main():
    make_dataset()
    dr, cu = import_reference()
    args = types.SimpleNamespace(resolution=1, data_device="cpu", load2gpu_on_the_fly=False)
    out = {}
    for tag, white in (("black", False), ("white", True)):
        for split in ("train", "test"):
            infos = dr.readCamerasFromTransforms(DATA, "transforms_%s.json" % split, white, ".png", no_bg=True)
            cams = cu.cameraList_from_camInfos(infos, 1.0, args)
            k = "%s_%s_" % (tag, split)
            out[k + "R"] = np.stack([c.R for c in infos])
            out[k + "T"] = np.stack([c.T for c in infos])
            out[k + "fov"] = np.array([[c.FovX, c.FovY] for c in infos])
            out[k + "fid"] = np.array([float(c.fid) for c in cams], np.float32)
            out[k + "name"] = np.array([c.image_name for c in infos])
            out[k + "wvt"] = np.stack([c.world_view_transform.numpy() for c in cams])
            out[k + "full"] = np.stack([c.full_proj_transform.numpy() for c in cams])
            out[k + "center"] = np.stack([c.camera_center.numpy() for c in cams])
            out[k + "image"] = np.stack([c.original_image.numpy() for c in cams])
            out[k + "alpha"] = np.stack([c.gt_alpha_mask.numpy() for c in cams])
            if split == "train":
                norm = dr.getNerfppNorm(infos)
                out[k + "radius"] = np.float64(norm["radius"])
                out[k + "translate"] = np.asarray(norm["translate"], np.float64)
    np.savez_compressed(os.path.join(HERE, "dnerf_golden.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.startswith("black_train")})


if __name__ == "__main__":
    main()

"""Generates tests/golden/ply_golden.npz: the reference-side half of the point_cloud.ply pin.

`plyfile` is not in the image (and cannot be installed), so the BYTES of a PLY file stay pinned by the PLY specification only.  What
the reference hands to / expects from that package is pinned here by importing scene/gaussian_model.py with a RECORDING stand-in:
  * save side: GaussianModel.save_ply (scene/gaussian_model.py:245-256) runs on seeded parameters; the stand-in's
    PlyElement.describe captures the structured array it is given -- field names, order, dtypes, values -- and the element name;
  * load side: GaussianModel.load_ply (:263-306) runs on a stand-in PlyData.read that serves that same structured array
    (`elements[0][name]`, `elements[0].properties[i].name`): what it reconstructs (shapes, the channel-major SH unflattening, the
    name-prefix scans) is stored as the expected result of dgs_amd.io.load_surfels.
tests/test_io.py checks dgs_amd.io.save_surfels / load_surfels against both.  Run from the repo root:
    python tests/golden/make_ply_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_densify_golden import REF, cuda_to_cpu  # noqa: E402

P, SEED = 37, 11
RECORD = {}


class _Prop:
    def __init__(self, name):
        self.name = name


class _Element:
    def __init__(self, data, name):
        self.data, self.name = data, name
        self.properties = [_Prop(n) for n in data.dtype.names]

    def __getitem__(self, key):
        return self.data[key]


class PlyElement:
    @staticmethod
    def describe(data, name, **kw):
        RECORD["describe_name"] = name
        RECORD["describe_kwargs"] = dict(kw)
        RECORD["vertex"] = np.array(data, copy=True)
        return _Element(RECORD["vertex"], name)


class PlyData:
    def __init__(self, elements=(), **kw):
        self.elements = list(elements)
        RECORD["plydata_kwargs"] = dict(kw)     # text= / byte_order= would select ascii / big-endian: the reference passes none

    def write(self, path):
        RECORD["write_path"] = path

    @staticmethod
    def read(path):
        RECORD["read_path"] = path
        return PlyData([_Element(RECORD["vertex"], "vertex")])


def import_reference_model():
    ply = types.ModuleType("plyfile")
    ply.PlyData, ply.PlyElement = PlyData, PlyElement
    knn = types.ModuleType("simple_knn"); knn_c = types.ModuleType("simple_knn._C"); knn_c.distCUDA2 = None
    sys.modules.update({"plyfile": ply, "simple_knn": knn, "simple_knn._C": knn_c})
    sys.path.insert(0, REF)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gaussian_model_ply", os.path.join(REF, "scene", "gaussian_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    m = import_reference_model()
    m.mkdir_p = lambda p: None     # save_ply creates the directory of its path first: nothing is written here
    g = torch.Generator().manual_seed(SEED)
    rn = lambda *s: torch.randn(*s, generator=g)
    init = {"xyz": rn(P, 3), "f_dc": rn(P, 1, 3), "f_rest": rn(P, 15, 3), "opacity": rn(P, 1), "scaling": rn(P, 2), "rotation": rn(P, 4),
            "feature": rn(P, 8)}
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
            "rotation": "_rotation", "feature": "feature"}
    gm = m.GaussianModel(3, fea_dim=8, with_motion_mask=False)
    for n, a in attr.items():
        setattr(gm, a, torch.nn.Parameter(init[n].clone()))
    gm.save_ply("/nonexistent/point_cloud/iteration_7/point_cloud.ply")
    v = RECORD["vertex"]
    assert RECORD["describe_name"] == "vertex" and not RECORD["describe_kwargs"] and not RECORD["plydata_kwargs"]
    out = {"in_" + n: t.numpy() for n, t in init.items()}
    out["field_names"] = np.array(list(v.dtype.names))
    out["field_dtypes"] = np.array([v.dtype[n].str for n in v.dtype.names])
    out["table"] = np.stack([np.asarray(v[n], np.float64) for n in v.dtype.names], axis=1)
    # load side
    gm2 = m.GaussianModel(3, fea_dim=8, with_motion_mask=False)
    with cuda_to_cpu():
        gm2.load_ply("/nonexistent/point_cloud/iteration_7/point_cloud.ply")
    for n, a in attr.items():
        out["loaded_" + n] = getattr(gm2, a).detach().numpy()
    out["loaded_active_sh_degree"] = np.array(gm2.active_sh_degree)
    np.savez_compressed(os.path.join(HERE, "ply_golden.npz"), **out)
    print("fields:", len(v.dtype.names), list(v.dtype.names)[:8], "...", "dtype", set(out["field_dtypes"].tolist()))
    for n in attr:
        print(n, out["loaded_" + n].shape, float(np.abs(out["loaded_" + n] - out["in_" + n]).max()))


if __name__ == "__main__":
    main()

"""Generates tests/golden/deform_state_keys.json: names, shapes and dtypes of the state_dict the IMPORTED reference's
ControlNodeWarp (utils/time_utils.py) saves into deform.pth (scene/deform_model.py:41-44), for node_num = 48.
Run from the repo root:  python tests/golden/make_io_golden.py
"""
import json
import os

from make_deform_golden import import_reference

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tu = import_reference()
    ref = tu.ControlNodeWarp(is_blender=True, node_num=48, K=3, hyper_dim=8, local_frame=True, d_rot_as_res=True,
                             with_arap_loss=False, with_node_weight=True)
    keys = [[k, list(v.shape), str(v.dtype)] for k, v in ref.state_dict().items()]
    with open(os.path.join(HERE, "deform_state_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    print(len(keys), "entries")


if __name__ == "__main__":
    main()

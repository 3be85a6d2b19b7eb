#!/usr/bin/env python
"""Compile the robot description files of the reference asset tree into flat ModelSpec JSON files.

Run in the development container (needs /root/reference/assets or --asset-root); the resulting
isaacgymenvs_amd/models/*.json are committed so that the GPU box (no /root/reference) can build the kernels.
Asset options mirror the reference task code (file:line cited per entry).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isaacgymenvs_amd.assets.model import load_asset  # noqa: E402

ENTRIES = {
    # reference cartpole.py:72-88 (fix_base_link=True)
    # the rail sits 2 m above the plane (cartpole.py:93) and self-collision is filtered (:107): no contact geometry
    "cartpole": dict(file="urdf/cartpole.urdf", fix_base_link=True, collide_body_filter=lambda n: False),
    # reference ant.py:139-157
    "ant": dict(file="mjcf/nv_ant.xml"),
    # reference humanoid.py:142-157
    "humanoid": dict(file="mjcf/nv_humanoid.xml"),
    # reference anymal_terrain.py:214-231: collapse_fixed_joints, replace_cylinder_with_capsule, density 0.001 (only used
    # for links without <inertial>), armature 0, fix_base_link False
    "anymal": dict(file="urdf/anymal_c/urdf/anymal_minimal.urdf", density=0.001, replace_cylinder_with_capsule=True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asset-root", default="/root/reference/assets")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isaacgymenvs_amd", "models"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for name, e in ENTRIES.items():
        kw = {k: v for k, v in e.items() if k != "file"}
        spec = load_asset(os.path.join(a.asset_root, e["file"]), name=name, **kw)
        spec.save(os.path.join(a.out, name + ".json"))
        print(f"{name}: nb={spec.nb} nd={spec.nd} nv={spec.nv} nsph={len(spec.sph_body)} mass={spec.total_mass():.4f}")


if __name__ == "__main__":
    main()

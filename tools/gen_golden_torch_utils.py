#!/usr/bin/env python
"""Golden vectors for isaacgymenvs_amd/utils/torch_jit_utils.py, produced by running the REFERENCE's own
isaacgymenvs/utils/torch_jit_utils.py (imported with tools/gen_golden.py's stubs).  Development container only (needs /root/reference);
writes tests/golden/torch_jit_utils.npz, which is committed.  Every case is `<function>__in<i>` inputs and `<function>__out<i>` outputs."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402


def unit(g, n):
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def cases(n=96, seed=5):
    """name -> tuple of input tensors (the same inputs feed the reference here and this repo's functions in the test)"""
    g = torch.Generator().manual_seed(seed)
    q, q2 = unit(g, n), unit(g, n)
    q[0] = torch.tensor([0.0, 0.0, 0.0, 1.0]); q[1] = torch.tensor([0.0, 0.0, 0.0, -1.0]); q[2] = torch.tensor([0.0, 2 ** -0.5, 0.0, 2 ** -0.5])   # identity, its double cover, pitch = +90 deg
    q2[3] = q[3]; q2[4] = -q[4]
    v, t = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
    ang = (torch.rand(n, generator=g) * 4 - 2) * np.pi
    lo = -torch.rand(7, generator=g) - 0.1
    up = torch.rand(7, generator=g) + 0.1
    x7 = torch.randn(n, 7, generator=g)
    pose = torch.cat([t, q2], dim=-1)
    em = torch.randn(n, 3, generator=g) * 1.5; em[0] = 0.0; em[1] = torch.tensor([1e-7, 0.0, 0.0])
    tt = torch.rand(n, 1, generator=g); tt[0] = 0.0; tt[1] = 1.0
    return {
        "quat_mul": (q, q2), "normalize": (torch.cat([v, torch.zeros(2, 3)]),), "quat_apply": (q, v), "quat_rotate": (q, v), "quat_rotate_inverse": (q, v),
        "quat_conjugate": (q,), "quat_unit": (q * 3.0,), "quat_from_angle_axis": (ang, v), "normalize_angle": (ang * 3,),
        "tf_inverse": (q, t), "tf_apply": (q, t, v), "tf_vector": (q, v), "tf_combine": (q, t, q2, v), "get_basis_vector": (q, v),
        "get_euler_xyz": (q,), "quat_from_euler_xyz": (ang, ang.flip(0) * 0.5, ang * 0.25), "tensor_clamp": (x7, lo.expand(n, 7), up.expand(n, 7)),
        "scale": (x7, lo, up), "unscale": (x7, lo, up), "scale_transform": (x7, lo, up), "unscale_transform": (x7, lo, up), "saturate": (x7, lo, up),
        "quat_diff_rad": (q, q2), "local_to_world_space": (v, pose), "my_quat_rotate": (q, v), "quat_to_angle_axis": (q,),
        "angle_axis_to_exp_map": (ang, torch.nn.functional.normalize(v, dim=-1)), "quat_to_exp_map": (q,), "quat_to_tan_norm": (q,),
        "euler_xyz_to_exp_map": (ang, ang.flip(0) * 0.5, ang * 0.25), "exp_map_to_angle_axis": (em,), "exp_map_to_quat": (em,), "slerp": (q, q2, tt),
        "calc_heading": (q,), "calc_heading_quat": (q,), "calc_heading_quat_inv": (q,),
        "compute_rot": (q, v, t, torch.tensor([1000.0, 0.0, 0.0]).repeat(n, 1), t * 0.1),
        "quaternion_to_matrix": (q,),
    }


def run(mod, out=None):
    out = {} if out is None else out
    for name, ins in cases().items():
        res = getattr(mod, name)(*[a.clone() for a in ins])
        res = res if isinstance(res, tuple) else (res,)
        for i, a in enumerate(ins):
            out[f"{name}__in{i}"] = a.numpy()
        for i, r in enumerate(res):
            out[f"{name}__out{i}"] = r.numpy()
    # functions with non-tensor arguments / chained cases
    q = cases()["quat_mul"][0]
    out["quat_axis__in0"] = q.numpy()
    for ax in range(3):
        out[f"quat_axis__out{ax}"] = mod.quat_axis(q, ax).numpy()
    v0, v1 = torch.tensor([1.0, 0.0, 0.0]).repeat(len(q), 1), torch.tensor([0.0, 0.0, 1.0]).repeat(len(q), 1)
    inv = torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(len(q), 1)
    tgt = cases()["quat_apply"][1].clone(); tgt[:, 2] = 0.0
    res = mod.compute_heading_and_up(q, inv, tgt, v0, v1, 2)
    out["compute_heading_and_up__in0"] = tgt.numpy()
    for i, r in enumerate(res):
        out[f"compute_heading_and_up__out{i}"] = r.numpy()
    m = mod.quaternion_to_matrix(q)
    out["matrix_to_quaternion__in0"] = m.numpy()
    out["matrix_to_quaternion__out0"] = mod.matrix_to_quaternion(m).numpy()
    out["get_axis_params__out0"] = np.asarray([mod.get_axis_params(-9.81, 2), mod.get_axis_params(1.5, 1, x_value=0.3), mod.get_axis_params(2.0, 0)], np.float64)
    out["copysign__out0"] = mod.copysign(1.5, torch.tensor([-2.0, 0.0, 3.0])).numpy()
    return out


def main():
    torch.set_num_threads(1)
    G.import_reference()
    ref = importlib.import_module("isaacgymenvs.utils.torch_jit_utils")
    out = run(ref)
    path = os.path.join(G.OUT, "torch_jit_utils.npz")
    np.savez_compressed(path, **out)
    print(path, len(out), "arrays")


if __name__ == "__main__":
    main()

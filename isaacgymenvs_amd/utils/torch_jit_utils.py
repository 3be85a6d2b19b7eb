"""Quaternion / transform / scaling helpers with the names and argument conventions user scripts import from the reference's
`isaacgymenvs/utils/torch_jit_utils.py` (and from `isaacgym.torch_utils`, which carries the same names: the shim re-exports this module).

Host-side convenience only: the engine's kernels do their own quaternion arithmetic (csrc/core/quat.hpp); nothing on the step path calls
these.  Quaternions are xyzw (`torch_jit_utils.py:48`).  Everything is written on three primitives -- `_vw` (split), `_cross`, and the
Rodrigues form `v + w t + u x t` with `t = 2 u x v` -- and accepts any number of leading batch dimensions (the reference's versions take
`[N, 4]`); results agree with the reference's functions to fp32 rounding (`tests/test_torch_jit_utils.py`, golden vectors generated from the
reference's own functions by `tools/gen_golden_torch_utils.py`).  Reference line numbers in the comments are into that file.
"""
from __future__ import annotations

import math

import numpy as np
import torch

_TWO_PI = 2.0 * math.pi


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):  # :37
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def _vw(q):
    return q[..., :3], q[..., 3:4]


def _cross(a, b):
    return torch.linalg.cross(a, b, dim=-1)


def _unit_vec(like, axis):
    e = torch.zeros_like(like[..., :3])
    e[..., axis] = 1.0
    return e


def _turn(q, v, sign):
    """v rotated by q (sign +1) or by its inverse (sign -1)"""
    u, w = _vw(q)
    t = 2.0 * _cross(u, v)
    return v + sign * w * t + _cross(u, t)


# ------------------------------------------------------------------ quaternion algebra
def quat_mul(a, b):  # :42-63 (Hamilton product, xyzw)
    assert a.shape == b.shape
    ua, wa = _vw(a)
    ub, wb = _vw(b)
    return torch.cat([wa * ub + wb * ua + _cross(ua, ub), wa * wb - (ua * ub).sum(-1, keepdim=True)], dim=-1)


def normalize(x, eps: float = 1e-9):  # :66-67
    return x / x.norm(p=2, dim=-1, keepdim=True).clamp_min(eps)


def quat_apply(a, b):  # :71-77
    shape = b.shape
    return _turn(a.reshape(-1, 4), b.reshape(-1, 3), 1.0).view(shape)


def quat_rotate(q, v):  # :81-90 (for unit q the same rotation as quat_apply)
    u, w = _vw(q)
    return v * (2.0 * w * w - 1.0) + 2.0 * w * _cross(u, v) + 2.0 * u * (u * v).sum(-1, keepdim=True)


def quat_rotate_inverse(q, v):  # :94-103
    u, w = _vw(q)
    return v * (2.0 * w * w - 1.0) - 2.0 * w * _cross(u, v) + 2.0 * u * (u * v).sum(-1, keepdim=True)


my_quat_rotate = quat_rotate  # :411-419


def quat_conjugate(a):  # :107-110
    return torch.cat([-a[..., :3], a[..., 3:]], dim=-1)


def quat_unit(a):  # :114-115
    return normalize(a)


def quat_from_angle_axis(angle, axis):  # :119-123
    half = 0.5 * angle.unsqueeze(-1)
    return quat_unit(torch.cat([normalize(axis) * half.sin(), half.cos()], dim=-1))


def normalize_angle(x):  # :127-128 -> (-pi, pi]
    return torch.atan2(torch.sin(x), torch.cos(x))


def quat_axis(q, axis: int = 0):  # :280-284
    return quat_rotate(q, _unit_vec(q, axis))


def quat_diff_rad(a, b):  # :354-372
    d = quat_mul(a, quat_conjugate(b))
    return 2.0 * torch.asin(d[..., :3].norm(p=2, dim=-1).clamp(max=1.0))


# ------------------------------------------------------------------ rigid transforms
def tf_inverse(q, t):  # :132-134
    qi = quat_conjugate(q)
    return qi, -quat_apply(qi, t)


def tf_apply(q, t, v):  # :138-139
    return quat_apply(q, v) + t


def tf_vector(q, v):  # :143-144
    return quat_apply(q, v)


def tf_combine(q1, t1, q2, t2):  # :148-149
    return quat_mul(q1, q2), quat_apply(q1, t2) + t1


def get_basis_vector(q, v):  # :153-154
    return quat_rotate(q, v)


def local_to_world_space(pos_offset_local, pose_global):  # :376-393
    return _turn(pose_global[..., 3:7], pos_offset_local, 1.0) + pose_global[..., 0:3]


def normalise_quat_in_pose(pose):  # :397-408 (normalises the caller's quaternion columns in place, like the reference)
    quat = pose[:, 3:7]
    quat /= torch.norm(quat, dim=-1, p=2).reshape(-1, 1)
    return torch.cat([pose[:, 0:3], quat], dim=-1)


# ------------------------------------------------------------------ Euler angles
def get_axis_params(value, axis_idx, x_value=0.0, dtype=float, n_dims=3):  # :157-165
    """arguments for a `Vec3` that is `value` along `axis_idx`; element 0 is then overwritten with `x_value` (the reference's order)"""
    assert axis_idx < n_dims, "the axis dim should be within the vector dimensions"
    out = np.zeros((n_dims,))
    out[axis_idx] = value
    out[0] = x_value
    return list(out.astype(dtype))


def copysign(a: float, b):  # :169-172
    return abs(float(a)) * torch.sign(b).to(torch.float)


def get_euler_xyz(q):  # :176-195 -> roll, pitch, yaw, each wrapped to [0, 2 pi)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    roll = torch.atan2(2.0 * (w * x + y * z), w * w - x * x - y * y + z * z)
    sp = 2.0 * (w * y - z * x)
    pitch = torch.where(sp.abs() >= 1.0, copysign(0.5 * math.pi, sp), torch.asin(sp.clamp(-1.0, 1.0)))
    yaw = torch.atan2(2.0 * (w * z + x * y), w * w + x * x - y * y - z * z)
    return roll % _TWO_PI, pitch % _TWO_PI, yaw % _TWO_PI


def quat_from_euler_xyz(roll, pitch, yaw):  # :199-212
    cr, sr = torch.cos(0.5 * roll), torch.sin(0.5 * roll)
    cp, sp = torch.cos(0.5 * pitch), torch.sin(0.5 * pitch)
    cy, sy = torch.cos(0.5 * yaw), torch.sin(0.5 * yaw)
    return torch.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
                        cy * cr * cp + sy * sr * sp], dim=-1)


# ------------------------------------------------------------------ random draws, clamps, scaling
def torch_rand_float(lower: float, upper: float, shape, device):  # :216-218
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def torch_random_dir_2(shape, device):  # :222-225
    ang = torch_rand_float(-math.pi, math.pi, shape, device).squeeze(-1)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)


def tensor_clamp(t, min_t, max_t):  # :229-230
    return torch.max(torch.min(t, max_t), min_t)


saturate = tensor_clamp  # :333-350 (same operation, (x, lower, upper))


def scale(x, lower, upper):  # :234-235  [-1, 1] -> [lower, upper]
    return 0.5 * (x + 1.0) * (upper - lower) + lower


def unscale(x, lower, upper):  # :239-240  [lower, upper] -> [-1, 1]
    return (2.0 * x - upper - lower) / (upper - lower)


def unscale_np(x, lower, upper):  # :243-244
    return (2.0 * x - upper - lower) / (upper - lower)


def scale_transform(x, lower, upper):  # :292-309 (normalise to [-1, 1])
    return 2.0 * (x - 0.5 * (lower + upper)) / (upper - lower)


def unscale_transform(x, lower, upper):  # :313-329 (back to [lower, upper])
    return x * (upper - lower) * 0.5 + 0.5 * (lower + upper)


# ------------------------------------------------------------------ locomotion observation helpers
def compute_heading_and_up(torso_rotation, inv_start_rot, to_target, vec0, vec1, up_idx: int):  # :248-262
    torso_quat = quat_mul(torso_rotation, inv_start_rot)
    up_vec = quat_rotate(torso_quat, vec1)
    heading_vec = quat_rotate(torso_quat, vec0)
    return torso_quat, up_vec[:, up_idx], (heading_vec * normalize(to_target)).sum(-1), up_vec, heading_vec


def compute_rot(torso_quat, velocity, ang_velocity, targets, torso_positions):  # :266-276
    roll, pitch, yaw = get_euler_xyz(torso_quat)
    # the reference measures the walk-target angle with the z (not y) difference (:272-273); kept
    walk_target_angle = torch.atan2(targets[:, 2] - torso_positions[:, 2], targets[:, 0] - torso_positions[:, 0])
    return (quat_rotate_inverse(torso_quat, velocity), quat_rotate_inverse(torso_quat, ang_velocity), roll, pitch, yaw,
            walk_target_angle - yaw)


# ------------------------------------------------------------------ axis-angle / exponential map
def quat_to_angle_axis(q):  # :423-442 (q normalised); below the 1e-5 threshold: angle 0 about z
    w = q[..., 3]
    s = torch.sqrt(1.0 - w * w)
    ok = s > 1e-5
    angle = torch.where(ok, normalize_angle(2.0 * torch.acos(w)), torch.zeros_like(w))
    axis = torch.where(ok.unsqueeze(-1), q[..., :3] / s.unsqueeze(-1), _unit_vec(q, 2))
    return angle, axis


def angle_axis_to_exp_map(angle, axis):  # :446-450
    return angle.unsqueeze(-1) * axis


def quat_to_exp_map(q):  # :454-459
    return angle_axis_to_exp_map(*quat_to_angle_axis(q))


def euler_xyz_to_exp_map(roll, pitch, yaw):  # :563-566
    return quat_to_exp_map(quat_from_euler_xyz(roll, pitch, yaw))


def exp_map_to_angle_axis(exp_map):  # :570-586
    raw = torch.norm(exp_map, dim=-1)
    angle = normalize_angle(raw)
    ok = angle > 1e-5
    axis = torch.where(ok.unsqueeze(-1), exp_map / raw.unsqueeze(-1), _unit_vec(exp_map, 2))
    return torch.where(ok, angle, torch.zeros_like(angle)), axis


def exp_map_to_quat(exp_map):  # :589-592
    return quat_from_angle_axis(*exp_map_to_angle_axis(exp_map))


def quat_to_tan_norm(q):  # :548-560: the rotated x axis followed by the rotated z axis
    return torch.cat([quat_rotate(q, _unit_vec(q, 0)), quat_rotate(q, _unit_vec(q, 2))], dim=-1)


def slerp(q0, q1, t):  # :595-627 (t broadcasts against [..., 1])
    c = (q0 * q1).sum(-1, keepdim=True)
    q1 = torch.where(c < 0, -q1, q1)
    c = c.abs()
    half = torch.acos(c)
    s = torch.sqrt(1.0 - c * c)
    out = (torch.sin((1.0 - t) * half) * q0 + torch.sin(t * half) * q1) / s
    out = torch.where(s.abs() < 0.001, 0.5 * q0 + 0.5 * q1, out)
    return torch.where(c >= 1.0, q0, out)


def calc_heading(q):  # :630-640: yaw of the rotated x axis
    d = quat_rotate(q, _unit_vec(q, 0))
    return torch.atan2(d[..., 1], d[..., 0])


def calc_heading_quat(q):  # :643-653
    return quat_from_angle_axis(calc_heading(q), _unit_vec(q, 2))


def calc_heading_quat_inv(q):  # :656-666
    return quat_from_angle_axis(-calc_heading(q), _unit_vec(q, 2))


# ------------------------------------------------------------------ rotation matrices (wxyz, as in the reference: :462-545)
def quaternion_to_matrix(quaternions):
    """`quaternions` real part FIRST (w, x, y, z) -- the one place the reference departs from xyzw (:462-487)"""
    w, x, y, z = torch.unbind(quaternions, -1)
    k = 2.0 / (quaternions * quaternions).sum(-1)
    rows = (1 - k * (y * y + z * z), k * (x * y - z * w), k * (x * z + y * w),
            k * (x * y + z * w), 1 - k * (x * x + z * z), k * (y * z - x * w),
            k * (x * z - y * w), k * (y * z + x * w), 1 - k * (x * x + y * y))
    return torch.stack(rows, -1).reshape(quaternions.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix):
    """rotation matrices -> (w, x, y, z); picks, per matrix, the best-conditioned of the four candidate reconstructions (:501-545)"""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    batch = matrix.shape[:-2]
    m = matrix.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, dim=-1)
    mag = torch.sqrt(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22],
                                 dim=-1).clamp_min(0.0))
    a, b, c = m21 - m12, m02 - m20, m10 - m01
    d, e, f = m10 + m01, m02 + m20, m12 + m21
    cand = torch.stack([torch.stack([mag[..., 0] ** 2, a, b, c], dim=-1), torch.stack([a, mag[..., 1] ** 2, d, e], dim=-1),
                        torch.stack([b, d, mag[..., 2] ** 2, f], dim=-1), torch.stack([c, e, f, mag[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * mag[..., None].clamp_min(0.1))
    pick = mag.argmax(dim=-1)
    return torch.gather(cand, -2, pick[..., None, None].expand(batch + (1, 4))).squeeze(-2)

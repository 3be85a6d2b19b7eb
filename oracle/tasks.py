"""CPU restatement (numpy, fp32) of the reference's per-task compute functions and step loop.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function cites the reference lines it follows (paths relative to /root/reference/isaacgymenvs/).  Pinned by
tests/test_oracle_golden.py against tests/golden/*.npz, which were produced by running the reference's own
@torch.jit.script functions (tools/gen_golden.py).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


# ------------------------------------------------------------------ utils/torch_jit_utils.py
def quat_mul(a, b):  # :42-63
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = f32(0.5) * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return np.stack([x, y, z, w], axis=-1)


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=-1)


def quat_rotate(q, v, inverse=False):  # :80-104
    qw = q[:, 3:4]
    qv = q[:, :3]
    a = v * (f32(2.0) * qw * qw - f32(1.0))
    b = _cross(qv, v) * qw * f32(2.0)
    d = ((qv[:, 0] * v[:, 0] + qv[:, 1] * v[:, 1]) + qv[:, 2] * v[:, 2])[:, None]
    c = qv * d * f32(2.0)
    return (a - b + c) if inverse else (a + b + c)


def py_mod(a, b):
    m = np.fmod(a, b)
    return np.where((m != 0) & ((b < 0) != (m < 0)), m + b, m).astype(f32)


def get_euler_roll_yaw(q):  # :175-195
    qx, qy, qz, qw = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    sinr_cosp = f32(2.0) * (qw * qx + qy * qz)
    cosr_cosp = qw * qw - qx * qx - qy * qy + qz * qz
    roll = np.arctan2(sinr_cosp, cosr_cosp).astype(f32)
    siny_cosp = f32(2.0) * (qw * qz + qx * qy)
    cosy_cosp = qw * qw + qx * qx - qy * qy - qz * qz
    yaw = np.arctan2(siny_cosp, cosy_cosp).astype(f32)
    two_pi = f32(2 * np.pi)
    return py_mod(roll, two_pi), py_mod(yaw, two_pi)


def normalize_angle(x):  # :126-128
    return np.arctan2(np.sin(x), np.cos(x)).astype(f32)


def unscale(x, lo, up):  # :238-240
    return (f32(2.0) * x - up - lo) / (up - lo)


# ------------------------------------------------------------------ tasks/ant.py:374-408, tasks/humanoid.py:378-413
def compute_locomotion_observations(hum, root_states, targets, potentials, inv_start_rot, dof_pos, dof_vel, dof_force,
                                    lower, upper, dof_vel_scale, sensors, actions, dt, contact_force_scale,
                                    angular_velocity_scale, basis_vec0, basis_vec1):
    r = root_states.astype(f32)
    pos, rot, vel, angvel = r[:, 0:3], r[:, 3:7], r[:, 7:10], r[:, 10:13]
    to_target = (targets.astype(f32) - pos).copy()
    to_target[:, 2] = 0
    prev_potentials_new = potentials.astype(f32).copy()
    nrm = np.sqrt((to_target[:, 0] * to_target[:, 0] + to_target[:, 1] * to_target[:, 1]) + to_target[:, 2] * to_target[:, 2]).astype(f32)
    potentials = (-nrm / f32(dt)).astype(f32)
    target_dirs = to_target / np.maximum(nrm, f32(1e-9))[:, None]  # normalize(), :66-67
    torso_quat = quat_mul(rot, inv_start_rot.astype(f32))           # compute_heading_and_up :247-262
    up_vec = quat_rotate(torso_quat, basis_vec1.astype(f32))
    heading_vec = quat_rotate(torso_quat, basis_vec0.astype(f32))
    up_proj = up_vec[:, 2]
    heading_proj = (heading_vec[:, 0] * target_dirs[:, 0] + heading_vec[:, 1] * target_dirs[:, 1]) + heading_vec[:, 2] * target_dirs[:, 2]
    vel_loc = quat_rotate(torso_quat, vel, inverse=True)             # compute_rot :265-276
    angvel_loc = quat_rotate(torso_quat, angvel, inverse=True)
    roll, yaw = get_euler_roll_yaw(torso_quat)
    walk_target_angle = np.arctan2(targets[:, 2].astype(f32) - pos[:, 2], targets[:, 0].astype(f32) - pos[:, 0]).astype(f32)
    angle_to_target = walk_target_angle - yaw
    dof_pos_scaled = unscale(dof_pos.astype(f32), lower.astype(f32), upper.astype(f32))
    if hum:
        roll, yaw, angle_to_target = normalize_angle(roll), normalize_angle(yaw), normalize_angle(angle_to_target)
        cols = [pos[:, 2:3], vel_loc, angvel_loc * f32(angular_velocity_scale), yaw[:, None], roll[:, None],
                angle_to_target[:, None], up_proj[:, None], heading_proj[:, None], dof_pos_scaled,
                dof_vel.astype(f32) * f32(dof_vel_scale), dof_force.astype(f32) * f32(contact_force_scale),
                sensors.astype(f32) * f32(contact_force_scale), actions.astype(f32)]
    else:
        cols = [pos[:, 2:3], vel_loc, angvel_loc, yaw[:, None], roll[:, None], angle_to_target[:, None],
                up_proj[:, None], heading_proj[:, None], dof_pos_scaled, dof_vel.astype(f32) * f32(dof_vel_scale),
                sensors.astype(f32) * f32(contact_force_scale), actions.astype(f32)]
    obs = np.concatenate(cols, axis=-1).astype(f32)
    return obs, potentials, prev_potentials_new, up_vec.astype(f32), heading_vec.astype(f32)


# ------------------------------------------------------------------ tasks/ant.py:325-371, tasks/humanoid.py:323-375
def compute_locomotion_reward(hum, obs, reset_buf, progress_buf, actions, up_weight, heading_weight, potentials,
                              prev_potentials, actions_cost_scale, energy_cost_scale, joints_at_limit_cost_scale,
                              termination_height, death_cost, max_episode_length, motor_efforts=None,
                              max_motor_effort=None):
    obs = obs.astype(f32)
    actions = actions.astype(f32)
    nd = actions.shape[1]
    hw, uw = f32(heading_weight), f32(up_weight)
    heading_reward = np.where(obs[:, 11] > f32(0.8), hw, hw * obs[:, 11] / f32(0.8)).astype(f32)
    up_reward = np.where(obs[:, 10] > f32(0.93), f32(0) + uw, f32(0)).astype(f32)
    actions_cost = np.zeros(len(obs), f32)
    electricity_cost = np.zeros(len(obs), f32)
    dof_at_limit_cost = np.zeros(len(obs), f32)
    for d in range(nd):  # sequential fp32 sums, like the kernel
        actions_cost = actions_cost + actions[:, d] * actions[:, d]
        if hum:
            ratio = f32(motor_efforts[d]) / f32(max_motor_effort)
            ap = np.abs(obs[:, 12 + d])
            scaled = f32(joints_at_limit_cost_scale) * (ap - f32(0.98)) / f32(0.02)
            dof_at_limit_cost = dof_at_limit_cost + (ap > f32(0.98)).astype(f32) * scaled * ratio
            electricity_cost = electricity_cost + np.abs(actions[:, d] * obs[:, 12 + nd + d]) * ratio
        else:
            electricity_cost = electricity_cost + np.abs(actions[:, d] * obs[:, 12 + nd + d])
            dof_at_limit_cost = dof_at_limit_cost + (obs[:, 12 + d] > f32(0.99)).astype(f32)
    alive = f32(2.0 if hum else 0.5)
    progress_reward = potentials.astype(f32) - prev_potentials.astype(f32)
    if hum:
        total = progress_reward + alive + up_reward + heading_reward - f32(actions_cost_scale) * actions_cost - \
            f32(energy_cost_scale) * electricity_cost - dof_at_limit_cost
    else:
        total = progress_reward + alive + up_reward + heading_reward - f32(actions_cost_scale) * actions_cost - \
            f32(energy_cost_scale) * electricity_cost - dof_at_limit_cost * f32(joints_at_limit_cost_scale)
    fallen = obs[:, 0] < f32(termination_height)
    total = np.where(fallen, f32(death_cost), total).astype(f32)
    reset = np.where(fallen, 1, reset_buf).astype(np.int64)
    reset = np.where(progress_buf.astype(f32) >= f32(max_episode_length) - f32(1), 1, reset).astype(np.int64)
    return total, reset


# ------------------------------------------------------------------ tasks/cartpole.py:180-196
def compute_cartpole_reward(pole_angle, pole_vel, cart_vel, cart_pos, reset_dist, reset_buf, progress_buf,
                            max_episode_length):
    pa, pv, cv, cp = (x.astype(f32) for x in (pole_angle, pole_vel, cart_vel, cart_pos))
    reward = f32(1.0) - pa * pa - f32(0.01) * np.abs(cv) - f32(0.005) * np.abs(pv)
    reward = np.where(np.abs(cp) > f32(reset_dist), f32(-2.0), reward)
    reward = np.where(np.abs(pa) > f32(np.pi / 2), f32(-2.0), reward).astype(f32)
    reset = np.where(np.abs(cp) > f32(reset_dist), 1, reset_buf)
    reset = np.where(np.abs(pa) > f32(np.pi / 2), 1, reset)
    reset = np.where(progress_buf.astype(f32) >= f32(max_episode_length) - f32(1), 1, reset).astype(np.int64)
    return reward, reset


# ------------------------------------------------------------------ counter-based reset RNG (csrc/core/rng.hpp)
def _fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def mi_uniform(seed, env, episode, k):
    """Bit-exact twin of mi::uniform01.  env/episode/k broadcastable uint32 arrays."""
    with np.errstate(over="ignore"):
        seed = np.uint32(seed & 0xFFFFFFFF)
        env, episode, k = (np.asarray(x).astype(np.uint32) for x in (env, episode, k))
        h = _fmix32(seed ^ (env * np.uint32(0x9E3779B1)).astype(np.uint32))
        h = _fmix32(h ^ (episode * np.uint32(0x85EBCA77)).astype(np.uint32))
        h = _fmix32(h ^ (k * np.uint32(0xC2B2AE3D)).astype(np.uint32))
    return (h >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)


def fold_seed(seed64):
    return (seed64 ^ (seed64 >> 32)) & 0xFFFFFFFF


# ------------------------------------------------------------------ full VecTask.step restatement on the CPU physics oracle
class OracleLocomotionEnv:
    """vec_task.py:360-408 + ant.py / humanoid.py pre/post_physics_step on oracle/physics.c (AoS, numpy).

    `params` is the same MiLocoParams ctypes struct the HIP engine receives, so both sides read identical numbers.
    """

    def __init__(self, hum, spec, sensor_bodies, sim_params: dict, params, num_envs, seed=0, env_id_offset=0,
                 precision="f32", control_freq_inv=1):
        from .engine import OracleEngine
        self.hum, self.N, self.p, self.nd = hum, num_envs, params, spec.nd
        self.eng = OracleEngine(spec, num_envs, params=sim_params, sensor_bodies=sensor_bodies, precision=precision)
        self.seed, self.off, self.cfi = fold_seed(seed), env_id_offset, control_freq_inv
        nd = self.nd
        self.lower = np.array(params.dof_lower[:nd], f32)
        self.upper = np.array(params.dof_upper[:nd], f32)
        self.init_dof = np.array(params.initial_dof_pos[:nd], f32)
        self.gear = np.array(params.gear[:nd], f32)
        self.initial_root = np.zeros((num_envs, 13), f32)
        self.initial_root[:, 2] = params.start_height
        self.initial_root[:, 6] = 1
        self.eng.root[:] = self.initial_root
        self.eng.q[:] = self.init_dof
        self.potentials = np.full(num_envs, f32(-1000.0) / f32(params.dt), f32)
        self.prev_potentials = self.potentials.copy()
        self.reset_buf = np.ones(num_envs, np.int64)
        self.progress_buf = np.zeros(num_envs, np.int64)
        self.episode = np.zeros(num_envs, np.uint32)
        self.obs_buf = np.zeros((num_envs, 12 + nd * (4 if hum else 3) + 6 * len(sensor_bodies)), f32)
        self.rew_buf = np.zeros(num_envs, f32)
        self.timeout_buf = np.zeros(num_envs, bool)
        self.targets = np.tile(np.array(params.targets[:], f32), (num_envs, 1))
        self.inv_start_rot = np.tile(np.array(params.inv_start_rot[:], f32), (num_envs, 1))
        self.b0 = np.tile(np.array(params.basis_vec0[:], f32), (num_envs, 1))
        self.b1 = np.tile(np.array(params.basis_vec1[:], f32), (num_envs, 1))

    def reset_idx(self, ids):  # ant.py:252-279
        p, nd = self.p, self.nd
        if len(ids) == 0:
            return
        genv = (self.off + ids).astype(np.uint32)[:, None]
        ep = self.episode[ids][:, None]
        k = np.arange(nd, dtype=np.uint32)[None, :]
        up, lo = f32(p.reset_pos_noise), -f32(p.reset_pos_noise)
        rp = (up - lo) * mi_uniform(self.seed, genv, ep, k) + lo
        vu, vl = f32(p.reset_vel_noise), -f32(p.reset_vel_noise)
        rv = (vu - vl) * mi_uniform(self.seed, genv, ep, k + np.uint32(nd)) + vl
        self.eng.q[ids] = np.maximum(np.minimum(self.init_dof + rp, self.upper), self.lower)
        self.eng.qd[ids] = rv
        self.eng.root[ids] = self.initial_root[ids]
        self.eng.lam[ids] = 0
        tx = f32(p.targets[0]) - self.initial_root[ids, 0]
        ty = f32(p.targets[1]) - self.initial_root[ids, 1]
        pp = (-np.sqrt((tx * tx + ty * ty) + f32(0)) / f32(p.dt)).astype(f32)
        self.prev_potentials[ids] = pp
        self.potentials[ids] = pp
        self.episode[ids] += 1
        self.progress_buf[ids] = 0
        self.reset_buf[ids] = 0

    def step(self, actions):
        p = self.p
        a = np.clip(actions.astype(f32), -f32(p.clip_actions), f32(p.clip_actions))  # vec_task.py:374
        self.actions = a
        tau = a * self.gear * f32(p.power_scale)                                      # ant.py:281-285
        for _ in range(self.cfi):
            self.eng.step(tau)                                                        # vec_task.py:379-382
        self.progress_buf += 1                                                        # ant.py:287-297
        self.reset_idx(np.nonzero(self.reset_buf)[0])
        e = self.eng
        self.obs_buf, self.potentials, self.prev_potentials, self.up_vec, self.heading_vec = \
            compute_locomotion_observations(self.hum, e.root.astype(f32), self.targets, self.potentials, self.inv_start_rot,
                                            e.q.astype(f32), e.qd.astype(f32), e.dof_force.astype(f32), self.lower,
                                            self.upper, p.dof_vel_scale, e.sensor.astype(f32), a, p.dt,
                                            p.contact_force_scale, p.angular_velocity_scale, self.b0, self.b1)
        self.rew_buf, self.reset_buf = compute_locomotion_reward(
            self.hum, self.obs_buf, self.reset_buf, self.progress_buf, a, p.up_weight, p.heading_weight, self.potentials,
            self.prev_potentials, p.actions_cost, p.energy_cost, p.joints_at_limit_cost, p.termination_height,
            p.death_cost, p.max_episode_length, self.gear, p.max_motor_effort)
        self.timeout_buf = (self.progress_buf.astype(f32) >= f32(p.max_episode_length) - f32(1)) & (self.reset_buf != 0)
        return self.obs_buf, self.rew_buf, self.reset_buf


class OracleCartpoleEnv:
    """vec_task.py:360-408 + cartpole.py:131-174 on the CPU physics oracle."""

    def __init__(self, spec, sim_params: dict, params, num_envs, seed=0, env_id_offset=0, precision="f32"):
        from .engine import OracleEngine
        self.N, self.p = num_envs, params
        self.eng = OracleEngine(spec, num_envs, params=sim_params, precision=precision)
        self.eng.root[:, 2] = 2.0  # cartpole.py:93
        self.seed, self.off = fold_seed(seed), env_id_offset
        self.reset_buf = np.ones(num_envs, np.int64)
        self.progress_buf = np.zeros(num_envs, np.int64)
        self.episode = np.zeros(num_envs, np.uint32)

    def reset_idx(self, ids):  # cartpole.py:144-157
        if len(ids) == 0:
            return
        genv = (self.off + ids).astype(np.uint32)[:, None]
        ep = self.episode[ids][:, None]
        k = np.arange(2, dtype=np.uint32)[None, :]
        self.eng.q[ids] = f32(0.2) * (mi_uniform(self.seed, genv, ep, k) - f32(0.5))
        self.eng.qd[ids] = f32(0.5) * (mi_uniform(self.seed, genv, ep, k + np.uint32(2)) - f32(0.5))
        self.eng.lam[ids] = 0
        self.episode[ids] += 1
        self.progress_buf[ids] = 0
        self.reset_buf[ids] = 0

    def step(self, actions):
        p = self.p
        a = np.clip(actions.astype(f32).reshape(-1), -f32(p.clip_actions), f32(p.clip_actions))
        tau = np.zeros((self.N, 2), f32)
        tau[:, 0] = a * f32(p.max_push_effort)  # cartpole.py:159-163
        self.eng.step(tau)
        self.progress_buf += 1
        self.reset_idx(np.nonzero(self.reset_buf)[0])
        q, qd = self.eng.q.astype(f32), self.eng.qd.astype(f32)
        self.obs_buf = np.stack([q[:, 0], qd[:, 0], q[:, 1], qd[:, 1]], axis=-1)
        self.rew_buf, self.reset_buf = compute_cartpole_reward(self.obs_buf[:, 2], self.obs_buf[:, 3], self.obs_buf[:, 1],
                                                               self.obs_buf[:, 0], p.reset_dist, self.reset_buf,
                                                               self.progress_buf, p.max_episode_length)
        self.timeout_buf = (self.progress_buf.astype(f32) >= f32(p.max_episode_length) - f32(1)) & (self.reset_buf != 0)
        return self.obs_buf, self.rew_buf, self.reset_buf

"""CPU restatement (numpy, fp32) of the reference's remaining @torch.jit.script task functions.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product path is csrc/kernels_jit_twins.hip.

Each function cites the reference lines it follows (paths relative to /root/reference/isaacgymenvs/).  Pinned by
tests/test_oracle_golden.py against tests/golden/jit_twins_*.npz = outputs of the reference's own jitted functions
(tools/gen_golden_jit_twins.py).
"""
from __future__ import annotations

import numpy as np

from .tasks import (f32, normalize_angle, quat_apply, quat_conjugate, quat_from_angle_axis, quat_mul, quat_rotate)


def _norm3(v):
    v = v.astype(f32)
    return np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]).astype(f32)


def tf_combine(q1, t1, q2, t2):  # utils/torch_jit_utils.py:148-149
    return quat_mul(q1.astype(f32), q2.astype(f32)), (quat_apply(q1.astype(f32), t2.astype(f32)) + t1.astype(f32)).astype(f32)


def quat_diff_rad(a, b):  # utils/torch_jit_utils.py:354-372
    m = quat_mul(a.astype(f32), quat_conjugate(b.astype(f32)))
    return (f32(2.0) * np.arcsin(np.minimum(_norm3(m[:, 0:3]), f32(1.0)))).astype(f32)


def local_to_world_space(pos_local, pose):  # utils/torch_jit_utils.py:376-393
    qp = np.concatenate([pos_local.astype(f32), np.zeros((len(pos_local), 1), f32)], axis=-1)
    qg = pose[:, 3:7].astype(f32)
    off = quat_mul(qg, quat_mul(qp, quat_conjugate(qg)))[:, 0:3]
    return (off + pose[:, 0:3].astype(f32)).astype(f32)


# ------------------------------------------------------------------ tasks/ball_balance.py
def compute_bbot_reward(tray_positions, ball_positions, ball_velocities, ball_radius, reset_buf, progress_buf, max_episode_length):
    """ball_balance.py:459-476."""
    p, v = ball_positions.astype(f32), ball_velocities.astype(f32)
    ball_dist = np.sqrt((p[:, 0] * p[:, 0] + (p[:, 2] - f32(0.7)) * (p[:, 2] - f32(0.7))) + p[:, 1] * p[:, 1]).astype(f32)
    ball_speed = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]).astype(f32)
    reward = ((f32(1.0) / (f32(1.0) + ball_dist)) * (f32(1.0) / (f32(1.0) + ball_speed))).astype(f32)
    reset = np.where(progress_buf.astype(f32) >= f32(max_episode_length) - f32(1), 1, reset_buf)
    reset = np.where(p[:, 2] < f32(ball_radius) * f32(1.5), 1, reset).astype(np.int64)
    return reward, reset


# ------------------------------------------------------------------ tasks/ingenuity.py
def compute_ingenuity_reward(root_positions, target_root_positions, root_quats, root_linvels, root_angvels, reset_buf, progress_buf,
                             max_episode_length):
    """ingenuity.py:410-442."""
    target_dist = _norm3(target_root_positions.astype(f32) - root_positions.astype(f32))
    pos_reward = (f32(1.0) / (f32(1.0) + target_dist * target_dist)).astype(f32)
    z = np.zeros((len(root_quats), 3), f32)
    z[:, 2] = 1
    ups = quat_rotate(root_quats.astype(f32), z)
    tiltage = np.abs(f32(1) - ups[:, 2]).astype(f32)
    up_reward = (f32(5.0) / (f32(1.0) + tiltage * tiltage)).astype(f32)
    spinnage = np.abs(root_angvels[:, 2].astype(f32))
    spinnage_reward = (f32(1.0) / (f32(1.0) + spinnage * spinnage)).astype(f32)
    reward = (pos_reward + pos_reward * (up_reward + spinnage_reward)).astype(f32)
    die = np.where(target_dist > f32(8.0), 1, 0)
    die = np.where(root_positions[:, 2].astype(f32) < f32(0.5), 1, die)
    reset = np.where(progress_buf.astype(f32) >= f32(max_episode_length) - f32(1), 1, die).astype(np.int64)
    return reward, reset


# ------------------------------------------------------------------ tasks/franka_cabinet.py
def compute_franka_cabinet_reward(reset_buf, progress_buf, actions, cabinet_dof_pos, franka_grasp_pos, drawer_grasp_pos, franka_grasp_rot,
                                  drawer_grasp_rot, franka_lfinger_pos, franka_rfinger_pos, gripper_forward_axis, drawer_inward_axis,
                                  gripper_up_axis, drawer_up_axis, num_envs, dist_reward_scale, rot_reward_scale, around_handle_reward_scale,
                                  open_reward_scale, finger_dist_reward_scale, action_penalty_scale, distX_offset, max_episode_length):
    """franka_cabinet.py:488-553."""
    d = _norm3(franka_grasp_pos.astype(f32) - drawer_grasp_pos.astype(f32))
    dist_reward = (f32(1.0) / (f32(1.0) + d * d)).astype(f32)
    dist_reward = dist_reward * dist_reward
    dist_reward = np.where(d <= f32(0.02), dist_reward * f32(2), dist_reward).astype(f32)
    a1 = quat_apply(franka_grasp_rot.astype(f32), gripper_forward_axis.astype(f32))
    a2 = quat_apply(drawer_grasp_rot.astype(f32), drawer_inward_axis.astype(f32))
    a3 = quat_apply(franka_grasp_rot.astype(f32), gripper_up_axis.astype(f32))
    a4 = quat_apply(drawer_grasp_rot.astype(f32), drawer_up_axis.astype(f32))
    dot1 = ((a1[:, 0] * a2[:, 0] + a1[:, 1] * a2[:, 1]) + a1[:, 2] * a2[:, 2]).astype(f32)
    dot2 = ((a3[:, 0] * a4[:, 0] + a3[:, 1] * a4[:, 1]) + a3[:, 2] * a4[:, 2]).astype(f32)
    rot_reward = (f32(0.5) * (np.sign(dot1) * (dot1 * dot1) + np.sign(dot2) * (dot2 * dot2))).astype(f32)
    lf, rf, dg = franka_lfinger_pos.astype(f32), franka_rfinger_pos.astype(f32), drawer_grasp_pos.astype(f32)
    around = (lf[:, 2] > dg[:, 2]) & (rf[:, 2] < dg[:, 2])
    around_handle_reward = np.where(around, f32(0.5), f32(0)).astype(f32)
    lfd, rfd = np.abs(lf[:, 2] - dg[:, 2]), np.abs(rf[:, 2] - dg[:, 2])
    finger_dist_reward = np.where(around, (f32(0.04) - lfd) + (f32(0.04) - rfd), f32(0)).astype(f32)
    action_penalty = np.zeros(len(d), f32)
    for k in range(actions.shape[1]):
        action_penalty = action_penalty + actions[:, k].astype(f32) * actions[:, k].astype(f32)
    dp = cabinet_dof_pos[:, 3].astype(f32)
    open_reward = (dp * around_handle_reward + dp).astype(f32)
    r = (((((f32(dist_reward_scale) * dist_reward + f32(rot_reward_scale) * rot_reward) + f32(around_handle_reward_scale) * around_handle_reward) +
           f32(open_reward_scale) * open_reward) + f32(finger_dist_reward_scale) * finger_dist_reward) -
         f32(action_penalty_scale) * action_penalty).astype(f32)
    r = np.where(dp > f32(0.01), r + f32(0.5), r).astype(f32)
    r = np.where(dp > f32(0.2), r + around_handle_reward, r).astype(f32)
    r = np.where(dp > f32(0.39), r + f32(2.0) * around_handle_reward, r).astype(f32)
    r = np.where(lf[:, 0] < dg[:, 0] - f32(distX_offset), f32(-1), r).astype(f32)
    r = np.where(rf[:, 0] < dg[:, 0] - f32(distX_offset), f32(-1), r).astype(f32)
    reset = np.where(dp > f32(0.39), 1, reset_buf)
    reset = np.where(progress_buf.astype(f32) >= f32(max_episode_length) - f32(1), 1, reset).astype(np.int64)
    return r, reset


def compute_grasp_transforms(hand_rot, hand_pos, franka_local_grasp_rot, franka_local_grasp_pos, drawer_rot, drawer_pos,
                             drawer_local_grasp_rot, drawer_local_grasp_pos):
    """franka_cabinet.py:556-568."""
    gfr, gfp = tf_combine(hand_rot, hand_pos, franka_local_grasp_rot, franka_local_grasp_pos)
    gdr, gdp = tf_combine(drawer_rot, drawer_pos, drawer_local_grasp_rot, drawer_local_grasp_pos)
    return gfr, gfp, gdr, gdp


# ------------------------------------------------------------------ tasks/franka_cube_stack.py
def axisangle2quat(vec, eps=1e-6):
    """franka_cube_stack.py:40-71."""
    vec = vec.astype(f32)
    angle = _norm3(vec)
    quat = np.zeros((len(vec), 4), f32)
    quat[:, 3] = 1
    idx = angle > f32(eps)
    a = angle[idx][:, None]
    quat[idx, 0:3] = vec[idx] * np.sin(a / f32(2.0)).astype(f32) / a
    quat[idx, 3] = np.cos(a[:, 0] / f32(2.0)).astype(f32)
    return quat


def compute_franka_cube_stack_reward(reset_buf, progress_buf, actions, states, reward_settings, max_episode_length):
    """franka_cube_stack.py:697-752."""
    s = {k: v.astype(f32) for k, v in states.items()}
    cubeA_size, cubeB_size = s["cubeA_size"], s["cubeB_size"]
    target_height = (cubeB_size + cubeA_size / f32(2.0)).astype(f32)
    d = _norm3(s["cubeA_pos_relative"])
    d_lf = _norm3(s["cubeA_pos"] - s["eef_lf_pos"])
    d_rf = _norm3(s["cubeA_pos"] - s["eef_rf_pos"])
    dist_reward = (f32(1) - np.tanh(f32(10.0) * ((d + d_lf) + d_rf) / f32(3))).astype(f32)
    cubeA_height = (s["cubeA_pos"][:, 2] - f32(reward_settings["table_height"])).astype(f32)
    lifted = (cubeA_height - cubeA_size) > f32(0.04)
    lift = lifted.astype(f32)
    off = s["cubeA_to_cubeB_pos"].copy()
    off[:, 2] = off[:, 2] + (cubeA_size + cubeB_size) / f32(2)
    d_ab = _norm3(off)
    align_reward = ((f32(1) - np.tanh(f32(10.0) * d_ab)) * lift).astype(f32)
    dist_reward = np.maximum(dist_reward, align_reward)
    ab = s["cubeA_to_cubeB_pos"]
    aligned = np.sqrt(ab[:, 0] * ab[:, 0] + ab[:, 1] * ab[:, 1]).astype(f32) < f32(0.02)
    on_b = np.abs(cubeA_height - target_height) < f32(0.02)
    away = d > f32(0.04)
    stack = aligned & on_b & away
    rewards = np.where(stack, f32(reward_settings["r_stack_scale"]) * stack.astype(f32),
                       (f32(reward_settings["r_dist_scale"]) * dist_reward + f32(reward_settings["r_lift_scale"]) * lift) +
                       f32(reward_settings["r_align_scale"]) * align_reward).astype(f32)
    reset = np.where((progress_buf.astype(f32) >= f32(max_episode_length) - f32(1)) | stack, 1, reset_buf).astype(np.int64)
    return rewards, reset


# ------------------------------------------------------------------ tasks/allegro_hand.py
def randomize_rotation_pen(rand0, rand1, max_angle, x_unit, y_unit, z_unit):
    """allegro_hand.py:728-732."""
    r0 = rand0.astype(f32)
    return quat_mul(quat_from_angle_axis(f32(0.5) * f32(np.pi) + r0 * f32(max_angle), x_unit), quat_from_angle_axis(r0 * f32(np.pi), z_unit))


# ------------------------------------------------------------------ tasks/trifinger.py
def lgsk_kernel(x, scale=50.0, eps=2.0):
    """trifinger.py:1260-1274."""
    s = x.astype(f32) * f32(scale)
    return (f32(1.0) / ((np.exp(s).astype(f32) + f32(eps)) + np.exp(-s).astype(f32))).astype(f32)


def gen_keypoints(pose, num_keypoints=8, size=(0.065, 0.065, 0.065)):
    """trifinger.py:1277-1290."""
    out = np.zeros((len(pose), num_keypoints, 3), f32)
    for i in range(num_keypoints):
        corner = np.array([(1 if ((i >> k) & 1) == 0 else -1) * f32(size[k]) / f32(2) for k in range(3)], f32)
        out[:, i, :] = local_to_world_space(np.tile(corner, (len(pose), 1)), pose)
    return out


def compute_trifinger_reward(obs_buf, reset_buf, progress_buf, episode_length, dt, finger_move_penalty_weight, finger_reach_object_weight,
                             object_dist_weight, object_rot_weight, env_steps_count, object_goal_poses_buf, object_state, last_object_state,
                             fingertip_state, last_fingertip_state, use_keypoints):
    """trifinger.py:1292-1383; returns (reward, reset, info)."""
    ft, lft = fingertip_state.astype(f32), last_fingertip_state.astype(f32)
    os_, los = object_state.astype(f32), last_object_state.astype(f32)
    vel = ((ft[:, :, 0:3] - lft[:, :, 0:3]) / f32(dt)).astype(f32).reshape(len(ft), 9)
    acc = np.zeros(len(ft), f32)
    for k in range(9):
        acc = acc + vel[:, k] * vel[:, k]
    finger_movement_penalty = (f32(finger_move_penalty_weight) * acc).astype(f32)
    dsum = np.zeros(len(ft), f32)
    for i in range(3):
        dsum = dsum + (_norm3(ft[:, i, 0:3] - os_[:, 0:3]) - _norm3(lft[:, i, 0:3] - los[:, 0:3]))
    sched = f32(1.0) if 0 <= env_steps_count <= 5e7 else f32(0.0)
    finger_reach_object_reward = (f32(finger_reach_object_weight) * sched * dsum).astype(f32)
    if use_keypoints:
        ok, gk = gen_keypoints(os_[:, 0:7]), gen_keypoints(object_goal_poses_buf[:, 0:7].astype(f32))
        s = np.zeros(len(ft), f32)
        for i in range(8):
            s = s + lgsk_kernel(_norm3(ok[:, i] - gk[:, i]), 30.0, 2.0)
        pose_reward = (f32(object_dist_weight) * f32(dt) * (s / f32(8))).astype(f32)
    else:
        dist = _norm3(os_[:, 0:3] - object_goal_poses_buf[:, 0:3].astype(f32))
        object_dist_reward = (f32(object_dist_weight) * f32(dt) * lgsk_kernel(dist, 50.0, 2.0)).astype(f32)
        angles = quat_diff_rad(os_[:, 3:7], object_goal_poses_buf[:, 3:7])
        object_rot_reward = (f32(object_rot_weight) * f32(dt) / (f32(3.) * np.abs(angles) + f32(0.01))).astype(f32)
        pose_reward = object_dist_reward + object_rot_reward
    total = ((finger_movement_penalty + finger_reach_object_reward) + pose_reward).astype(f32)
    reset = np.where(progress_buf >= episode_length - 1, 1, 0).astype(np.int64)
    info = {"finger_movement_penalty": finger_movement_penalty, "finger_reach_object_reward": finger_reach_object_reward,
            "pose_reward": finger_reach_object_reward, "reward": total}   # 'pose_reward' aliases the reach reward in the reference (:1378)
    return total, reset, info


def compute_trifinger_observations_states(asymmetric_obs, dof_position, dof_velocity, object_state, object_goal_poses, actions,
                                          fingertip_state, joint_torques, tip_wrenches):
    """trifinger.py:1386-1420."""
    n = len(dof_position)
    obs = np.concatenate([dof_position, dof_velocity, object_state[:, 0:7], object_goal_poses, actions], axis=-1).astype(f32)
    if asymmetric_obs:
        states = np.concatenate([obs, object_state[:, 7:13], fingertip_state.reshape(n, -1), joint_torques, tip_wrenches], axis=-1).astype(f32)
    else:
        states = obs
    return obs, states


# ------------------------------------------------------------------ tasks/amp/humanoid_amp_base.py
def quat_to_tan_norm(q):  # utils/torch_jit_utils.py:548-560
    t = np.zeros((len(q), 3), f32); t[:, 0] = 1
    n = np.zeros((len(q), 3), f32); n[:, 2] = 1
    return np.concatenate([quat_rotate(q.astype(f32), t), quat_rotate(q.astype(f32), n)], axis=-1).astype(f32)


def exp_map_to_quat(exp_map):  # utils/torch_jit_utils.py:577-603
    e = exp_map.astype(f32)
    angle = _norm3(e)
    with np.errstate(divide="ignore", invalid="ignore"):
        axis = (e / angle[:, None]).astype(f32)
    angle = normalize_angle(angle)
    mask = angle > f32(1e-5)
    angle = np.where(mask, angle, f32(0)).astype(f32)
    default = np.zeros_like(e); default[:, 2] = 1
    axis = np.where(mask[:, None], axis, default).astype(f32)
    return quat_from_angle_axis(angle, axis)


def calc_heading_quat_inv(q):  # utils/torch_jit_utils.py:627-667
    ref = np.zeros((len(q), 3), f32); ref[:, 0] = 1
    d = quat_rotate(q.astype(f32), ref)
    heading = np.arctan2(d[:, 1], d[:, 0]).astype(f32)
    z = np.zeros((len(q), 3), f32); z[:, 2] = 1
    return quat_from_angle_axis(-heading, z)


AMP_DOF_OFFSETS = [0, 3, 6, 9, 10, 13, 14, 17, 18, 21, 24, 25, 28]


def amp_dof_to_obs(pose):
    """amp/humanoid_amp_base.py:462-492."""
    out = []
    for j in range(len(AMP_DOF_OFFSETS) - 1):
        a, b = AMP_DOF_OFFSETS[j], AMP_DOF_OFFSETS[j + 1]
        jp = pose[:, a:b].astype(f32)
        out.append(quat_to_tan_norm(exp_map_to_quat(jp)) if b - a == 3 else jp)
    return np.concatenate(out, axis=-1).astype(f32)


def compute_humanoid_amp_observations(root_states, dof_pos, dof_vel, key_body_pos, local_root_obs):
    """amp/humanoid_amp_base.py:494-528 (== humanoid_amp.py:299-330 build_amp_observations)."""
    r = root_states.astype(f32)
    hinv = calc_heading_quat_inv(r[:, 3:7])
    rr = quat_mul(hinv, r[:, 3:7]) if local_root_obs else r[:, 3:7]
    n, nk = key_body_pos.shape[0], key_body_pos.shape[1]
    local = (key_body_pos.astype(f32) - r[:, None, 0:3]).reshape(n * nk, 3)
    hrep = np.repeat(hinv[:, None, :], nk, axis=1).reshape(n * nk, 4)
    flat = quat_rotate(hrep, local).reshape(n, nk * 3)
    return np.concatenate([r[:, 2:3], quat_to_tan_norm(rr), quat_rotate(hinv, r[:, 7:10]), quat_rotate(hinv, r[:, 10:13]),
                           amp_dof_to_obs(dof_pos), dof_vel.astype(f32), flat], axis=-1).astype(f32)


def compute_humanoid_amp_reset(reset_buf, progress_buf, contact_buf, contact_body_ids, rigid_body_pos, max_episode_length,
                               enable_early_termination, termination_height):
    """amp/humanoid_amp_base.py:536-564."""
    terminated = np.zeros_like(reset_buf)
    if enable_early_termination:
        masked = contact_buf.astype(f32).copy()
        masked[:, contact_body_ids, :] = 0
        fall_contact = np.any(np.any(masked > f32(0.1), axis=-1), axis=-1)
        fall_height = rigid_body_pos[..., 2].astype(f32) < f32(termination_height)
        fall_height[:, contact_body_ids] = False
        fall_height = np.any(fall_height, axis=-1)
        has_fallen = fall_contact & fall_height & (progress_buf > 1)
        terminated = np.where(has_fallen, 1, terminated)
    reset = np.where(progress_buf.astype(f32) >= f32(max_episode_length) - f32(1), 1, terminated).astype(np.int64)
    return reset, terminated.astype(np.int64)


# ------------------------------------------------------------------ tasks/dextreme/allegro_hand_dextreme.py
def compute_hand_reward_dextreme(rew_buf, reset_buf, reset_goal_buf, progress_buf, hold_count_buf, cur_targets, prev_targets, hand_dof_vel,
                                 successes, consecutive_successes, max_episode_length, object_pos, object_rot, target_pos, target_rot,
                                 dist_reward_scale, rot_reward_scale, rot_eps, actions, action_penalty_scale, action_delta_penalty_scale,
                                 success_tolerance, reach_goal_bonus, fall_dist, fall_penalty, max_consecutive_successes, av_factor,
                                 num_success_hold_steps):
    """dextreme/allegro_hand_dextreme.py:1598-1663; returns the reference's 15-tuple."""
    def sumsq(x):
        acc = np.zeros(len(x), f32)
        for k in range(x.shape[1]):
            acc = acc + x[:, k].astype(f32) * x[:, k].astype(f32)
        return acc
    goal_dist = _norm3(object_pos.astype(f32) - target_pos.astype(f32))
    rot_dist = quat_diff_rad(object_rot, target_rot)
    dist_rew = (goal_dist * f32(dist_reward_scale)).astype(f32)
    rot_rew = (f32(1.0) / (np.abs(rot_dist) + f32(rot_eps)) * f32(rot_reward_scale)).astype(f32)
    action_penalty = (f32(action_penalty_scale) * sumsq(actions)).astype(f32)
    action_delta_penalty = (f32(action_delta_penalty_scale) * sumsq(cur_targets.astype(f32) - prev_targets.astype(f32))).astype(f32)
    velocity_penalty = (f32(-0.05) * sumsq(hand_dof_vel.astype(f32) / f32(5.0 - 1.0))).astype(f32)
    near = np.abs(rot_dist) <= f32(success_tolerance)
    goal_reached = np.where(near, 1, reset_goal_buf)
    hold = np.where(goal_reached != 0, hold_count_buf + 1, 0).astype(np.int64)
    goal_resets = np.where(hold > num_success_hold_steps, 1, reset_goal_buf).astype(np.int64)
    successes = (successes.astype(f32) + goal_resets.astype(f32)).astype(f32)
    reach_goal_rew = np.where(goal_resets == 1, f32(reach_goal_bonus), f32(0)).astype(f32)
    fell = goal_dist >= f32(fall_dist)
    fall_rew = np.where(fell, f32(fall_penalty), f32(0)).astype(f32)
    resets = np.where(fell, 1, reset_buf)
    progress = progress_buf.copy()
    if max_consecutive_successes > 0:
        progress = np.where(near, 0, progress)
        resets = np.where(successes >= max_consecutive_successes, 1, resets)
    timed_out = progress.astype(f32) >= f32(max_episode_length) - f32(1)
    resets = np.where(timed_out, 1, resets).astype(np.int64)
    timeout_rew = np.where(timed_out, f32(0.5) * f32(fall_penalty), f32(0)).astype(f32)
    reward = (((((((dist_rew + rot_rew) + action_penalty) + action_delta_penalty) + velocity_penalty) + reach_goal_rew) + fall_rew) + timeout_rew).astype(f32)
    num_resets = f32(resets.sum())
    fin = f32(np.sum(successes * resets.astype(f32), dtype=f32))
    cs = f32(consecutive_successes)
    cons = f32(av_factor) * fin / num_resets + (f32(1.0) - f32(av_factor)) * cs if num_resets > 0 else cs
    return (reward, resets, goal_resets, progress, hold, successes, f32(cons), dist_rew, rot_rew, action_penalty, action_delta_penalty,
            velocity_penalty, reach_goal_rew, fall_rew, timeout_rew)


# ------------------------------------------------------------------ tasks/trifinger.py cuboid-pose samplers on injected draws
def tri_random_xy(u, max_com_distance_to_center):
    """trifinger.py:1427-1439; u[:, 0] is the radius draw, u[:, 1] the angle draw."""
    radius = np.sqrt(u[:, 0].astype(f32)) * f32(max_com_distance_to_center)
    theta = f32(2 * np.pi) * u[:, 1].astype(f32)
    return (radius * np.cos(theta).astype(f32)).astype(f32), (radius * np.sin(theta).astype(f32)).astype(f32)


def tri_random_z(u, min_height, max_height):
    """trifinger.py:1442-1448."""
    return ((f32(max_height) - f32(min_height)) * u.astype(f32) + f32(min_height)).astype(f32)


def tri_random_orientation(g):
    """trifinger.py:1460-1470."""
    g = g.astype(f32)
    n = np.maximum(np.sqrt(((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]) + g[:, 3] * g[:, 3]).astype(f32), f32(1e-12))
    return (g / n[:, None]).astype(f32)


def tri_random_orientation_within_angle(u, base, max_angle):
    """trifinger.py:1472-1493."""
    u = u.astype(f32)
    c = np.cos(u[:, 0] * f32(max_angle)).astype(f32)
    n = np.sqrt((f32(1.) - c) / f32(2.)).astype(f32)
    q = np.zeros((len(u), 4), f32)
    q[:, 3] = np.sqrt((f32(1) + c) / f32(2.))
    q[:, 2] = (u[:, 1] * f32(2.) - f32(1.)) * n
    s = np.sqrt(f32(1) - q[:, 2] * q[:, 2]).astype(f32)
    ang = f32(2 * np.pi) * u[:, 2]
    q[:, 0] = (s * np.cos(ang).astype(f32)) * n
    q[:, 1] = (s * np.sin(ang).astype(f32)) * n
    return quat_mul(tri_random_orientation(q), base.astype(f32))


def tri_random_angular_vel(g, magnitude_stdev):
    """trifinger.py:1495-1503; g[:, 0:3] the axis draws, g[:, 3] the magnitude draw."""
    g = g.astype(f32)
    axis = g[:, 0:3] / _norm3(g[:, 0:3])[:, None]
    return ((g[:, 3:4] * f32(magnitude_stdev)) * axis).astype(f32)


def tri_random_yaw_orientation(u):
    """trifinger.py:1505-1512 with quat_from_euler_xyz (utils/torch_jit_utils.py:199-212), roll = pitch = 0."""
    yaw = f32(2 * np.pi) * u.astype(f32)
    cy, sy = np.cos(yaw * f32(0.5)).astype(f32), np.sin(yaw * f32(0.5)).astype(f32)
    z = np.zeros_like(cy)
    return np.stack([z, z, sy, cy], axis=-1).astype(f32)

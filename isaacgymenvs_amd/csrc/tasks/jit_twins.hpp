// jit_twins.hpp -- per-env maths of the remaining @torch.jit.script task functions of the reference, one
// function per reference function, evaluated expression by expression in the reference's order (FP contraction off):
//   isaacgymenvs/tasks/ball_balance.py:459      compute_bbot_reward
//   isaacgymenvs/tasks/ingenuity.py:410         compute_ingenuity_reward
//   isaacgymenvs/tasks/franka_cabinet.py:488    compute_franka_reward, :556 compute_grasp_transforms
//   isaacgymenvs/tasks/franka_cube_stack.py:40  axisangle2quat, :697 compute_franka_reward
//   isaacgymenvs/tasks/allegro_hand.py:728      randomize_rotation_pen   (compute_hand_reward :663 == shadow_hand.py:746)
//   isaacgymenvs/tasks/trifinger.py:1260        lgsk_kernel, :1277 gen_keypoints, :1292 compute_trifinger_reward
//   isaacgymenvs/tasks/amp/humanoid_amp_base.py:462 dof_to_obs, :494 compute_humanoid_observations, :536 compute_humanoid_reset
//   isaacgymenvs/tasks/dextreme/allegro_hand_dextreme.py:1598 compute_hand_reward
// and the torch_jit_utils.py helpers they call (quat_apply :70, quat_from_angle_axis :119, tf_combine :148,
// quat_diff_rad :354, local_to_world_space :376, exp_map_to_quat :599, quat_to_tan_norm :548, calc_heading_quat_inv :656).
#pragma once
#include "../core/quat.hpp"
#include "ball_balance.hpp"   // bbot_reward
#include "ingenuity.hpp"   // ingenuity_reward

namespace mi {

MI_HD float norm3(const float* v) { MI_NO_CONTRACT return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
MI_HD void cross3_nc(const float* a, const float* b, float* o) {
    MI_NO_CONTRACT
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
// torch_jit_utils.py:70-77
MI_HD void quat_apply(const float* a, const float* b, float* o) {
    MI_NO_CONTRACT
    float t[3], u[3];
    cross3_nc(a, b, t);
    for (int k = 0; k < 3; ++k) t[k] = t[k] * 2.f;
    cross3_nc(a, t, u);
    for (int k = 0; k < 3; ++k) o[k] = (b[k] + a[3] * t[k]) + u[k];
}
MI_HD void quat_conj(const float* a, float* o) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
// torch_jit_utils.py:119-123 (normalize :66 clamps the norm at 1e-9)
MI_HD void quat_from_angle_axis(float angle, const float* axis, float* o) {
    MI_NO_CONTRACT
    const float theta = angle / 2.f;
    const float an = fmaxf(norm3(axis), 1e-9f);
    const float s = sinf(theta);
    float q[4] = {(axis[0] / an) * s, (axis[1] / an) * s, (axis[2] / an) * s, cosf(theta)};
    const float qn = fmaxf(sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]), 1e-9f);
    for (int k = 0; k < 4; ++k) o[k] = q[k] / qn;
}
// torch_jit_utils.py:148-149
MI_HD void tf_combine(const float* q1, const float* t1, const float* q2, const float* t2, float* q, float* t) {
    MI_NO_CONTRACT
    quat_mul(q1, q2, q);
    float r[3];
    quat_apply(q1, t2, r);
    for (int k = 0; k < 3; ++k) t[k] = r[k] + t1[k];
}
// torch_jit_utils.py:354-372
MI_HD float quat_diff_rad(const float* a, const float* b) {
    float bc[4], m[4];
    quat_conj(b, bc);
    quat_mul(a, bc, m);
    return 2.0f * asinf(fminf(norm3(m), 1.0f));
}
// torch_jit_utils.py:376-393
MI_HD void local_to_world_space(const float* p_local, const float* pose, float* o) {
    MI_NO_CONTRACT
    const float qp[4] = {p_local[0], p_local[1], p_local[2], 0.f};
    float qc[4], t[4], r[4];
    quat_conj(pose + 3, qc);
    quat_mul(qp, qc, t);
    quat_mul(pose + 3, t, r);
    for (int k = 0; k < 3; ++k) o[k] = r[k] + pose[k];
}

// ------------------------------------------------------------------------------------------------ BallBalance: tasks/ball_balance.hpp (shared with the task kernels)

// ------------------------------------------------------------------------------------------------ Ingenuity: tasks/ingenuity.hpp (shared with the task kernels)

// ------------------------------------------------------------------------------------------------ FrankaCabinet
struct FrankaCabinetRewardParams {  // mirrors MiFrankaCabinetRewardParams; the float arguments of franka_cabinet.py:488-497
    float dist_reward_scale, rot_reward_scale, around_handle_reward_scale, open_reward_scale;
    float finger_dist_reward_scale, action_penalty_scale, distX_offset, max_episode_length;
};
MI_HD float sign_of(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
MI_HD void franka_cabinet_reward(const FrankaCabinetRewardParams& p, long long reset_in, long long progress, const float* actions, int na,
                                 float drawer_pos /* cabinet_dof_pos[:, 3] */, const float* franka_grasp_pos, const float* drawer_grasp_pos,
                                 const float* franka_grasp_rot, const float* drawer_grasp_rot, const float* lf, const float* rf,
                                 const float* gripper_forward_axis, const float* drawer_inward_axis, const float* gripper_up_axis,
                                 const float* drawer_up_axis, float* reward, long long* reset) {
    MI_NO_CONTRACT
    const float dv[3] = {franka_grasp_pos[0] - drawer_grasp_pos[0], franka_grasp_pos[1] - drawer_grasp_pos[1], franka_grasp_pos[2] - drawer_grasp_pos[2]};
    const float d = norm3(dv);
    float dist_reward = 1.0f / (1.0f + d * d);
    dist_reward = dist_reward * dist_reward;
    if (d <= 0.02f) dist_reward = dist_reward * 2.f;
    float a1[3], a2[3], a3[3], a4[3];
    quat_apply(franka_grasp_rot, gripper_forward_axis, a1);
    quat_apply(drawer_grasp_rot, drawer_inward_axis, a2);
    quat_apply(franka_grasp_rot, gripper_up_axis, a3);
    quat_apply(drawer_grasp_rot, drawer_up_axis, a4);
    const float dot1 = (a1[0] * a2[0] + a1[1] * a2[1]) + a1[2] * a2[2];
    const float dot2 = (a3[0] * a4[0] + a3[1] * a4[1]) + a3[2] * a4[2];
    const float rot_reward = 0.5f * (sign_of(dot1) * (dot1 * dot1) + sign_of(dot2) * (dot2 * dot2));
    const bool around = (lf[2] > drawer_grasp_pos[2]) && (rf[2] < drawer_grasp_pos[2]);
    const float around_handle_reward = around ? 0.5f : 0.f;
    const float lfd = fabsf(lf[2] - drawer_grasp_pos[2]), rfd = fabsf(rf[2] - drawer_grasp_pos[2]);
    const float finger_dist_reward = around ? ((0.04f - lfd) + (0.04f - rfd)) : 0.f;
    float action_penalty = 0.f;
    for (int k = 0; k < na; ++k) action_penalty += actions[k] * actions[k];
    const float open_reward = drawer_pos * around_handle_reward + drawer_pos;
    float r = ((((p.dist_reward_scale * dist_reward + p.rot_reward_scale * rot_reward) + p.around_handle_reward_scale * around_handle_reward) +
                p.open_reward_scale * open_reward) + p.finger_dist_reward_scale * finger_dist_reward) - p.action_penalty_scale * action_penalty;
    if (drawer_pos > 0.01f) r = r + 0.5f;
    if (drawer_pos > 0.2f) r = r + around_handle_reward;
    if (drawer_pos > 0.39f) r = r + 2.0f * around_handle_reward;
    if (lf[0] < drawer_grasp_pos[0] - p.distX_offset) r = -1.f;
    if (rf[0] < drawer_grasp_pos[0] - p.distX_offset) r = -1.f;
    long long rs = reset_in;
    if (drawer_pos > 0.39f) rs = 1;
    if ((float)progress >= p.max_episode_length - 1.f) rs = 1;
    *reward = r;
    *reset = rs;
}

// ------------------------------------------------------------------------------------------------ FrankaCubeStack
struct FrankaCubeStackRewardParams {  // mirrors MiFrankaCubeStackRewardParams; reward_settings of franka_cube_stack.py:82-87 + table height
    float r_dist_scale, r_lift_scale, r_align_scale, r_stack_scale, table_height, max_episode_length;
};
MI_HD void axisangle2quat(const float* vec, float eps, float* q) {   // franka_cube_stack.py:40-71
    MI_NO_CONTRACT
    const float angle = norm3(vec);
    if (angle > eps) {
        const float s = sinf(angle / 2.0f);
        for (int k = 0; k < 3; ++k) q[k] = vec[k] * s / angle;
        q[3] = cosf(angle / 2.0f);
    } else {
        q[0] = q[1] = q[2] = 0.f; q[3] = 1.f;
    }
}
MI_HD void franka_cube_stack_reward(const FrankaCubeStackRewardParams& p, long long reset_in, long long progress, float cubeA_size,
                                    float cubeB_size, const float* cubeA_pos, const float* cubeA_pos_relative, const float* eef_lf_pos,
                                    const float* eef_rf_pos, const float* cubeA_to_cubeB_pos, float* reward, long long* reset) {
    MI_NO_CONTRACT
    const float target_height = cubeB_size + cubeA_size / 2.0f;
    const float d = norm3(cubeA_pos_relative);
    const float l[3] = {cubeA_pos[0] - eef_lf_pos[0], cubeA_pos[1] - eef_lf_pos[1], cubeA_pos[2] - eef_lf_pos[2]};
    const float r[3] = {cubeA_pos[0] - eef_rf_pos[0], cubeA_pos[1] - eef_rf_pos[1], cubeA_pos[2] - eef_rf_pos[2]};
    const float d_lf = norm3(l), d_rf = norm3(r);
    float dist_reward = 1.f - tanhf(10.0f * ((d + d_lf) + d_rf) / 3.f);
    const float cubeA_height = cubeA_pos[2] - p.table_height;
    const bool lifted = (cubeA_height - cubeA_size) > 0.04f;
    const float lift_reward = lifted ? 1.f : 0.f;
    const float off[3] = {cubeA_to_cubeB_pos[0], cubeA_to_cubeB_pos[1], cubeA_to_cubeB_pos[2] + (cubeA_size + cubeB_size) / 2.f};
    const float d_ab = norm3(off);
    const float align_reward = (1.f - tanhf(10.0f * d_ab)) * lift_reward;
    dist_reward = fmaxf(dist_reward, align_reward);
    const bool aligned = sqrtf(cubeA_to_cubeB_pos[0] * cubeA_to_cubeB_pos[0] + cubeA_to_cubeB_pos[1] * cubeA_to_cubeB_pos[1]) < 0.02f;
    const bool on_b = fabsf(cubeA_height - target_height) < 0.02f;
    const bool away = d > 0.04f;
    const bool stack = aligned && on_b && away;
    *reward = stack ? p.r_stack_scale * 1.f
                    : (p.r_dist_scale * dist_reward + p.r_lift_scale * lift_reward) + p.r_align_scale * align_reward;
    *reset = (((float)progress >= p.max_episode_length - 1.f) || stack) ? 1 : reset_in;
}

// ------------------------------------------------------------------------------------------------ AllegroHand
MI_HD void randomize_rotation_pen(float rand0, float /*rand1*/, float max_angle, const float* x_unit, const float* /*y_unit*/,
                                  const float* z_unit, float* o) {   // allegro_hand.py:728-732 (rand1 and y_unit are unused there too)
    MI_NO_CONTRACT
    float a[4], b[4];
    quat_from_angle_axis(0.5f * 3.14159265358979323846f + rand0 * max_angle, x_unit, a);
    quat_from_angle_axis(rand0 * 3.14159265358979323846f, z_unit, b);
    quat_mul(a, b, o);
}

// ------------------------------------------------------------------------------------------------ Trifinger
MI_HD float lgsk(float x, float scale, float eps) {   // trifinger.py:1260-1274
    MI_NO_CONTRACT
    const float s = x * scale;
    return 1.0f / ((expf(s) + eps) + expf(-s));
}
MI_HD void gen_keypoints(const float* pose, const float* size, float* out /* [8][3] */) {   // trifinger.py:1277-1290
    MI_NO_CONTRACT
    for (int i = 0; i < 8; ++i) {
        float c[3];
        for (int k = 0; k < 3; ++k) c[k] = ((((i >> k) & 1) == 0) ? 1.f : -1.f) * size[k] / 2.f;
        local_to_world_space(c, pose, out + 3 * i);
    }
}
struct TrifingerRewardParams {   // mirrors MiTrifingerRewardParams; scalar arguments of trifinger.py:1292-1309
    int episode_length;
    float dt, finger_move_penalty_weight, finger_reach_object_weight, object_dist_weight, object_rot_weight;
    long long env_steps_count;
    int use_keypoints;
    float keypoint_size[3];       // gen_keypoints default (0.065, 0.065, 0.065)
};
MI_HD void trifinger_reward(const TrifingerRewardParams& p, long long progress, const float* goal_pose /*7*/, const float* object_state /*13*/,
                            const float* last_object_state, const float* fingertip_state /*[3][13]*/, const float* last_fingertip_state,
                            float* reward, long long* reset, float* finger_movement_penalty_out, float* finger_reach_object_reward_out) {
    MI_NO_CONTRACT
    float acc = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = (fingertip_state[13 * i + k] - last_fingertip_state[13 * i + k]) / p.dt;
            acc += v * v;
        }
    const float finger_movement_penalty = p.finger_move_penalty_weight * acc;
    float dsum = 0.f;
    for (int i = 0; i < 3; ++i) {
        const float a[3] = {fingertip_state[13 * i] - object_state[0], fingertip_state[13 * i + 1] - object_state[1], fingertip_state[13 * i + 2] - object_state[2]};
        const float b[3] = {last_fingertip_state[13 * i] - last_object_state[0], last_fingertip_state[13 * i + 1] - last_object_state[1],
                            last_fingertip_state[13 * i + 2] - last_object_state[2]};
        dsum += norm3(a) - norm3(b);
    }
    const float sched = (0 <= p.env_steps_count && (double)p.env_steps_count <= 5e7) ? 1.f : 0.f;
    const float finger_reach_object_reward = p.finger_reach_object_weight * sched * dsum;
    float pose_reward;
    if (p.use_keypoints) {
        float ok[24], gk[24];
        gen_keypoints(object_state, p.keypoint_size, ok);
        gen_keypoints(goal_pose, p.keypoint_size, gk);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) {
            const float dl[3] = {ok[3 * i] - gk[3 * i], ok[3 * i + 1] - gk[3 * i + 1], ok[3 * i + 2] - gk[3 * i + 2]};
            s += lgsk(norm3(dl), 30.f, 2.f);
        }
        pose_reward = p.object_dist_weight * p.dt * (s / 8.f);
    } else {
        const float dl[3] = {object_state[0] - goal_pose[0], object_state[1] - goal_pose[1], object_state[2] - goal_pose[2]};
        const float object_dist_reward = p.object_dist_weight * p.dt * lgsk(norm3(dl), 50.f, 2.f);
        const float angles = quat_diff_rad(object_state + 3, goal_pose + 3);
        const float object_rot_reward = p.object_rot_weight * p.dt / (3.f * fabsf(angles) + 0.01f);
        pose_reward = object_dist_reward + object_rot_reward;
    }
    *reward = (finger_movement_penalty + finger_reach_object_reward) + pose_reward;
    *reset = (progress >= (long long)p.episode_length - 1) ? 1 : 0;
    *finger_movement_penalty_out = finger_movement_penalty;
    *finger_reach_object_reward_out = finger_reach_object_reward;
}

// The cuboid-pose samplers (trifinger.py:1427-1505) draw from torch.rand / torch.randn INSIDE the jitted function; the twins take the
// same draws as an input tensor (column order = the order the reference calls the generator in) and apply the same map.
MI_HD void tri_random_xy(const float* u /*2*/, float max_com_distance_to_center, float* x, float* y) {   // :1427-1439
    MI_NO_CONTRACT
    float radius = sqrtf(u[0]);
    radius = radius * max_com_distance_to_center;
    const float theta = (2.f * 3.14159265358979323846f) * u[1];
    *x = radius * cosf(theta);
    *y = radius * sinf(theta);
}
MI_HD float tri_random_z(float u, float min_height, float max_height) { MI_NO_CONTRACT return (max_height - min_height) * u + min_height; }   // :1442-1448
MI_HD void tri_random_orientation(const float* g /*4 normal draws*/, float* q) {   // :1460-1470 (F.normalize, eps 1e-12)
    MI_NO_CONTRACT
    const float n = fmaxf(sqrtf(((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]) + g[3] * g[3]), 1e-12f);
    for (int k = 0; k < 4; ++k) q[k] = g[k] / n;
}
MI_HD void tri_random_orientation_within_angle(const float* u /*3*/, const float* base, float max_angle, float* out) {   // :1472-1493
    MI_NO_CONTRACT
    const float c = cosf(u[0] * max_angle);
    const float n = sqrtf((1.f - c) / 2.f);
    float q[4];
    q[3] = sqrtf((1.f + c) / 2.f);
    q[2] = (u[1] * 2.f - 1.f) * n;
    const float s = sqrtf(1.f - q[2] * q[2]);
    const float ang = (2.f * 3.14159265358979323846f) * u[2];
    q[0] = (s * cosf(ang)) * n;
    q[1] = (s * sinf(ang)) * n;
    const float m = fmaxf(sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]), 1e-12f);
    for (int k = 0; k < 4; ++k) q[k] = q[k] / m;
    quat_mul(q, base, out);
}
MI_HD void tri_random_angular_vel(const float* g /*4 normal draws: axis 3, magnitude 1*/, float magnitude_stdev, float* w) {   // :1495-1503
    MI_NO_CONTRACT
    const float n = norm3(g);
    const float mag = g[3] * magnitude_stdev;
    for (int k = 0; k < 3; ++k) w[k] = mag * (g[k] / n);
}
MI_HD void tri_random_yaw_orientation(float u, float* q) {   // :1505-1512 with quat_from_euler_xyz (torch_jit_utils.py:199-212), roll = pitch = 0
    MI_NO_CONTRACT
    const float yaw = (2.f * 3.14159265358979323846f) * u;
    const float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f), cr = 1.f, sr = 0.f, cp = 1.f, sp = 0.f;
    q[3] = cy * cr * cp + sy * sr * sp;
    q[0] = cy * sr * cp - sy * cr * sp;
    q[1] = cy * cr * sp + sy * sr * cp;
    q[2] = sy * cr * cp - cy * sr * sp;
}

// ------------------------------------------------------------------------------------------------ HumanoidAMP
MI_HD void quat_to_tan_norm(const float* q, float* o /*6*/) {   // torch_jit_utils.py:548-560
    const float rt[3] = {1.f, 0.f, 0.f}, rn[3] = {0.f, 0.f, 1.f};
    quat_rotate_s(q, rt, 1.f, o);
    quat_rotate_s(q, rn, 1.f, o + 3);
}
MI_HD void exp_map_to_quat(const float* e, float* q) {   // torch_jit_utils.py:577-603
    MI_NO_CONTRACT
    const float n = norm3(e);
    float axis[3] = {e[0] / n, e[1] / n, e[2] / n};
    float angle = normalize_angle(n);
    if (!(angle > 1e-5f)) { angle = 0.f; axis[0] = 0.f; axis[1] = 0.f; axis[2] = 1.f; }
    quat_from_angle_axis(angle, axis, q);
}
MI_HD void calc_heading_quat_inv(const float* q, float* o) {   // torch_jit_utils.py:627-667
    const float ref[3] = {1.f, 0.f, 0.f}, z[3] = {0.f, 0.f, 1.f};
    float d[3];
    quat_rotate_s(q, ref, 1.f, d);
    const float heading = atan2f(d[1], d[0]);
    quat_from_angle_axis(-heading, z, o);
}
constexpr int kAmpDof = 28, kAmpDofObs = 52, kAmpJoints = 12;
MI_HD void amp_dof_to_obs(const float* pose /*28*/, float* o /*52*/) {   // humanoid_amp_base.py:462-492
    const int off[kAmpJoints + 1] = {0, 3, 6, 9, 10, 13, 14, 17, 18, 21, 24, 25, 28};
    int w = 0;
    for (int j = 0; j < kAmpJoints; ++j) {
        const int sz = off[j + 1] - off[j];
        if (sz == 3) {
            float q[4];
            exp_map_to_quat(pose + off[j], q);
            quat_to_tan_norm(q, o + w);
            w += 6;
        } else {
            o[w] = pose[off[j]];
            w += 1;
        }
    }
}
// humanoid_amp_base.py:494-528 == humanoid_amp.py:299-330 (build_amp_observations); obs = 1 + 6 + 3 + 3 + 52 + 28 + 3 nk
MI_HD void amp_observations(const float* root /*13*/, const float* dof_pos, const float* dof_vel, const float* key_body_pos /*[nk][3]*/, int nk,
                            bool local_root_obs, float* obs) {
    MI_NO_CONTRACT
    float hinv[4], rr[4];
    calc_heading_quat_inv(root + 3, hinv);
    if (local_root_obs) quat_mul(hinv, root + 3, rr);
    else for (int k = 0; k < 4; ++k) rr[k] = root[3 + k];
    obs[0] = root[2];
    quat_to_tan_norm(rr, obs + 1);
    quat_rotate_s(hinv, root + 7, 1.f, obs + 7);
    quat_rotate_s(hinv, root + 10, 1.f, obs + 10);
    amp_dof_to_obs(dof_pos, obs + 13);
    for (int k = 0; k < kAmpDof; ++k) obs[13 + kAmpDofObs + k] = dof_vel[k];
    for (int b = 0; b < nk; ++b) {
        const float l[3] = {key_body_pos[3 * b] - root[0], key_body_pos[3 * b + 1] - root[1], key_body_pos[3 * b + 2] - root[2]};
        quat_rotate_s(hinv, l, 1.f, obs + 13 + kAmpDofObs + kAmpDof + 3 * b);
    }
}
// humanoid_amp_base.py:536-564; contact_body_mask bit b set <=> body b is in contact_body_ids
MI_HD void amp_reset(long long progress, const float* contact /*[nb][3]*/, const float* body_pos /*[nb][3]*/, int nb, unsigned long long contact_body_mask,
                     float max_episode_length, bool enable_early_termination, float termination_height, long long* reset, long long* terminated) {
    long long term = 0;
    if (enable_early_termination) {
        bool fall_contact = false, fall_height = false;
        for (int b = 0; b < nb; ++b) {
            if ((contact_body_mask >> b) & 1ull) continue;
            for (int k = 0; k < 3; ++k) fall_contact = fall_contact || (contact[3 * b + k] > 0.1f);
            fall_height = fall_height || (body_pos[3 * b + 2] < termination_height);
        }
        if (fall_contact && fall_height && progress > 1) term = 1;
    }
    *terminated = term;
    *reset = ((float)progress >= max_episode_length - 1.f) ? 1 : term;
}

// ------------------------------------------------------------------------------------------------ AllegroHand (DeXtreme)
struct DextremeRewardParams {   // mirrors MiDextremeRewardParams; scalar arguments of allegro_hand_dextreme.py:1598-1606
    float max_episode_length, dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale, action_delta_penalty_scale;
    float success_tolerance, reach_goal_bonus, fall_dist, fall_penalty;
    int max_consecutive_successes;
    float av_factor;
    int num_success_hold_steps;
};
// per-env part of allegro_hand_dextreme.py:1607-1660; out8 = dist_rew, rot_rew, action_penalty, action_delta_penalty, velocity_penalty,
// reach_goal_rew, fall_rew, timeout_rew.  The cross-env consecutive-successes average is reduced by the caller (kernel).
MI_HD void dextreme_reward(const DextremeRewardParams& p, long long reset_in, long long reset_goal_in, long long* progress, long long* hold_count,
                           const float* cur_targets, const float* prev_targets, const float* hand_dof_vel, int nd, float* successes,
                           const float* object_pos, const float* object_rot, const float* target_pos, const float* target_rot,
                           const float* actions, int na, float* reward, long long* resets, long long* goal_resets, float* out8) {
    MI_NO_CONTRACT
    const float dp[3] = {object_pos[0] - target_pos[0], object_pos[1] - target_pos[1], object_pos[2] - target_pos[2]};
    const float goal_dist = norm3(dp);
    const float rot_dist = quat_diff_rad(object_rot, target_rot);
    const float dist_rew = goal_dist * p.dist_reward_scale;
    const float rot_rew = 1.0f / (fabsf(rot_dist) + p.rot_eps) * p.rot_reward_scale;
    float a2 = 0.f, d2 = 0.f, v2 = 0.f;
    for (int k = 0; k < na; ++k) a2 += actions[k] * actions[k];
    for (int k = 0; k < nd; ++k) { const float d = cur_targets[k] - prev_targets[k]; d2 += d * d; }
    for (int k = 0; k < nd; ++k) { const float v = hand_dof_vel[k] / (5.0f - 1.0f); v2 += v * v; }
    const float action_penalty = p.action_penalty_scale * a2;
    const float action_delta_penalty = p.action_delta_penalty_scale * d2;
    const float velocity_penalty = -0.05f * v2;
    const bool near = fabsf(rot_dist) <= p.success_tolerance;
    const long long goal_reached = near ? 1 : reset_goal_in;
    const long long hc = goal_reached ? *hold_count + 1 : 0;
    const long long gr = (hc > p.num_success_hold_steps) ? 1 : reset_goal_in;
    const float succ = *successes + (float)gr;
    const float reach_goal_rew = (gr == 1) ? p.reach_goal_bonus : 0.f;
    const bool fell = goal_dist >= p.fall_dist;
    const float fall_rew = fell ? p.fall_penalty : 0.f;
    long long rs = fell ? 1 : reset_in;
    long long pr = *progress;
    if (p.max_consecutive_successes > 0) {
        if (near) pr = 0;
        if (succ >= (float)p.max_consecutive_successes) rs = 1;
    }
    const bool timed_out = (float)pr >= p.max_episode_length - 1.f;
    if (timed_out) rs = 1;
    const float timeout_rew = timed_out ? 0.5f * p.fall_penalty : 0.f;
    *reward = ((((((dist_rew + rot_rew) + action_penalty) + action_delta_penalty) + velocity_penalty) + reach_goal_rew) + fall_rew) + timeout_rew;
    *resets = rs; *goal_resets = gr; *progress = pr; *hold_count = hc; *successes = succ;
    out8[0] = dist_rew; out8[1] = rot_rew; out8[2] = action_penalty; out8[3] = action_delta_penalty; out8[4] = velocity_penalty;
    out8[5] = reach_goal_rew; out8[6] = fall_rew; out8[7] = timeout_rew;
}

}  // namespace mi

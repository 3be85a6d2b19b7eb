// shadow_hand.hpp -- ShadowHand task maths for one env (reference isaacgymenvs/tasks/shadow_hand.py).
//   compute_hand_reward   :746-800 (@torch.jit.script)       randomize_rotation :803-806 (@torch.jit.script)
//   compute_full_state    :528-584 (the 211-wide "full_state" observation)
// Only the task functions (the hand + object physics is core/hand_engine.hpp / hand_engine_mw.hpp).  Expression order follows the
// reference (fp32, contraction off).
#pragma once
#include "../core/quat.hpp"

namespace mi {

struct HandRewardParams {  // mirrors MiHandRewardParams (include/mi_engine.h)
    float max_episode_length;        // float, like the jitted signature (:749)
    float dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale;
    float success_tolerance, reach_goal_bonus, fall_dist, fall_penalty;
    int max_consecutive_successes;
    float av_factor;
    int ignore_z_rot;
};

struct HandParams {   // mirrors MiHandParams (include/mi_engine.h): what ShadowHand.__init__ reads (shadow_hand.py:45-110)
    HandRewardParams rew;
    float vel_obs_scale, force_torque_obs_scale;              // :61-62
    float reset_position_noise, reset_dof_pos_noise, reset_dof_vel_noise;   // :69-72
    float act_moving_average, dof_speed_scale, dt;            // :80-81, sim dt
    int use_relative_control;                                 // :79
    float clip_actions;
    float object_init_pos[3];                                 // hand start + (0, -0.39, 0.10), :309-315
    float goal_init_pos[3];                                   // object init - 0.04 z, :393-395
    float hand_pos[3], hand_quat[4];                          // actor pose (:306-307) x mount orientation (robot.xml:3)
    float cube_half, cube_mass, cube_inertia, mu;             // cube_multicolor.urdf
    int actuated[20];                                         // dof index of each of the 20 actuators (:268-269)
    int obs_type, num_obs, asymmetric_obs;                    // observationType (:97-110), asymmetric_observations (:88)
    short obs_map[160];                                       // obs_buf[:, k] = full_state[:, obs_map[k]] for obs_type != 0
    float force_scale, force_prob_range[2], force_decay, force_decay_interval;   // :69-72
    int object_shape;                                         // objectType (:86-96): 0 block (cube_* above), 1 pen (capsule), 2 egg (ellipsoid)
    float object_dims[3], object_inertia[3];                  // egg: semi-axes (egg.xml:10); pen: radius, half length (pen.xml:20); principal inertias
};

MI_HD void quat_conjugate(const float* a, float* o) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }

// per-env part of compute_hand_reward (:757-790); the cross-env consecutive_successes average (:792-797) is reduced by
// the caller from `resets` and `successes * resets`
MI_HD void hand_reward(const HandRewardParams& p, const float* object_pos, const float* object_rot, const float* target_pos,
                       const float* target_rot, const float* actions, int nact, long long reset_in, long long reset_goal_in,
                       long long progress_in, float successes_in, float* reward, long long* resets, long long* goal_resets,
                       long long* progress_out, float* successes_out) {
    MI_NO_CONTRACT
    const float dx = object_pos[0] - target_pos[0], dy = object_pos[1] - target_pos[1], dz = object_pos[2] - target_pos[2];
    const float goal_dist = sqrtf((dx * dx + dy * dy) + dz * dz);
    float tol = p.success_tolerance;
    if (p.ignore_z_rot) tol = 2.0f * tol;
    float conj[4], qd[4];
    quat_conjugate(target_rot, conj);
    quat_mul(object_rot, conj, qd);
    const float vn = sqrtf((qd[0] * qd[0] + qd[1] * qd[1]) + qd[2] * qd[2]);
    const float rot_dist = 2.0f * asinf(fminf(vn, 1.0f));
    const float dist_rew = goal_dist * p.dist_reward_scale;
    const float rot_rew = 1.0f / (fabsf(rot_dist) + p.rot_eps) * p.rot_reward_scale;
    float action_penalty = 0.f;
    for (int i = 0; i < nact; ++i) action_penalty += actions[i] * actions[i];
    float r = dist_rew + rot_rew + action_penalty * p.action_penalty_scale;
    const bool hit = fabsf(rot_dist) <= tol;
    const long long gr = hit ? 1 : reset_goal_in;
    const float succ = successes_in + (float)gr;
    if (gr == 1) r = r + p.reach_goal_bonus;
    const bool fell = goal_dist >= p.fall_dist;
    if (fell) r = r + p.fall_penalty;
    long long rs = fell ? 1 : reset_in;
    long long prog = progress_in;
    if (p.max_consecutive_successes > 0) {
        if (hit) prog = 0;
        if (succ >= (float)p.max_consecutive_successes) rs = 1;
    }
    const bool timeout = (float)prog >= p.max_episode_length - 1.f;
    if (timeout) rs = 1;
    if (p.max_consecutive_successes > 0 && timeout) r = r + 0.5f * p.fall_penalty;
    *reward = r; *resets = rs; *goal_resets = gr; *progress_out = prog; *successes_out = succ;
}

// quat_from_angle_axis (torch_jit_utils.py:119-123) with normalize (:66-67) and quat_unit
MI_HD void quat_from_angle_axis(float angle, const float* axis, float* q) {
    MI_NO_CONTRACT
    const float theta = angle / 2.f;
    const float n = fmaxf(sqrtf((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]), 1e-9f);
    const float s = sinf(theta), c = cosf(theta);
    float t[4] = {axis[0] / n * s, axis[1] / n * s, axis[2] / n * s, c};
    const float m = fmaxf(sqrtf(((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]) + t[3] * t[3]), 1e-9f);
    for (int i = 0; i < 4; ++i) q[i] = t[i] / m;
}
// randomize_rotation (:803-806)
MI_HD void randomize_rotation(float rand0, float rand1, const float* x_unit, const float* y_unit, float* q) {
    MI_NO_CONTRACT
    const float PI_ = 3.141592653589793f;
    float a[4], b[4];
    quat_from_angle_axis(rand0 * PI_, x_unit, a);
    quat_from_angle_axis(rand1 * PI_, y_unit, b);
    quat_mul(a, b, q);
}
// randomize_rotation_pen (:809-813): rand1 and y_unit are unused by the reference
MI_HD void randomize_rotation_pen(float rand0, float /*rand1*/, float max_angle, const float* x_unit, const float* /*y_unit*/, const float* z_unit,
                                  float* q) {
    MI_NO_CONTRACT
    const float PI_ = 3.141592653589793f;
    float a[4], b[4];
    quat_from_angle_axis(0.5f * PI_ + rand0 * max_angle, x_unit, a);
    quat_from_angle_axis(rand0 * PI_, z_unit, b);
    quat_mul(a, b, q);
}

}  // namespace mi

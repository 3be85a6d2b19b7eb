// anymal.hpp -- AnymalTerrain task maths for one env (reference isaacgymenvs/tasks/anymal_terrain.py).
//
//   pre_physics_step PD torques   :441-451       check_termination   :294-300
//   post_physics_step             :453-485       compute_reward      :315-382
//   compute_observations          :302-313       get_heights         :515-538
//   reset_idx                     :384-425       update_terrain_level:427-435
//   push_robots                   :437-439       quat_apply_yaw      :676-681, wrap_to_pi :683-687
// The reference runs these as ~150 eager PyTorch ops per step; here they are one pass per env inside the post kernel.
// Expression order follows the reference (fp32, contraction off) so that results agree to rounding.
#pragma once
#include "../core/quat.hpp"
#include "../core/rng.hpp"

namespace mi {

constexpr int kAnymalDof = 12;
constexpr int kAnymalHeightPts = 140;
constexpr int kAnymalObs = 188;       // 3+3+3+3+12+12+140+12 (anymal_terrain.py:305-313)
constexpr int kAnymalSums = 13;       // episode_sums keys (:166-168), in the order below
// order of reward terms / episode sums: lin_vel_xy, lin_vel_z, ang_vel_z, ang_vel_xy, orient, torques, joint_acc, base_height,
//                                       air_time, collision, stumble, action_rate, hip

struct AnymalParams {  // mirrors MiAnymalParams in include/mi_engine.h (same layout)
    // normalisation (:55-60)
    float lin_vel_scale, ang_vel_scale, dof_pos_scale, dof_vel_scale, height_meas_scale, action_scale;
    // reward scales, already multiplied by the control dt like the reference does (:104-105)
    float rew_termination, rew_lin_vel_xy, rew_lin_vel_z, rew_ang_vel_z, rew_ang_vel_xy, rew_orient, rew_torque, rew_joint_acc,
          rew_base_height, rew_air_time, rew_collision, rew_stumble, rew_action_rate, rew_hip;
    float command_x[2], command_y[2], command_yaw[2];   // :80-82
    float base_init_state[13];                           // :85-89
    float default_dof_pos[kAnymalDof];                   // :155-159
    float kp, kd, torque_limit;                          // control.stiffness / damping (:100-101), clip +-80 (:444-445)
    float dt;                                            // control dt = decimation * sim.dt (:95)
    float max_episode_length_s;                          // :96
    int max_episode_length;                              // :97
    int push_interval;                                   // :98 ; <= 0 disables pushing
    int allow_knee_contacts;                             // :99
    int decimation;                                      // :94
    int add_noise;                                       // :176
    // noise scale per observation group (:174-186): lin vel, ang vel, gravity, dof pos, dof vel, heights
    float noise_lin_vel, noise_ang_vel, noise_gravity, noise_dof_pos, noise_dof_vel, noise_height;
    int curriculum;                                      // terrain.curriculum (:102)
    float clip_actions;
    float friction_range[2];                             // learn.frictionRange (:237)
    float terrain_mu;                                    // terrain.staticFriction (:208)
};

struct AnymalTerrainDesc {   // device-side view of the terrain (set once through mi_engine_set_terrain)
    const short* hs;         // [rows*cols] int16 height samples
    int rows, cols;
    float hscale, vscale, border;
    const float* origins;    // [levels][types][3] terrain.env_origins
    int levels, types;
    float env_length;
    float slope_threshold;   // terrain.slopeTreshold (anymal_terrain.py:576), <= 0: the uncorrected mesh (option "terrain_slope_threshold")
    int walls;               // the risers of the corrected mesh collide from the side (option "terrain_walls", default 1)
};

// body indices in the compiled model (base, then LF, RF, LH, RH x {HIP, THIGH, SHANK}); `footName: SHANK`, `kneeName: THIGH`
MI_HD constexpr int anymal_foot_body(int k) { return 3 + 3 * k; }
MI_HD constexpr int anymal_knee_body(int k) { return 2 + 3 * k; }

// torch_jit_utils.py quat_apply (:67-73): v + w*t + xyz x t, t = 2 * (xyz x v)
MI_HD void quat_apply(const float* q, const float* v, float* o) {
    MI_NO_CONTRACT
    const float t[3] = {(q[1] * v[2] - q[2] * v[1]) * 2.f, (q[2] * v[0] - q[0] * v[2]) * 2.f, (q[0] * v[1] - q[1] * v[0]) * 2.f};
    const float c[3] = {q[1] * t[2] - q[2] * t[1], q[2] * t[0] - q[0] * t[2], q[0] * t[1] - q[1] * t[0]};
    for (int i = 0; i < 3; ++i) o[i] = v[i] + q[3] * t[i] + c[i];
}
// anymal_terrain.py:683-687.  `angles %= 2*np.pi` inside @torch.jit.script executes as aten::fmod_ (C semantics, sign of
// the dividend) -- pinned by the golden vectors of the reference's own function: inputs in (-2pi, -pi) are NOT wrapped.
MI_HD float wrap_to_pi(float a) {
    MI_NO_CONTRACT
    const float TWO_PI = 6.283185307179586f, PI_ = 3.141592653589793f;
    a = fmodf(a, TWO_PI);
    a -= TWO_PI * ((a > PI_) ? 1.f : 0.f);
    return a;
}
// counter-based draws: reset stream keyed by (env, episode), per-step stream keyed by (env, step | 0x80000000)
MI_HD float anymal_rand_step(uint32_t seed, uint32_t genv, uint32_t step, uint32_t k) {
    return uniform01(seed ^ 0x5bd1e995u, genv, step, k);
}

// pre_physics_step, one decimation iteration (:443-446)
MI_HD void anymal_pd_torques(const AnymalParams& p, const float* actions, const float* q, const float* qd, float* tau) {
    MI_NO_CONTRACT
    for (int d = 0; d < kAnymalDof; ++d) {
        const float t = p.kp * (p.action_scale * actions[d] + p.default_dof_pos[d] - q[d]) - p.kd * qd[d];
        tau[d] = fminf(fmaxf(t, -p.torque_limit), p.torque_limit);
    }
}

// height scan: 140 points of a 1 m x 1.6 m grid (:487-498), yaw-rotated about the base (:676-681), sampled with the
// reference's "min of two grid neighbours" rule (:527-536).  p_idx = ix * 10 + iy (torch.meshgrid 'ij' flatten).
MI_HD float anymal_height_at(const AnymalTerrainDesc& T, const float* yaw_quat, const float* root_pos, int p_idx) {
    MI_NO_CONTRACT
    const int ix = p_idx / 10, iy = p_idx % 10;
    const int xi = (ix < 7) ? (ix - 8) : (ix - 5);     // -8..-2, 2..8
    const int yi = (iy < 5) ? (iy - 5) : (iy - 4);     // -5..-1, 1..5
    const float pl[3] = {0.1f * (float)xi, 0.1f * (float)yi, 0.f};
    float pw[3];
    quat_apply(yaw_quat, pl, pw);
    float px = pw[0] + root_pos[0], py = pw[1] + root_pos[1];
    px += T.border; py += T.border;
    long long gx = (long long)(px / T.hscale), gy = (long long)(py / T.hscale);   // .long() truncates toward zero
    gx = gx < 0 ? 0 : (gx > T.rows - 2 ? T.rows - 2 : gx);
    gy = gy < 0 ? 0 : (gy > T.cols - 2 ? T.cols - 2 : gy);
    const short h1 = T.hs[gx * T.cols + gy], h2 = T.hs[(gx + 1) * T.cols + gy + 1];
    return (float)(h1 < h2 ? h1 : h2) * T.vscale;
}

// ------------------------------------------------------------------------------------------------------------------
// Anymal on flat ground (reference isaacgymenvs/tasks/anymal.py): PD position drives, 48 observations, 3 reward terms.
//   pre_physics_step :226-229      compute_anymal_reward       :311-351 (@torch.jit.script)
//   reset_idx        :274-301      compute_anymal_observations :354-386 (@torch.jit.script)
constexpr int kAnymalFlatObs = 48;    // 3+3+3+3+12+12+12 (anymal.py:376-384)

struct AnymalFlatParams {  // mirrors MiAnymalFlatParams in include/mi_engine.h (same layout)
    float lin_vel_scale, ang_vel_scale, dof_pos_scale, dof_vel_scale, action_scale;   // :47-51
    float rew_lin_vel_xy, rew_ang_vel_z, rew_torque;       // learn.*RewardScale, already multiplied by dt (:96-97)
    float command_x[2], command_y[2], command_yaw[2];      // randomCommandVelocityRanges (:64-66)
    float base_init_state[13];                              // :74-80
    float default_dof_pos[kAnymalDof];                      // :133-136
    float kp, kd, torque_limit;                             // control.stiffness / damping (:203-206); URDF effort limit
    int max_episode_length;                                 // :92
    float clip_actions;
};

// compute_anymal_observations (anymal.py:354-386); gravity_vec = (0, 0, -1) (:143).  Note the reference projects gravity
// with quat_rotate, not quat_rotate_inverse (:372) -- kept.
MI_HD void anymal_flat_observations(const AnymalFlatParams& p, const float* root, const float* cmd, const float* q, const float* qd,
                                    const float* act, float* obs) {
    MI_NO_CONTRACT
    float blv[3], bav[3], pg[3];
    const float gvec[3] = {0.f, 0.f, -1.f};
    quat_rotate_s(root + 3, root + 7, -1.f, blv);
    quat_rotate_s(root + 3, root + 10, -1.f, bav);
    quat_rotate_s(root + 3, gvec, 1.f, pg);
    for (int i = 0; i < 3; ++i) {
        obs[i] = blv[i] * p.lin_vel_scale;
        obs[3 + i] = bav[i] * p.ang_vel_scale;
        obs[6 + i] = pg[i];
    }
    obs[9] = cmd[0] * p.lin_vel_scale; obs[10] = cmd[1] * p.lin_vel_scale; obs[11] = cmd[2] * p.ang_vel_scale;
    for (int d = 0; d < kAnymalDof; ++d) {
        obs[12 + d] = (q[d] - p.default_dof_pos[d]) * p.dof_pos_scale;
        obs[24 + d] = qd[d] * p.dof_vel_scale;
        obs[36 + d] = act[d];
    }
}

// compute_anymal_reward (anymal.py:311-351).  contact = net contact force of base / knee bodies, [3] each.
MI_HD void anymal_flat_reward(const AnymalFlatParams& p, const float* root, const float* cmd, const float* torques,
                              const float* base_contact, const float (*knee_contact)[3], long long episode_length, float* rew,
                              long long* reset) {
    MI_NO_CONTRACT
    float blv[3], bav[3];
    quat_rotate_s(root + 3, root + 7, -1.f, blv);
    quat_rotate_s(root + 3, root + 10, -1.f, bav);
    const float ex = cmd[0] - blv[0], ey = cmd[1] - blv[1], ez = cmd[2] - bav[2];
    const float lin_vel_error = ex * ex + ey * ey;
    const float ang_vel_error = ez * ez;
    const float rew_lin = expf(-lin_vel_error / 0.25f) * p.rew_lin_vel_xy;
    const float rew_ang = expf(-ang_vel_error / 0.25f) * p.rew_ang_vel_z;
    float tsq = 0.f;
    for (int d = 0; d < kAnymalDof; ++d) tsq += torques[d] * torques[d];
    const float total = (rew_lin + rew_ang) + tsq * p.rew_torque;
    *rew = fmaxf(total, 0.f);
    auto norm3 = [](const float* f) MI_LAMBDA { return sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]); };
    bool rs = norm3(base_contact) > 1.f;
    for (int k = 0; k < 4; ++k) rs = rs || (norm3(knee_contact[k]) > 1.f);
    rs = rs || (episode_length >= (long long)p.max_episode_length - 1);   // no terminal reward for time-outs
    *reset = rs ? 1 : 0;
}

// reset_idx (anymal.py:274-301) with the engine's counter-based draws (one stream per (env, episode))
MI_HD void anymal_flat_reset(const AnymalFlatParams& p, uint32_t seed, uint32_t genv, uint32_t ep, float* root, float* q, float* qd,
                             float* cmd) {
    MI_NO_CONTRACT
    for (int d = 0; d < kAnymalDof; ++d) {
        q[d] = p.default_dof_pos[d] * ((1.5f - 0.5f) * uniform01(seed, genv, ep, (uint32_t)d) + 0.5f);
        qd[d] = (0.1f - (-0.1f)) * uniform01(seed, genv, ep, (uint32_t)(kAnymalDof + d)) + (-0.1f);
    }
    for (int k = 0; k < 13; ++k) root[k] = p.base_init_state[k];
    cmd[0] = (p.command_x[1] - p.command_x[0]) * uniform01(seed, genv, ep, 2 * kAnymalDof + 0) + p.command_x[0];
    cmd[1] = (p.command_y[1] - p.command_y[0]) * uniform01(seed, genv, ep, 2 * kAnymalDof + 1) + p.command_y[0];
    cmd[2] = (p.command_yaw[1] - p.command_yaw[0]) * uniform01(seed, genv, ep, 2 * kAnymalDof + 2) + p.command_yaw[0];
}

}  // namespace mi

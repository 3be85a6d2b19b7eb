// quadcopter.hpp -- Quadcopter task maths for one env (reference isaacgymenvs/tasks/quadcopter.py).
//   reset_idx :254-274   pre_physics_step :276-292   compute_observations :320-331
//   compute_quadcopter_reward :348-386 (@torch.jit.script)
#pragma once
#include "../core/quat.hpp"
#include "../core/rng.hpp"

namespace mi {

constexpr int kQuadDof = 8, kQuadAct = 12, kQuadObs = 21, kQuadRotors = 4;

struct QuadcopterParams {  // mirrors MiQuadcopterParams in include/mi_engine.h (same layout)
    float max_episode_length;       // env.maxEpisodeLength (:45)
    float dt;                       // sim dt
    float dof_lower[kQuadDof], dof_upper[kQuadDof];   // asset joint limits (+-30 deg, :176,193)
    float max_thrust;               // :88
    float dof_action_speed_scale;   // 8 pi (:283)
    float thrust_action_speed_scale;  // 200 (:287)
    float drive_stiffness, drive_damping;   // DOF_MODE_POS, 1000 / 0 (:236-238)
    float max_angular_velocity;     // asset option, 4 pi (:208)
    float init_height;              // default_pose.p.z = 1 (:226)
    float clip_actions;
};

// compute_quadcopter_reward (quadcopter.py:348-386); reset_buf_in is not read by the reference beyond its shape
MI_HD void quadcopter_reward(const float* root, long long progress, float max_episode_length, float* reward, long long* reset) {
    MI_NO_CONTRACT
    const float x = root[0], y = root[1], z = root[2];
    const float target_dist = sqrtf((x * x + y * y) + (1.f - z) * (1.f - z));
    const float pos_reward = 1.0f / (1.0f + target_dist * target_dist);
    const float zaxis[3] = {0.f, 0.f, 1.f};
    float ups[3];
    quat_rotate_s(root + 3, zaxis, 1.f, ups);                 // quat_axis(root_quats, 2)
    const float tiltage = fabsf(1.f - ups[2]);
    const float up_reward = 1.0f / (1.0f + tiltage * tiltage);
    const float spinnage = fabsf(root[12]);
    const float spinnage_reward = 1.0f / (1.0f + spinnage * spinnage);
    *reward = pos_reward + pos_reward * (up_reward + spinnage_reward);
    long long die = 0;
    if (target_dist > 3.0f) die = 1;
    if (z < 0.3f) die = 1;
    *reset = ((float)progress >= max_episode_length - 1.f) ? 1 : die;
}

// compute_observations (:320-331)
MI_HD void quadcopter_observations(const float* root, const float* q, float* obs) {
    MI_NO_CONTRACT
    obs[0] = (0.0f - root[0]) / 3.f; obs[1] = (0.0f - root[1]) / 3.f; obs[2] = (1.0f - root[2]) / 3.f;
    for (int k = 0; k < 4; ++k) obs[3 + k] = root[3 + k];
    for (int k = 0; k < 3; ++k) { obs[7 + k] = root[7 + k] / 2.f; obs[10 + k] = root[10 + k] / 3.141592653589793f; }
    for (int d = 0; d < kQuadDof; ++d) obs[13 + d] = q[d];
}

// reset_idx (:254-274) with the engine's counter-based draws
MI_HD void quadcopter_reset(const QuadcopterParams& p, uint32_t seed, uint32_t genv, uint32_t ep, float* root, float* q, float* qd) {
    MI_NO_CONTRACT
    for (int k = 0; k < 13; ++k) root[k] = 0.f;
    root[2] = p.init_height; root[6] = 1.f;                                                       // initial_root_states
    root[0] += (1.5f - (-1.5f)) * uniform01(seed, genv, ep, 0) + (-1.5f);
    root[1] += (1.5f - (-1.5f)) * uniform01(seed, genv, ep, 1) + (-1.5f);
    root[2] += (1.5f - (-0.2f)) * uniform01(seed, genv, ep, 2) + (-0.2f);
    for (int d = 0; d < kQuadDof; ++d) {
        q[d] = (0.2f - (-0.2f)) * uniform01(seed, genv, ep, 3 + d) + (-0.2f);
        qd[d] = 0.f;
    }
}

}  // namespace mi

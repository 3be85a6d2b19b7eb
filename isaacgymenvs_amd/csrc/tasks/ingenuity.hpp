// ingenuity.hpp -- Ingenuity task maths for one env (reference isaacgymenvs/tasks/ingenuity.py).
//   set_targets :284-293   reset_idx :295-319   pre_physics_step :321-354   compute_observations :386-391
//   compute_ingenuity_reward :407-445 (@torch.jit.script) -- also the stand-alone mi_compute_ingenuity_reward of kernels_jit_twins.hip
#pragma once
#include "../core/quat.hpp"
#include "../core/rng.hpp"

namespace mi {

constexpr int kIngDof = 4, kIngAct = 6, kIngObs = 13, kIngRotors = 2, kIngBodies = 6;   // bodies_per_env counts the marker actor (:61-62)

struct IngenuityParams {   // mirrors MiIngenuityParams in include/mi_engine.h (same layout)
    float max_episode_length;          // env.maxEpisodeLength (:45)
    float dt;                          // sim dt
    float thrust_upper_limit;          // 2000 (:91)
    float thrust_lateral_component;    // 0.2 (:92)
    float thrust_action_speed_scale;   // 2000 (:337)
    float max_angular_velocity;        // asset option, 4 pi (:248)
    float init_height;                 // default_pose.p.z = 1 (:254)
    float rotor_speed;                 // 50: dof_velocities[:, 1] = -50, [:, 3] = 50 at every reset (:298-299)
    int target_period;                 // 500: a new target whenever progress_buf % 500 == 0 (:324)
    float clip_actions;
};

// draw slots of one episode: 0-2 root position, 3-5 the target drawn inside reset_idx, 8 + 3 k ... the k-th periodic target
MI_HD void ingenuity_target(uint32_t seed, uint32_t genv, uint32_t ep, uint32_t slot, float* target) {   // set_targets (:284-288)
    MI_NO_CONTRACT
    target[0] = uniform01(seed, genv, ep, slot) * 10.f - 5.f;
    target[1] = uniform01(seed, genv, ep, slot + 1) * 10.f - 5.f;
    target[2] = uniform01(seed, genv, ep, slot + 2) + 1.f;
}

// reset_idx (:295-319): root at the initial pose plus a random offset; the dof positions are left as they are
MI_HD void ingenuity_reset_root(const IngenuityParams& p, uint32_t seed, uint32_t genv, uint32_t ep, float* root) {
    MI_NO_CONTRACT
    for (int k = 0; k < 13; ++k) root[k] = 0.f;
    root[2] = p.init_height; root[6] = 1.f;
    root[0] += (1.5f - (-1.5f)) * uniform01(seed, genv, ep, 0) + (-1.5f);
    root[1] += (1.5f - (-1.5f)) * uniform01(seed, genv, ep, 1) + (-1.5f);
    root[2] += (1.5f - (-0.2f)) * uniform01(seed, genv, ep, 2) + (-0.2f);
}

// compute_observations (:386-391)
MI_HD void ingenuity_observations(const float* root, const float* target, float* obs) {
    MI_NO_CONTRACT
    for (int k = 0; k < 3; ++k) obs[k] = (target[k] - root[k]) / 3.f;
    for (int k = 0; k < 4; ++k) obs[3 + k] = root[3 + k];
    for (int k = 0; k < 3; ++k) { obs[7 + k] = root[7 + k] / 2.f; obs[10 + k] = root[10 + k] / 3.141592653589793f; }
}

// compute_ingenuity_reward (:407-445); root_linvels and reset_buf are not read by the reference
MI_HD void ingenuity_reward(const float* pos, const float* target, const float* quat, const float* angvel, long long progress,
                            float max_episode_length, float* reward, long long* reset) {
    MI_NO_CONTRACT
    const float d[3] = {target[0] - pos[0], target[1] - pos[1], target[2] - pos[2]};
    const float target_dist = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float pos_reward = 1.0f / (1.0f + target_dist * target_dist);
    const float zaxis[3] = {0.f, 0.f, 1.f};
    float ups[3];
    quat_rotate_s(quat, zaxis, 1.f, ups);
    const float tiltage = fabsf(1.f - ups[2]);
    const float up_reward = 5.0f / (1.0f + tiltage * tiltage);
    const float spinnage = fabsf(angvel[2]);
    const float spinnage_reward = 1.0f / (1.0f + spinnage * spinnage);
    *reward = pos_reward + pos_reward * (up_reward + spinnage_reward);
    long long die = 0;
    if (target_dist > 8.0f) die = 1;
    if (pos[2] < 0.5f) die = 1;
    *reset = ((float)progress >= max_episode_length - 1.f) ? 1 : die;
}

}  // namespace mi

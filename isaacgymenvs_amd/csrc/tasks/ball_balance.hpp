// ball_balance.hpp -- BallBalance task maths for one env (reference isaacgymenvs/tasks/ball_balance.py).
//   reset_idx :349-393   pre_physics_step :395-413   compute_observations :322-337
//   compute_bbot_reward :459-476 (@torch.jit.script) -- also the stand-alone mi_compute_bbot_reward of kernels_jit_twins.hip
#pragma once
#include "../core/quat.hpp"
#include "../core/rng.hpp"

namespace mi {

constexpr int kBbotDof = 6, kBbotAct = 3, kBbotObs = 24, kBbotSensors = 3;

struct BallBalanceParams {   // mirrors MiBallBalanceParams in include/mi_engine.h (same layout)
    float max_episode_length;          // env.maxEpisodeLength (:59)
    float dt;                          // sim dt
    float action_speed_scale;          // env.actionSpeedScale (:60)
    float dof_lower[kBbotDof], dof_upper[kBbotDof];   // joint limits of the generated asset (:189-190, :212-213)
    float tray_height;                 // bbot_pose.p.z (:251-252)
    float ball_init_pos[3];            // ball_pose (:303-306)
    float clip_actions;
    // ---- physics (core/bbot_engine.hpp BbotPhys, same order)
    float pin_stiffness, pin_damping;
    float drive_kp, drive_kd;
    int actuated_mask;
    float ball_radius, ball_mass, ball_inertia, mu;
    float tray_radius, tray_half;
    float pin_offset[3];
    float pin_target[3][3];
    float sensor_pos[3][3];
};

// compute_bbot_reward (:459-476); tray_positions is not read by the reference
MI_HD void bbot_reward(const float* ball_pos, const float* ball_vel, float ball_radius, long long reset_in, long long progress,
                       float max_episode_length, float* reward, long long* reset) {
    MI_NO_CONTRACT
    const float ball_dist = sqrtf((ball_pos[0] * ball_pos[0] + (ball_pos[2] - 0.7f) * (ball_pos[2] - 0.7f)) + ball_pos[1] * ball_pos[1]);
    const float ball_speed = sqrtf((ball_vel[0] * ball_vel[0] + ball_vel[1] * ball_vel[1]) + ball_vel[2] * ball_vel[2]);
    const float pos_reward = 1.0f / (1.0f + ball_dist);
    const float speed_reward = 1.0f / (1.0f + ball_speed);
    *reward = pos_reward * speed_reward;
    long long r = ((float)progress >= max_episode_length - 1.f) ? 1 : reset_in;
    if (ball_pos[2] < ball_radius * 1.5f) r = 1;
    *reset = r;
}

// reset_idx (:349-393) with the engine's counter-based draws: the ball is dropped towards the tray from a random place
MI_HD void bbot_reset_ball(const BallBalanceParams& p, uint32_t seed, uint32_t genv, uint32_t ep, float* ball) {
    MI_NO_CONTRACT
    const float min_d = 0.001f, max_d = 0.5f, min_height = 1.0f, max_height = 2.0f, min_hs = 0.f, max_hs = 5.f;   // :355-360
    const float dist = (max_d - min_d) * uniform01(seed, genv, ep, 0) + min_d;
    const float angle = (3.14159265358979323846f - (-3.14159265358979323846f)) * uniform01(seed, genv, ep, 1) + (-3.14159265358979323846f);   // torch_random_dir_2
    const float dir[2] = {cosf(angle), sinf(angle)};
    const float speedscale = (dist - min_d) / (max_d - min_d);
    const float hspeed = (max_hs - min_hs) * uniform01(seed, genv, ep, 2) + min_hs;
    for (int k = 0; k < 13; ++k) ball[k] = 0.f;
    ball[0] = dist * dir[0]; ball[1] = dist * dir[1];
    ball[2] = (max_height - min_height) * uniform01(seed, genv, ep, 3) + min_height;
    ball[6] = 1.f;
    ball[7] = -speedscale * hspeed * dir[0]; ball[8] = -speedscale * hspeed * dir[1];
    ball[9] = -5.0f;                                                            // -torch_rand_float(5.0, 5.0)
}

// compute_observations (:322-337).  sensor: [3][6] force xyz, torque xyz per sensor
MI_HD void bbot_observations(const float* q, const float* qd, const float* ball, const float* sensor, float* obs) {
    MI_NO_CONTRACT
    for (int k = 0; k < 3; ++k) {
        obs[k] = q[1 + 2 * k]; obs[3 + k] = qd[1 + 2 * k];                      // actuated dofs 1, 3, 5
        obs[6 + k] = ball[k]; obs[9 + k] = ball[7 + k];
        obs[12 + k] = sensor[6 * k + 0] / 20.f;                                 // sensor_forces[..., 0]: the x component of each sensor
        obs[15 + k] = sensor[6 * k + 3] / 20.f;                                 // sensor_torques[..., 0] / [..., 1] / [..., 2]
        obs[18 + k] = sensor[6 * k + 4] / 20.f;
        obs[21 + k] = sensor[6 * k + 5] / 20.f;
    }
}

}  // namespace mi

// locomotion.hpp -- observation / reward / reset maths shared by the Ant and Humanoid tasks.
//
// Scalar (one env) restatements of the reference's @torch.jit.script functions, written so the fused step kernel
// and the stand-alone C-ABI entry points run the same code:
//   compute_ant_observations        reference isaacgymenvs/tasks/ant.py:374-408
//   compute_ant_reward              reference isaacgymenvs/tasks/ant.py:325-371
//   compute_humanoid_observations   reference isaacgymenvs/tasks/humanoid.py:378-413
//   compute_humanoid_reward         reference isaacgymenvs/tasks/humanoid.py:323-375
//   compute_heading_and_up / compute_rot   reference isaacgymenvs/utils/torch_jit_utils.py:247-276
//   reset_idx                       reference ant.py:252-279 / humanoid.py:253-279
// Expression order follows the reference line by line (fp32, contraction off) so results agree to rounding.
#pragma once
#include "../core/quat.hpp"
#include "../core/rng.hpp"

namespace mi {

constexpr int kMaxDof = 32;

struct LocoParams {  // mirrors MiLocoParams in include/mi_engine.h (same layout)
    float dt;                      // cfg sim.dt
    float dof_vel_scale;           // env.dofVelocityScale
    float contact_force_scale;     // env.contactForceScale
    float angular_velocity_scale;  // env.angularVelocityScale (humanoid)
    float power_scale;             // env.powerScale
    float heading_weight, up_weight;
    float actions_cost, energy_cost, joints_at_limit_cost;
    float death_cost, termination_height;
    float max_episode_length;      // passed as float like the reference (ant.py:342)
    float clip_actions;            // env.clipActions (vec_task.py:374)
    float max_motor_effort;        // humanoid.py:170
    float start_height;            // actor spawn z (ant.py:164 0.44, humanoid.py:179 1.34)
    float gear[kMaxDof];        // motor efforts in actuator-file order, applied in dof order like the reference
    float dof_lower[kMaxDof], dof_upper[kMaxDof];
    float initial_dof_pos[kMaxDof];
    float targets[3];              // ant.py:110 [1000,0,0]
    float inv_start_rot[4];        // conj(start rotation)
    float basis_vec0[3], basis_vec1[3];
    float reset_pos_noise, reset_vel_noise;  // 0.2 / 0.1 (ant.py:257-258)
};

// obs layout: [z, vel_loc3, angvel_loc3(*s), yaw, roll, angle_to_target, up_proj, heading_proj,
//              dof_pos_scaled ND, dof_vel*s ND, (HUM: dof_force*s ND), sensors*s NSV, actions ND]
template <int ND, int NSV, bool HUM>
struct Loco {
    static constexpr int NOBS = 12 + ND * (HUM ? 4 : 3) + NSV;

    // ---- the parts of the observation row (also what a kernel that deals an env's columns out to several lanes calls: the four-lane post
    // kernel measured in round 3, step_kernels.hpp)
    // the 12 root-derived columns; also potentials (ant.py:390-393) and the up / heading vectors
    MI_HD static void obs_root(const LocoParams& p, const float* root, const float* targets, float potentials_in, const float* inv_start_rot,
                               const float* basis0, const float* basis1, float* obs12, float* potentials_out, float* prev_potentials_out,
                               float* up_vec, float* heading_vec) {
        MI_NO_CONTRACT
        const float* pos = root;
        const float* rot = root + 3;
        const float* vel = root + 7;
        const float* angvel = root + 10;
        float to_target[3] = {targets[0] - pos[0], targets[1] - pos[1], 0.f};
        *prev_potentials_out = potentials_in;
        const float nrm = sqrtf((to_target[0] * to_target[0] + to_target[1] * to_target[1]) + to_target[2] * to_target[2]);
        *potentials_out = -nrm / p.dt;
        // compute_heading_and_up (torch_jit_utils.py:247-262)
        const float nc = fmaxf(nrm, 1e-9f);
        const float target_dirs[3] = {to_target[0] / nc, to_target[1] / nc, to_target[2] / nc};
        float torso_quat[4];
        quat_mul(rot, inv_start_rot, torso_quat);
        quat_rotate_s(torso_quat, basis1, 1.f, up_vec);
        quat_rotate_s(torso_quat, basis0, 1.f, heading_vec);
        const float up_proj = up_vec[2];
        const float heading_proj = (heading_vec[0] * target_dirs[0] + heading_vec[1] * target_dirs[1]) + heading_vec[2] * target_dirs[2];
        // compute_rot (:265-276)
        float vel_loc[3], angvel_loc[3];
        quat_rotate_s(torso_quat, vel, -1.f, vel_loc);
        quat_rotate_s(torso_quat, angvel, -1.f, angvel_loc);
        float roll, yaw;
        euler_roll_yaw(torso_quat, &roll, &yaw);
        const float walk_target_angle = atan2f(targets[2] - pos[2], targets[0] - pos[0]);
        float angle_to_target = walk_target_angle - yaw;
        float ang_scale = 1.f;
        if constexpr (HUM) {  // humanoid.py:402-404
            roll = normalize_angle(roll);
            yaw = normalize_angle(yaw);
            angle_to_target = normalize_angle(angle_to_target);
            ang_scale = p.angular_velocity_scale;
        }
        obs12[0] = pos[2];
        obs12[1] = vel_loc[0]; obs12[2] = vel_loc[1]; obs12[3] = vel_loc[2];
        obs12[4] = angvel_loc[0] * ang_scale; obs12[5] = angvel_loc[1] * ang_scale; obs12[6] = angvel_loc[2] * ang_scale;
        obs12[7] = yaw; obs12[8] = roll; obs12[9] = angle_to_target; obs12[10] = up_proj; obs12[11] = heading_proj;
    }
    // columns of one dof: scaled position, scaled velocity, (Humanoid) scaled joint force
    MI_HD static void obs_dof(const LocoParams& p, float dof_pos, float dof_vel, float dof_force, float lower, float upper, float* pos_scaled,
                              float* vel_scaled, float* force_scaled) {
        MI_NO_CONTRACT
        *pos_scaled = unscale(dof_pos, lower, upper);
        *vel_scaled = dof_vel * p.dof_vel_scale;
        *force_scaled = dof_force * p.contact_force_scale;
    }
    static constexpr int COL_POS = 12, COL_VEL = 12 + ND, COL_FORCE = 12 + 2 * ND, COL_SENS = 12 + ND * (HUM ? 3 : 2), COL_ACT = COL_SENS + NSV;

    MI_HD static void observations(const LocoParams& p, const float* root, const float* targets, float potentials_in,
                                   const float* inv_start_rot, const float* dof_pos, const float* dof_vel,
                                   const float* dof_force, const float* lower, const float* upper,
                                   const float* sensors, const float* actions, const float* basis0,
                                   const float* basis1, float* obs, float* potentials_out,
                                   float* prev_potentials_out, float* up_vec, float* heading_vec) {
        MI_NO_CONTRACT
        obs_root(p, root, targets, potentials_in, inv_start_rot, basis0, basis1, obs, potentials_out, prev_potentials_out, up_vec, heading_vec);
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float ps, vs, fs;
            obs_dof(p, dof_pos[d], dof_vel[d], HUM ? dof_force[d] : 0.f, lower[d], upper[d], &ps, &vs, &fs);
            obs[COL_POS + d] = ps;
            obs[COL_VEL + d] = vs;
            if constexpr (HUM) obs[COL_FORCE + d] = fs;
        });
        sfor<NSV>([&](auto K) MI_LAMBDA { obs[COL_SENS + K] = sensors[K] * p.contact_force_scale; });
        sfor<ND>([&](auto D) MI_LAMBDA { obs[COL_ACT + D] = actions[D]; });
    }

    MI_HD static float unscale(float x, float lo, float up) {  // torch_jit_utils.py:238-240
        MI_NO_CONTRACT
        return (2.0f * x - up - lo) / (up - lo);
    }

    // ---- reward.  The three sums over the dofs (torch.sum(..., dim=-1) in the reference, whose order is torch's business) are DEFINED here
    // as four strided partial sums, dof d into partial d % 4 in increasing d, combined as (p0 + p1) + (p2 + p3): an order that four lanes
    // per env reproduce with an xor-1, then an xor-2 shuffle, so every form -- kernels, CPU backend, stand-alone twins -- agrees bit for bit.
    struct DofSums { float actions = 0.f, electricity = 0.f, at_limit = 0.f; };
    MI_HD static void reward_dof(const LocoParams& p, float action, float pos_scaled, float vel_scaled, float gear, DofSums& s) {
        MI_NO_CONTRACT
        s.actions += action * action;
        if constexpr (HUM) {
            const float ratio = gear / p.max_motor_effort;
            const float ap = fabsf(pos_scaled);
            const float scaled = p.joints_at_limit_cost * (ap - 0.98f) / 0.02f;
            s.at_limit += ((ap > 0.98f) ? 1.f : 0.f) * scaled * ratio;
            s.electricity += fabsf(action * vel_scaled) * ratio;
        } else {
            s.electricity += fabsf(action * vel_scaled);
            s.at_limit += (pos_scaled > 0.99f) ? 1.f : 0.f;
        }
    }
    MI_HD static void reward_total(const LocoParams& p, float height, float up_proj, float heading_proj, const DofSums& s, long long reset_in,
                                   long long progress, float potentials, float prev_potentials, float* reward_out, long long* reset_out) {
        MI_NO_CONTRACT
        const float heading_reward = (heading_proj > 0.8f) ? p.heading_weight : p.heading_weight * heading_proj / 0.8f;
        const float up_reward = (up_proj > 0.93f) ? (0.f + p.up_weight) : 0.f;
        const float alive_reward = HUM ? 2.0f : 0.5f;
        const float progress_reward = potentials - prev_potentials;
        float total;
        if constexpr (HUM)
            total = progress_reward + alive_reward + up_reward + heading_reward - p.actions_cost * s.actions -
                    p.energy_cost * s.electricity - s.at_limit;
        else
            total = progress_reward + alive_reward + up_reward + heading_reward - p.actions_cost * s.actions -
                    p.energy_cost * s.electricity - s.at_limit * p.joints_at_limit_cost;
        const bool fallen = height < p.termination_height;
        if (fallen) total = p.death_cost;
        long long reset = fallen ? 1 : reset_in;
        if ((float)progress >= p.max_episode_length - 1.f) reset = 1;
        *reward_out = total;
        *reset_out = reset;
    }
    MI_HD static void reward(const LocoParams& p, const float* obs, long long reset_in, long long progress,
                             const float* actions, float potentials, float prev_potentials, float* reward_out,
                             long long* reset_out) {
        MI_NO_CONTRACT
        DofSums part[4];
        for (int d = 0; d < ND; ++d) reward_dof(p, actions[d], obs[COL_POS + d], obs[COL_VEL + d], p.gear[d], part[d & 3]);
        DofSums s;
        s.actions = (part[0].actions + part[1].actions) + (part[2].actions + part[3].actions);
        s.electricity = (part[0].electricity + part[1].electricity) + (part[2].electricity + part[3].electricity);
        s.at_limit = (part[0].at_limit + part[1].at_limit) + (part[2].at_limit + part[3].at_limit);
        reward_total(p, obs[0], obs[10], obs[11], s, reset_in, progress, potentials, prev_potentials, reward_out, reset_out);
    }

    // reset_idx, the draws of one dof (torch_rand_float + tensor_clamp, ant.py:257-263)
    MI_HD static void reset_dof(const LocoParams& p, uint32_t seed, uint32_t genv, uint32_t episode, int d, float initial, float lower, float upper,
                                float* dof_pos, float* dof_vel) {
        MI_NO_CONTRACT
        const float up = p.reset_pos_noise, lo = -p.reset_pos_noise;
        const float rp = (up - lo) * uniform01(seed, genv, episode, (uint32_t)d) + lo;        // torch_rand_float
        const float vu = p.reset_vel_noise, vl = -p.reset_vel_noise;
        const float rv = (vu - vl) * uniform01(seed, genv, episode, (uint32_t)(ND + d)) + vl;
        *dof_pos = fmaxf(fminf(initial + rp, upper), lower);  // tensor_clamp
        *dof_vel = rv;
    }
    MI_HD static float reset_potential(const LocoParams& p, const float* initial_root) {
        MI_NO_CONTRACT
        const float tx = p.targets[0] - initial_root[0], ty = p.targets[1] - initial_root[1];
        return -sqrtf((tx * tx + ty * ty) + 0.f) / p.dt;
    }
    // reset_idx for one env: returns the new dof state and potentials; root := initial root state
    MI_HD static void reset(const LocoParams& p, uint32_t seed, uint32_t genv, uint32_t episode, const float* initial_root,
                            float* root, float* dof_pos, float* dof_vel, float* potentials, float* prev_potentials) {
        MI_NO_CONTRACT
        for (int d = 0; d < ND; ++d) reset_dof(p, seed, genv, episode, d, p.initial_dof_pos[d], p.dof_lower[d], p.dof_upper[d], &dof_pos[d], &dof_vel[d]);
        for (int k = 0; k < 13; ++k) root[k] = initial_root[k];
        const float pp = reset_potential(p, initial_root);
        *prev_potentials = pp;
        *potentials = pp;
    }
};

// ------------------------------------------------------------------ cartpole (reference tasks/cartpole.py)
struct CartpoleParams {  // mirrors MiCartpoleParams
    float reset_dist;          // env.resetDist
    float max_push_effort;     // env.maxEffort
    float max_episode_length;  // 500 (cartpole.py:44)
    float clip_actions;
};
// compute_cartpole_reward, cartpole.py:180-196
MI_HD void cartpole_reward(const CartpoleParams& p, float pole_angle, float pole_vel, float cart_vel, float cart_pos,
                           long long reset_in, long long progress, float* reward_out, long long* reset_out) {
    MI_NO_CONTRACT
    float reward = 1.0f - pole_angle * pole_angle - 0.01f * fabsf(cart_vel) - 0.005f * fabsf(pole_vel);
    const float HALF_PI = 1.5707963267948966f;
    if (fabsf(cart_pos) > p.reset_dist) reward = -2.0f;
    if (fabsf(pole_angle) > HALF_PI) reward = -2.0f;
    long long reset = (fabsf(cart_pos) > p.reset_dist) ? 1 : reset_in;
    if (fabsf(pole_angle) > HALF_PI) reset = 1;
    if ((float)progress >= p.max_episode_length - 1.f) reset = 1;
    *reward_out = reward;
    *reset_out = reset;
}
// reset_idx, cartpole.py:144-157
MI_HD void cartpole_reset(uint32_t seed, uint32_t genv, uint32_t episode, float* dof_pos, float* dof_vel) {
    MI_NO_CONTRACT
    for (int d = 0; d < 2; ++d) {
        dof_pos[d] = 0.2f * (uniform01(seed, genv, episode, (uint32_t)d) - 0.5f);
        dof_vel[d] = 0.5f * (uniform01(seed, genv, episode, (uint32_t)(2 + d)) - 0.5f);
    }
}

}  // namespace mi

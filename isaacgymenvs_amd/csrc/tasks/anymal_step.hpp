// anymal_step.hpp -- the AnymalTerrain / Anymal control step for ONE env, host + device: what kernels_anymal.hip runs per lane and
// cpu/cpu_anymal.cpp runs per loop iteration (reference isaacgymenvs/tasks/anymal_terrain.py:384-485, anymal.py:231-301).
//
// Everything that crosses envs goes through a reduction policy RED supplied by the caller:
//     red.cmdnorm(v, x)                             sum(cx^2 + cy^2) over the envs that reset this step -> v.ep_stats[15]
//     red.extras(v, st_sums, st_cnt, level)         extras["episode"] partial sums -> v.ep_stats[0..14]
//     red.episode(v, e, valid, rew, reset, progress) job statistics (View::stats) + the env's running return
// On the device these are wave shuffles + one atomic per wave (kernels_anymal.hip DevRed); on the host per-thread accumulators that
// the caller adds up after its loop over envs (cpu/cpu_anymal.cpp HostRed).
#pragma once
#include "../arena.hpp"
#include "anymal.hpp"

namespace mi {

MI_HD float anymal_netf_norm(const View& v, const int e, const int b) {
    MI_NO_CONTRACT
    const int N = v.N;
    const float fx = v.netf[(3 * b) * N + e], fy = v.netf[(3 * b + 1) * N + e], fz = v.netf[(3 * b + 2) * N + e];
    return sqrtf((fx * fx + fy * fy) + fz * fz);
}

// ------------------------------------------------------------------------------------------------ curriculum pre-pass
// update_terrain_level (:431) compares every resetting env's walked distance with
//     torch.norm(self.commands[env_ids, :2]) * max_episode_length_s * 0.25
// where the norm runs over the commands of ALL envs that reset in this step (a batch-coupled quantity in the
// reference).  This pre-pass evaluates the termination condition (:294-300) and accumulates sum(cx^2 + cy^2) over the
// resetting envs into ep_stats[15]; the post pass takes its square root.
template <class RED>
MI_HD void anymal_cmdnorm_env(const View& v, const AnymalParams& p, const int e0, const RED& red) {
    const int N = v.N;
    float acc = 0.f;
    if (e0 < N) {
        const int e = e0;
        bool rs = anymal_netf_norm(v, e, 0) > 1.f;
        if (!p.allow_knee_contacts)
            for (int k = 0; k < 4; ++k) rs = rs || (anymal_netf_norm(v, e, anymal_knee_body(k)) > 1.f);
        if (v.progress[e] + 1 >= (long long)p.max_episode_length - 1) rs = true;
        if (rs) {
            const float cx = v.commands[e], cy = v.commands[N + e];
            acc = cx * cx + cy * cy;
        }
    }
    red.cmdnorm(v, acc);
}

// ------------------------------------------------------------------------------------------------ post_physics_step
// Order of operations = reference anymal_terrain.py:453-485 (including its quirks: the base-frame
// velocities / projected gravity of an env that resets this step are the PRE-reset ones, the yaw-aligned height scan
// uses the POST-reset pose, reset_buf stays 1 after reset_idx (:418), the termination reward looks at the previous
// step's timeout_buf because the base class refreshes it after post_physics_step, vec_task.py:394).
// NSPH3 = 3 * contact spheres of the compiled model (warm-start impulses cleared by a reset).  e0 >= N: a lane past the batch, which
// shadows the last env without storing (the device reductions want full waves).
// OBS_ALL = false (the device kernels with option fused_post): only the observation columns that use PRE-reset quantities (0 .. 8: base velocities and
// projected gravity, which reset_idx does not recompute, :465-472) are written here; commands, dof positions / velocities and actions -- all of them
// in memory in their post-reset form when this pass ends -- are written by anymal_obs_column from the scan kernel's threads, one per (env, column).
template <int NSPH3, class RED, bool OBS_ALL = true>
MI_HD void anymal_post_env(const View& v, const AnymalParams& p, const AnymalTerrainDesc& T, const unsigned step_counter, const int e0, const RED& red) {
    constexpr int ND = kAnymalDof;
    const int N = v.N;
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    float root[13], q[ND], qd[ND], act[ND], tau[ND], last_act[ND], last_qd[ND];
    sfor<13>([&](auto K) MI_LAMBDA { root[K] = v.root[K * N + e]; });
    // (the task's dof-state tensor: as of its last refresh, one sim step behind the physics -- View::dof_api; option dof_state_lag 0: the physics state)
    const float* const dofs = (v.dof_api != nullptr) ? v.dof_api : v.dof;
    sfor<ND>([&](auto K) MI_LAMBDA {
        q[K] = dofs[K * N + e]; qd[K] = dofs[(ND + K) * N + e];
        act[K] = v.actions[K * N + e]; tau[K] = v.tau[K * N + e];
        last_act[K] = v.last_actions[K * N + e]; last_qd[K] = v.last_dof_vel[K * N + e];
    });
    float cmd[4];
    sfor<4>([&](auto K) MI_LAMBDA { cmd[K] = v.commands[K * N + e]; });
    long long progress = v.progress[e] + 1;                                  // :458
    // push_robots every push_interval control steps (:461-462, :437-439): new xy velocity for EVERY env
    if (p.push_interval > 0 && (step_counter % (unsigned)p.push_interval) == 0u) {
        root[7] = 2.f * anymal_rand_step(v.seed, genv, step_counter | 0x80000000u, 0) - 1.f;
        root[8] = 2.f * anymal_rand_step(v.seed, genv, step_counter | 0x80000000u, 1) - 1.f;
        if (valid) { v.root[7 * N + e] = root[7]; v.root[8 * N + e] = root[8]; }
    }
    // prepare quantities (:465-472)
    float base_lin_vel[3], base_ang_vel[3], proj_g[3], fwd[3];
    const float gvec[3] = {0.f, 0.f, -1.f}, fvec[3] = {1.f, 0.f, 0.f};
    quat_rotate_s(root + 3, root + 7, -1.f, base_lin_vel);
    quat_rotate_s(root + 3, root + 10, -1.f, base_ang_vel);
    quat_rotate_s(root + 3, gvec, -1.f, proj_g);
    quat_apply(root + 3, fvec, fwd);
    float rew;
    long long reset;
    float sums[kAnymalSums];
    sfor<kAnymalSums>([&](auto K) MI_LAMBDA { sums[K] = v.episode_sums[K * N + e]; });
    float air[4];
    sfor<4>([&](auto K) MI_LAMBDA { air[K] = v.feet_air_time[K * N + e]; });
    {
        MI_NO_CONTRACT
        const float heading = atan2f(fwd[1], fwd[0]);
        cmd[2] = fminf(fmaxf(0.5f * wrap_to_pi(cmd[3] - heading), -1.f), 1.f);
        // check_termination (:294-300)
        bool rs = anymal_netf_norm(v, e, 0) > 1.f;
        int knee_contacts = 0;
        sfor<4>([&](auto K) MI_LAMBDA { knee_contacts += (anymal_netf_norm(v, e, anymal_knee_body(K)) > 1.f) ? 1 : 0; });
        if (!p.allow_knee_contacts) rs = rs || (knee_contacts > 0);
        if (progress >= (long long)p.max_episode_length - 1) rs = true;
        // compute_reward (:315-382)
        const float dvx = cmd[0] - base_lin_vel[0], dvy = cmd[1] - base_lin_vel[1];
        const float lin_vel_error = dvx * dvx + dvy * dvy;
        const float dwz = cmd[2] - base_ang_vel[2];
        const float ang_vel_error = dwz * dwz;
        float r[kAnymalSums];
        r[0] = expf(-lin_vel_error / 0.25f) * p.rew_lin_vel_xy;
        r[2] = expf(-ang_vel_error / 0.25f) * p.rew_ang_vel_z;
        r[1] = (base_lin_vel[2] * base_lin_vel[2]) * p.rew_lin_vel_z;
        r[3] = (base_ang_vel[0] * base_ang_vel[0] + base_ang_vel[1] * base_ang_vel[1]) * p.rew_ang_vel_xy;
        r[4] = (proj_g[0] * proj_g[0] + proj_g[1] * proj_g[1]) * p.rew_orient;
        const float bh = root[2] - 0.52f;
        r[7] = (bh * bh) * p.rew_base_height;
        float st = 0.f, sa = 0.f, sr = 0.f, sh = 0.f;
        for (int d = 0; d < ND; ++d) {
            st += tau[d] * tau[d];
            const float da = last_qd[d] - qd[d];
            sa += da * da;
            const float dr = last_act[d] - act[d];
            sr += dr * dr;
        }
        r[5] = st * p.rew_torque;
        r[6] = sa * p.rew_joint_acc;
        r[9] = (float)knee_contacts * p.rew_collision;
        int stumbles = 0;
        float air_rew = 0.f;
        sfor<4>([&](auto K) MI_LAMBDA {
            constexpr int b = anymal_foot_body(K);
            const float fx = v.netf[(3 * b) * N + e], fy = v.netf[(3 * b + 1) * N + e], fz = v.netf[(3 * b + 2) * N + e];
            stumbles += ((sqrtf(fx * fx + fy * fy) > 5.f) && (fabsf(fz) < 1.f)) ? 1 : 0;
            const bool contact = fz > 1.f;
            const bool first_contact = (air[K] > 0.f) && contact;
            air[K] += p.dt;
            air_rew += (air[K] - 0.5f) * (first_contact ? 1.f : 0.f);
            air[K] = contact ? 0.f : air[K];
        });
        r[10] = (float)stumbles * p.rew_stumble;
        r[11] = sr * p.rew_action_rate;
        r[8] = air_rew * p.rew_air_time;
        r[8] *= (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.1f) ? 1.f : 0.f;
        sh = fabsf(q[0] - p.default_dof_pos[0]) + fabsf(q[3] - p.default_dof_pos[3]) + fabsf(q[6] - p.default_dof_pos[6]) +
             fabsf(q[9] - p.default_dof_pos[9]);
        r[12] = sh * p.rew_hip;
        // total (:361-366): lin_vel_xy + ang_vel_z + lin_vel_z + ang_vel_xy + orient + base_height + torque + joint_acc +
        //                   collision + action_rate + airTime + hip + stumble, clipped at 0, + termination term
        float total = r[0] + r[2] + r[1] + r[3] + r[4] + r[7] + r[5] + r[6] + r[9] + r[11] + r[8] + r[12] + r[10];
        total = fmaxf(total, 0.f);
        const bool prev_timeout = v.timeout[e] != 0;
        total += p.rew_termination * ((rs && !prev_timeout) ? 1.f : 0.f);
        rew = total;
        reset = rs ? 1 : 0;
        sfor<kAnymalSums>([&](auto K) MI_LAMBDA { sums[K] += r[K]; });
    }
    // ------------------------------------------------------------------ reset_idx for flagged envs (:384-425)
    int ep = v.episode[e];
    int level = v.terrain_levels[e];
    float origin[3] = {v.env_origins[e], v.env_origins[N + e], v.env_origins[2 * N + e]};
    float st_sums[kAnymalSums];
    float st_cnt = 0.f;
    sfor<kAnymalSums>([&](auto K) MI_LAMBDA { st_sums[K] = 0.f; });
    if (reset != 0) {
        MI_NO_CONTRACT
        const uint32_t uep = (uint32_t)ep;
        // update_terrain_level (:427-435) -- skipped on the very first reset (init_done False) and without curriculum
        if (p.curriculum && T.hs != nullptr && ep > 0) {
            const float ddx = root[0] - origin[0], ddy = root[1] - origin[1];
            const float distance = sqrtf(ddx * ddx + ddy * ddy);
            // torch.norm(self.commands[env_ids, :2]): norm over ALL envs resetting this step (anymal_cmdnorm_env)
            const float cn = sqrtf(v.ep_stats[15]);
            level -= (distance < cn * p.max_episode_length_s * 0.25f) ? 1 : 0;
            level += (distance > T.env_length / 2.f) ? 1 : 0;
            level = (level < 0 ? 0 : level) % T.levels;
            const int type = v.terrain_types[e];
            sfor<3>([&](auto K) MI_LAMBDA { origin[K] = T.origins[(level * T.types + type) * 3 + K]; });
        }
        for (int d = 0; d < ND; ++d) {
            const float off = (1.5f - 0.5f) * uniform01(v.seed, genv, uep, (uint32_t)d) + 0.5f;          // torch_rand_float(0.5, 1.5)
            q[d] = p.default_dof_pos[d] * off;
            qd[d] = (0.1f - (-0.1f)) * uniform01(v.seed, genv, uep, (uint32_t)(ND + d)) + (-0.1f);
        }
        sfor<13>([&](auto K) MI_LAMBDA { root[K] = p.base_init_state[K]; });
        if (T.hs != nullptr) {   // custom_origins (:395-399)
            sfor<3>([&](auto K) MI_LAMBDA { root[K] += origin[K]; });
            root[0] += (0.5f - (-0.5f)) * uniform01(v.seed, genv, uep, 2 * ND + 0) + (-0.5f);
            root[1] += (0.5f - (-0.5f)) * uniform01(v.seed, genv, uep, 2 * ND + 1) + (-0.5f);
        }
        cmd[0] = (p.command_x[1] - p.command_x[0]) * uniform01(v.seed, genv, uep, 2 * ND + 2) + p.command_x[0];
        cmd[1] = (p.command_y[1] - p.command_y[0]) * uniform01(v.seed, genv, uep, 2 * ND + 3) + p.command_y[0];
        cmd[3] = (p.command_yaw[1] - p.command_yaw[0]) * uniform01(v.seed, genv, uep, 2 * ND + 4) + p.command_yaw[0];
        const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.25f) ? 1.f : 0.f;   // set small commands to zero (:415)
        sfor<4>([&](auto K) MI_LAMBDA { cmd[K] *= keep; });
        sfor<ND>([&](auto K) MI_LAMBDA { last_act[K] = 0.f; last_qd[K] = 0.f; });
        sfor<4>([&](auto K) MI_LAMBDA { air[K] = 0.f; });
        progress = 0;
        if (valid) {
            st_cnt = 1.f;
            sfor<kAnymalSums>([&](auto K) MI_LAMBDA { st_sums[K] = sums[K]; });
        }
        sfor<kAnymalSums>([&](auto K) MI_LAMBDA { sums[K] = 0.f; });
        ep += 1;
        if (valid) {
            sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = root[K]; });
            sfor<ND>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = q[K]; v.dof[(ND + K) * N + e] = qd[K]; });
            // (reset_idx writes the task's dof tensors and pushes them to the sim, :399-409: the lagging tensor holds the reset values too)
            if (v.dof_api != nullptr) sfor<ND>([&](auto K) MI_LAMBDA { v.dof_api[K * N + e] = q[K]; v.dof_api[(ND + K) * N + e] = qd[K]; });
            sfor<NSPH3>([&](auto K) MI_LAMBDA { v.lamc[K * N + e] = 0.f; });
            v.terrain_levels[e] = level;
            sfor<3>([&](auto K) MI_LAMBDA { v.env_origins[K * N + e] = origin[K]; });
        }
    }
    // extras["episode"] partial sums (:421-425)
    red.extras(v, st_sums, st_cnt, valid ? (float)level : 0.f);
    red.episode(v, e, valid, rew, reset, progress);
    if (!valid) return;
    // ------------------------------------------------------------------ compute_observations (:302-313) + noise (:481-482)
    float* ob = v.obs + (size_t)e * kAnymalObs;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * kAnymalObs;
    const uint32_t sk = step_counter | 0x80000000u;
    auto emit = [&](int k, float val, float noise_scale) MI_LAMBDA {
        MI_NO_CONTRACT
        if (p.add_noise) val += (2.f * anymal_rand_step(v.seed, genv, sk, (uint32_t)(16 + k)) - 1.f) * noise_scale;
        ob[k] = val;
        oc[k] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs);
    };
    {
        MI_NO_CONTRACT
        sfor<3>([&](auto K) MI_LAMBDA { emit(K, base_lin_vel[K] * p.lin_vel_scale, p.noise_lin_vel); });
        sfor<3>([&](auto K) MI_LAMBDA { emit(3 + K, base_ang_vel[K] * p.ang_vel_scale, p.noise_ang_vel); });
        sfor<3>([&](auto K) MI_LAMBDA { emit(6 + K, proj_g[K], p.noise_gravity); });
        if constexpr (OBS_ALL) {
            emit(9, cmd[0] * p.lin_vel_scale, 0.f); emit(10, cmd[1] * p.lin_vel_scale, 0.f); emit(11, cmd[2] * p.ang_vel_scale, 0.f);
            sfor<ND>([&](auto K) MI_LAMBDA { emit(12 + K, q[K] * p.dof_pos_scale, p.noise_dof_pos); });
            sfor<ND>([&](auto K) MI_LAMBDA { emit(24 + K, qd[K] * p.dof_vel_scale, p.noise_dof_vel); });
            // columns 36..175 (the 140-point height scan, :515-538) are written by anymal_height_point right after this pass:
            // on the device one thread per (env, point) instead of 140 serial gathers per lane
            sfor<ND>([&](auto K) MI_LAMBDA { emit(176 + K, act[K], 0.f); });
        }
    }
    // bookkeeping (:484-485, vec_task.py:394)
    sfor<ND>([&](auto K) MI_LAMBDA { v.last_actions[K * N + e] = act[K]; v.last_dof_vel[K * N + e] = qd[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { v.feet_air_time[K * N + e] = air[K]; v.commands[K * N + e] = cmd[K]; });
    sfor<kAnymalSums>([&](auto K) MI_LAMBDA { v.episode_sums[K * N + e] = sums[K]; });
    v.randomize[e] += 1;
    v.episode[e] = ep;
    v.rew[e] = rew;
    v.reset[e] = reset;     // stays 1 for an env that was just reset (:418)
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)((progress >= (long long)p.max_episode_length - 1) && (reset != 0));
}

// extras["episode"] (:421-425) for slot k of the 16: means over the envs reset this step, divided by max_episode_length_s; terrain level
// mean.  `mine` = ep_stats[k], `cnt` = ep_stats[13], both read BEFORE any slot is re-zeroed (the caller orders that).
MI_HD void anymal_extras_slot(const View& v, const AnymalParams& p, const int k, const float mine, const float cnt) {
    // job-wide extras of a multi-GPU run: the same sums, cumulative ([15] counts the steps; [14] the per-step level sums)
    // ([16 + k]: the low part of slot k's compensated sum)
    if (k < 15) two_sum_acc(v.ep_cum[k], v.ep_cum[16 + k], mine);
    if (k == 15) two_sum_acc(v.ep_cum[15], v.ep_cum[31], 1.f);
    if (k < kAnymalSums && cnt > 0.f) v.ep_means[k] = mine / cnt / p.max_episode_length_s;   // untouched when nobody reset (the
    if (k == 14) v.ep_means[14] = mine / (float)v.N;                                         // reference keeps the last dict)
    if (k == 13) v.ep_means[13] = mine;
    if (k < 16) v.ep_stats[k] = 0.f;
}

// get_heights (:515-538) + the height columns of compute_observations (:311) + their noise (:481-482) for scan point k of env e.
// Runs after the post pass because the scan uses the POST-reset base pose; same functions, same draw indices => same values as a serial scan.
MI_HD void anymal_height_point(const View& v, const AnymalParams& p, const AnymalTerrainDesc& T, const unsigned step_counter, const int e, const int k) {
    MI_NO_CONTRACT
    const int N = v.N;
    const float root[7] = {v.root[e], v.root[N + e], v.root[2 * N + e], 0.f, 0.f, v.root[5 * N + e], v.root[6 * N + e]};
    // yaw-only quaternion of the base orientation (:676-681)
    float yq[4] = {0.f, 0.f, root[5], root[6]};
    const float yn = fmaxf(sqrtf(yq[2] * yq[2] + yq[3] * yq[3]), 1e-9f);
    yq[2] /= yn; yq[3] /= yn;
    float hm = 0.f;
    if (T.hs != nullptr) hm = anymal_height_at(T, yq, root, k);
    float val = fminf(fmaxf(root[2] - 0.5f - hm, -1.f), 1.f) * p.height_meas_scale;
    if (p.add_noise) {
        const uint32_t genv = (uint32_t)(v.env_offset + e), sk = step_counter | 0x80000000u;
        val += (2.f * anymal_rand_step(v.seed, genv, sk, (uint32_t)(16 + 36 + k)) - 1.f) * p.noise_height;
    }
    const size_t o = (size_t)e * kAnymalObs + 36 + k;
    v.obs[o] = val;
    v.obs_out[(size_t)v.ring * N * kAnymalObs + o] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs);
}

// The observation columns whose inputs sit in memory in their final form once anymal_post_env is through (commands 9 .. 11, dof positions 12 .. 23,
// dof velocities 24 .. 35, actions 176 .. 187: 39 columns, c = 0 .. 38), for one env: same expressions, same noise draws (index 16 + column) as the
// emit() calls of anymal_post_env<OBS_ALL = true>.
constexpr int kAnymalPlainCols = 3 + 3 * kAnymalDof;
MI_HD void anymal_obs_column(const View& v, const AnymalParams& p, const unsigned step_counter, const int e, const int c) {
    MI_NO_CONTRACT
    constexpr int ND = kAnymalDof;
    const int N = v.N;
    const float* const dofs = (v.dof_api != nullptr) ? v.dof_api : v.dof;      // (the task's dof-state tensor, see anymal_post_env)
    int k;
    float val, noise_scale;
    if (c < 3) { k = 9 + c; val = v.commands[c * N + e] * (c < 2 ? p.lin_vel_scale : p.ang_vel_scale); noise_scale = 0.f; }
    else if (c < 3 + ND) { k = 12 + (c - 3); val = dofs[(c - 3) * N + e] * p.dof_pos_scale; noise_scale = p.noise_dof_pos; }
    else if (c < 3 + 2 * ND) { k = 24 + (c - 3 - ND); val = dofs[(ND + c - 3 - ND) * N + e] * p.dof_vel_scale; noise_scale = p.noise_dof_vel; }
    else { k = 176 + (c - 3 - 2 * ND); val = v.actions[(c - 3 - 2 * ND) * N + e]; noise_scale = 0.f; }
    if (p.add_noise) val += (2.f * anymal_rand_step(v.seed, (uint32_t)(v.env_offset + e), step_counter | 0x80000000u, (uint32_t)(16 + k)) - 1.f) * noise_scale;
    const size_t o = (size_t)e * kAnymalObs + k;
    v.obs[o] = val;
    v.obs_out[(size_t)v.ring * N * kAnymalObs + o] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs);
}

// ------------------------------------------------------------------------------------------------ init / explicit reset
// the constructor's reset_idx(arange(num_envs)) (:170) with init_done False: terrain level 0..maxInitMapLevel, random type
template <int NB3>
MI_HD void anymal_init_env(const View& v, const AnymalParams& p, const AnymalTerrainDesc& T, const int max_init_level, const int e) {
    const int N = v.N;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    // terrain_levels = randint(0, maxInitMapLevel+1), terrain_types = randint(0, numTerrains) (:260-261)
    int level = 0, type = 0;
    float origin[3] = {0.f, 0.f, 0.f};
    if (T.hs != nullptr) {
        level = (int)(uniform01(v.seed ^ 0x1234567u, genv, 0u, 0u) * (float)(max_init_level + 1));
        level = level > max_init_level ? max_init_level : level;
        type = (int)(uniform01(v.seed ^ 0x1234567u, genv, 0u, 1u) * (float)T.types);
        type = type >= T.types ? T.types - 1 : type;
        for (int k = 0; k < 3; ++k) origin[k] = T.origins[(level * T.types + type) * 3 + k];
    }
    v.terrain_levels[e] = level;
    v.terrain_types[e] = type;
    for (int k = 0; k < 3; ++k) v.env_origins[k * N + e] = origin[k];
    // friction buckets (:236-239, 279-281): 100 buckets U(frictionRange), env i uses bucket i % 100
    const float fb = (p.friction_range[1] - p.friction_range[0]) * uniform01(v.seed ^ 0x7654321u, (uint32_t)(genv % 100u), 0u, 0u) +
                     p.friction_range[0];
    v.friction[e] = fb;
    for (int k = 0; k < 4; ++k) { v.commands[k * N + e] = 0.f; v.feet_air_time[k * N + e] = 0.f; }
    for (int k = 0; k < kAnymalDof; ++k) { v.last_actions[k * N + e] = 0.f; v.last_dof_vel[k * N + e] = 0.f; }
    for (int k = 0; k < kAnymalSums; ++k) v.episode_sums[k * N + e] = 0.f;
    for (int k = 0; k < NB3; ++k) v.netf[k * N + e] = 0.f;
    if (e == 0) for (int k = 0; k < 16; ++k) { v.ep_stats[k] = 0.f; v.ep_means[k] = 0.f; v.ep_cum[k] = 0.f; v.ep_cum[16 + k] = 0.f; }
}

// reset_idx(env_ids) (:384-425) outside step(): same draws as the in-step reset of the env's current episode number.  ep_stats[15] holds
// sum(cx^2 + cy^2) over the envs of THIS call (torch.norm(self.commands[env_ids, :2]), :431), put there by the caller.
template <int NSPH3>
MI_HD void anymal_reset_env(const View& v, const AnymalParams& p, const AnymalTerrainDesc& T, const int e) {
    MI_NO_CONTRACT
    constexpr int ND = kAnymalDof;
    const int N = v.N;
    const uint32_t genv = (uint32_t)(v.env_offset + e), uep = (uint32_t)v.episode[e];
    // update_terrain_level (:427-435), as in the step's own reset: not before the first episode, not without curriculum
    if (p.curriculum && T.hs != nullptr && uep > 0) {
        const float ddx = v.root[e] - v.env_origins[e], ddy = v.root[N + e] - v.env_origins[N + e];
        const float distance = sqrtf(ddx * ddx + ddy * ddy);
        const float cn = sqrtf(v.ep_stats[15]);
        int level = v.terrain_levels[e];
        level -= (distance < cn * p.max_episode_length_s * 0.25f) ? 1 : 0;
        level += (distance > T.env_length / 2.f) ? 1 : 0;
        level = (level < 0 ? 0 : level) % T.levels;
        v.terrain_levels[e] = level;
        for (int k = 0; k < 3; ++k) v.env_origins[k * N + e] = T.origins[(level * T.types + v.terrain_types[e]) * 3 + k];
    }
    for (int d = 0; d < ND; ++d) {
        v.dof[d * N + e] = p.default_dof_pos[d] * ((1.5f - 0.5f) * uniform01(v.seed, genv, uep, (uint32_t)d) + 0.5f);
        v.dof[(ND + d) * N + e] = (0.1f - (-0.1f)) * uniform01(v.seed, genv, uep, (uint32_t)(ND + d)) + (-0.1f);
        if (v.dof_api != nullptr) { v.dof_api[d * N + e] = v.dof[d * N + e]; v.dof_api[(ND + d) * N + e] = v.dof[(ND + d) * N + e]; }
        v.last_actions[d * N + e] = 0.f; v.last_dof_vel[d * N + e] = 0.f;
    }
    float root[13];
    for (int k = 0; k < 13; ++k) root[k] = p.base_init_state[k];
    if (T.hs != nullptr) {
        for (int k = 0; k < 3; ++k) root[k] += v.env_origins[k * N + e];
        root[0] += (0.5f - (-0.5f)) * uniform01(v.seed, genv, uep, 2 * ND + 0) + (-0.5f);
        root[1] += (0.5f - (-0.5f)) * uniform01(v.seed, genv, uep, 2 * ND + 1) + (-0.5f);
    }
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
    float cmd[4];
    cmd[0] = (p.command_x[1] - p.command_x[0]) * uniform01(v.seed, genv, uep, 2 * ND + 2) + p.command_x[0];
    cmd[1] = (p.command_y[1] - p.command_y[0]) * uniform01(v.seed, genv, uep, 2 * ND + 3) + p.command_y[0];
    cmd[2] = v.commands[2 * N + e];
    cmd[3] = (p.command_yaw[1] - p.command_yaw[0]) * uniform01(v.seed, genv, uep, 2 * ND + 4) + p.command_yaw[0];
    const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.25f) ? 1.f : 0.f;
    for (int k = 0; k < 4; ++k) { v.commands[k * N + e] = cmd[k] * keep; v.feet_air_time[k * N + e] = 0.f; }
    for (int k = 0; k < kAnymalSums; ++k) v.episode_sums[k * N + e] = 0.f;
    for (int k = 0; k < NSPH3; ++k) v.lamc[k * N + e] = 0.f;
    v.episode[e] += 1;
    v.progress[e] = 0;
    v.reset[e] = 1;   // :418
}

// =================================================================================================== Anymal (flat ground)
// post_physics_step of isaacgymenvs/tasks/anymal.py:231-241: progress++, reset_idx of flagged envs, observations, reward.
// State tensors are written immediately (CPU-pipeline semantics), contact forces / dof forces are
// those of the last sim step -- so an env that terminates on a base contact is flagged again right after its reset and
// resets twice, exactly as the reference's stale `contact_forces` make it do.
template <int NSPH3, class RED>
MI_HD void anymal_flat_post_env(const View& v, const AnymalFlatParams& p, const int e0, const RED& red) {
    constexpr int ND = kAnymalDof;
    const int N = v.N;
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    float root[13], q[ND], qd[ND], act[ND], torques[ND], cmd[3];
    sfor<13>([&](auto K) MI_LAMBDA { root[K] = v.root[K * N + e]; });
    sfor<ND>([&](auto K) MI_LAMBDA {
        q[K] = v.dof[K * N + e]; qd[K] = v.dof[(ND + K) * N + e];
        act[K] = v.actions[K * N + e]; torques[K] = v.dof_force[K * N + e];
    });
    sfor<3>([&](auto K) MI_LAMBDA { cmd[K] = v.commands[K * N + e]; });
    long long progress = v.progress[e] + 1;
    int ep = v.episode[e];
    if (v.reset[e] != 0) {
        anymal_flat_reset(p, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, root, q, qd, cmd);
        ep += 1;
        progress = 0;
        if (valid) {
            sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = root[K]; });
            sfor<ND>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = q[K]; v.dof[(ND + K) * N + e] = qd[K]; v.laml[K * N + e] = 0.f; });
            sfor<3>([&](auto K) MI_LAMBDA { v.commands[K * N + e] = cmd[K]; });
            sfor<NSPH3>([&](auto K) MI_LAMBDA { v.lamc[K * N + e] = 0.f; });
        }
    }
    float obs[kAnymalFlatObs];
    anymal_flat_observations(p, root, cmd, q, qd, act, obs);
    float base_c[3], knee_c[4][3];
    sfor<3>([&](auto K) MI_LAMBDA { base_c[K] = v.netf[K * N + e]; });
    sfor<4>([&](auto J) MI_LAMBDA { sfor<3>([&](auto K) MI_LAMBDA { knee_c[J][K] = v.netf[(3 * anymal_knee_body(J) + K) * N + e]; }); });
    float rew;
    long long reset;
    anymal_flat_reward(p, root, cmd, torques, base_c, knee_c, progress, &rew, &reset);
    red.episode(v, e, valid, rew, reset, progress);
    if (!valid) return;
    v.randomize[e] += 1;
    v.episode[e] = ep;
    float* ob = v.obs + (size_t)e * kAnymalFlatObs;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * kAnymalFlatObs;
    sfor<kAnymalFlatObs>([&](auto K) MI_LAMBDA { ob[K] = obs[K]; oc[K] = fminf(fmaxf(obs[K], -v.clip_obs), v.clip_obs); });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)((progress >= (long long)p.max_episode_length - 1) && (reset != 0));   // vec_task.py:394
}

// constructor state (anymal.py:141-146): initial_root_states := base_init_state, then reset_idx(arange(num_envs))
template <int NB3>
MI_HD void anymal_flat_init_env(const View& v, const AnymalFlatParams& p, const int e) {
    const int N = v.N;
    for (int k = 0; k < 13; ++k) v.init_root[k * N + e] = p.base_init_state[k];
    for (int k = 0; k < NB3; ++k) v.netf[k * N + e] = 0.f;
    for (int k = 0; k < 3; ++k) v.commands[k * N + e] = 0.f;
}
template <int NSPH3>
MI_HD void anymal_flat_reset_env(const View& v, const AnymalFlatParams& p, const int e) {
    constexpr int ND = kAnymalDof;
    const int N = v.N;
    float root[13], q[ND], qd[ND], cmd[3];
    anymal_flat_reset(p, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)v.episode[e], root, q, qd, cmd);
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
    for (int d = 0; d < ND; ++d) { v.dof[d * N + e] = q[d]; v.dof[(ND + d) * N + e] = qd[d]; v.laml[d * N + e] = 0.f; }
    for (int k = 0; k < 3; ++k) v.commands[k * N + e] = cmd[k];
    for (int k = 0; k < NSPH3; ++k) v.lamc[k * N + e] = 0.f;
    v.episode[e] += 1;
    v.progress[e] = 0;
    v.reset[e] = 1;   // anymal.py:301
}

}  // namespace mi

// hand_task_kernels.hpp -- the in-hand manipulation tasks on core/hand_engine.hpp, a template over the task (hand_kernels.hpp: ShadowHandTask =
// reference isaacgymenvs/tasks/shadow_hand.py, AllegroHandTask = isaacgymenvs/tasks/allegro_hand.py; the two files share their control flow
// line by line -- the Allegro task has 16 dofs, all of them driven, no tendons, no fingertip states and no force sensors in its observations):
// pre_physics_step (deferred resets, actions -> position targets), hand + object physics sub-steps, post_physics_step (fingertip states,
// full_state observation, compute_hand_reward).  One env per lane; 32 envs per wave in the physics kernel (the compact contact store needs
// ~2.5 KB of LDS per env).  Instantiated in kernels_shadow_hand.hip and kernels_allegro_hand.hip.
#pragma once
#include "hand_kernels.hpp"

namespace mi {


__device__ __forceinline__ float hand_u(uint32_t seed, uint32_t genv, uint32_t ep, uint32_t k) { return 2.f * uniform01(seed, genv, ep, k) - 1.f; }
// random_force_prob (:198-199, 642-643): log-uniform in force_prob_range
__device__ __forceinline__ float hand_force_prob(const HandParams& p, float u) {
    MI_NO_CONTRACT
    return expf((logf(p.force_prob_range[0]) - logf(p.force_prob_range[1])) * u + logf(p.force_prob_range[1]));
}

// reset_target_pose (shadow_hand.py:586-602): new random goal orientation
__device__ __forceinline__ void hand_reset_goal(const View& v, const HandView& hv, const HandParams& p, int e, uint32_t genv) {
    const int N = v.N;
    const uint32_t gc = (uint32_t)hv.goal_count[e];
    const float r0 = hand_u(v.seed ^ 0x2545F491u, genv, gc, 0), r1 = hand_u(v.seed ^ 0x2545F491u, genv, gc, 1);
    const float xu[3] = {1.f, 0.f, 0.f}, yu[3] = {0.f, 1.f, 0.f};
    float q[4];
    randomize_rotation(r0, r1, xu, yu, q);
    sfor<3>([&](auto K) MI_LAMBDA { hv.goal_state[K * N + e] = p.goal_init_pos[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { hv.goal_state[(3 + K) * N + e] = q[K]; });
    hv.goal_count[e] = (int)gc + 1;
    hv.reset_goal[e] = 0;
}

// reset_idx (shadow_hand.py:604-668, allegro_hand.py:526-590) for one env
template <class HT>
__device__ __forceinline__ void hand_reset_env(const View& v, const HandView& hv, const HandParams& p, int e, uint32_t genv) {
    MI_NO_CONTRACT
    const int N = v.N, ND = HT::ND;
    const uint32_t ep = (uint32_t)v.episode[e];
    auto rf = [&](int k) MI_LAMBDA { return hand_u(v.seed, genv, ep, (uint32_t)k); };   // rand_floats[:, k], U(-1, 1)
    hand_reset_goal(v, hv, p, e, genv);
    // object: initial pose + position noise, random rotation, zero velocity
    hv.object_state[0 * N + e] = p.object_init_pos[0] + p.reset_position_noise * rf(0);
    hv.object_state[1 * N + e] = p.object_init_pos[1] + p.reset_position_noise * rf(1);
    hv.object_state[2 * N + e] = p.object_init_pos[2] + p.reset_position_noise * rf(2);
    const float xu[3] = {1.f, 0.f, 0.f}, yu[3] = {0.f, 1.f, 0.f};
    float q[4];
    randomize_rotation(rf(3), rf(4), xu, yu, q);
    if (p.object_shape == OBJ_CAPSULE) {                 // pen: randomize_rotation_pen with rand_angle_y = 0.3 (shadow_hand.py:626-629)
        const float zu[3] = {0.f, 0.f, 1.f};
        randomize_rotation_pen(rf(3), rf(4), 0.3f, xu, yu, zu, q);
    }
    sfor<4>([&](auto K) MI_LAMBDA { hv.object_state[(3 + K) * N + e] = q[K]; });
    sfor<6>([&](auto K) MI_LAMBDA { hv.object_state[(7 + K) * N + e] = 0.f; });
    // hand: default pose (0) + noise * random point of the joint range (:642-651)
    sfor<ND>([&](auto D) MI_LAMBDA {
        constexpr int d = D;
        const float delta_max = HT::M::dof_upper[d] - 0.f, delta_min = HT::M::dof_lower[d] - 0.f;
        const float rand_delta = delta_min + (delta_max - delta_min) * 0.5f * (rf(5 + d) + 1.f);
        const float pos = 0.f + p.reset_dof_pos_noise * rand_delta;
        v.dof[d * N + e] = pos;
        v.dof[(ND + d) * N + e] = 0.f + p.reset_dof_vel_noise * rf(5 + ND + d);
        hv.prev_targets[d * N + e] = pos;
        hv.cur_targets[d * N + e] = pos;
        v.laml[d * N + e] = 0.f;
    });
    sfor<3>([&](auto K) MI_LAMBDA { hv.rb_force[K * N + e] = 0.f; hv.obj_force[K * N + e] = 0.f; });       // :616
    hv.force_prob[e] = hand_force_prob(p, uniform01(v.seed, genv, ep, 5 + 2 * ND));                           // :642-643
    v.episode[e] = (int)ep + 1;
    v.progress[e] = 0;
    v.reset[e] = 0;
    hv.successes[e] = 0.f;
}

// pre_physics_step (shadow_hand.py:670-698): deferred resets, then actions -> targets
template <class HT>
__global__ __launch_bounds__(64) void hand_pre_kernel(View v, HandView hv, HandParams p, const float* __restrict__ actions_in, unsigned step_counter) {
    MI_NO_CONTRACT
    const int N = v.N;
    const int e = post_env_index<HandSim<typename HT::M>::LANES>(blockIdx.x, threadIdx.x, N);   // same env -> XCD mapping as the sub-step kernel
    if (e >= N) return;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    if (v.reset[e] != 0) hand_reset_env<HT>(v, hv, p, e, genv);           // also resets the goal (:615)
    else if (hv.reset_goal[e] != 0) hand_reset_goal(v, hv, p, e, genv);
    float raw_act[HT::NACT];
    sfor<HT::NACT>([&](auto A_) MI_LAMBDA { raw_act[A_] = actions_in[(size_t)e * HT::NACT + A_]; });
    if (v.act_noise.dist != 0)                                                                                     // vec_task.py:371-372 (one block: a real branch)
        sfor<HT::NACT>([&](auto A_) MI_LAMBDA { raw_act[A_] = apply_noise(v.act_noise, v.seed, genv, v.step, 1u, (uint32_t)A_, raw_act[A_]); });
    sfor<HT::NACT>([&](auto A_) MI_LAMBDA {
        constexpr int a = A_;
        const int d = p.actuated[a];
        const float act = fminf(fmaxf(raw_act[a], -p.clip_actions), p.clip_actions);                               // vec_task.py:374
        v.actions[a * N + e] = act;
        // the actuated dof index is a runtime table: read the limits through a tiny switch-free lookup
        float lo = 0.f, up = 0.f;
        sfor<HT::ND>([&](auto D) MI_LAMBDA { if (d == D) { lo = HT::M::dof_lower[D]; up = HT::M::dof_upper[D]; } });
        const float prev = hv.prev_targets[d * N + e];
        float t;
        if (p.use_relative_control) {
            t = prev + p.dof_speed_scale * p.dt * act;                                     // :686
        } else {
            t = 0.5f * (act + 1.0f) * (up - lo) + lo;                                      // scale(), torch_jit_utils.py:234-235
            t = p.act_moving_average * t + (1.0f - p.act_moving_average) * prev;            // :692-693
        }
        t = fmaxf(fminf(t, up), lo);                                                       // tensor_clamp
        hv.cur_targets[d * N + e] = t;
        hv.prev_targets[d * N + e] = t;                                                     // :697
    });
    if (p.force_scale > 0.f) {   // random forces on the object (:700-708)
        const float decay = powf(p.force_decay, p.dt / p.force_decay_interval);
        float f[3];
        sfor<3>([&](auto K) MI_LAMBDA { f[K] = hv.rb_force[K * N + e] * decay; });
        const uint32_t sk = step_counter | 0x80000000u, sd = v.seed ^ 0x9E3779B9u;
        if (uniform01(sd, genv, sk, 0) < hv.force_prob[e]) {
            // torch.randn(3) * object mass * force_scale; Box-Muller on the engine's counter-based uniforms
            const float u1 = fmaxf(uniform01(sd, genv, sk, 1), 1e-7f), u2 = uniform01(sd, genv, sk, 2);
            const float u3 = fmaxf(uniform01(sd, genv, sk, 3), 1e-7f), u4 = uniform01(sd, genv, sk, 4);
            const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u3));
            const float k = p.cube_mass * p.force_scale;
            f[0] = r1 * cosf(6.283185307179586f * u2) * k;
            f[1] = r1 * sinf(6.283185307179586f * u2) * k;
            f[2] = r2 * cosf(6.283185307179586f * u4) * k;
        }
        float q[4], fw[3];
        sfor<4>([&](auto K) MI_LAMBDA { q[K] = hv.object_state[(3 + K) * N + e]; });
        quat_rotate_s(q, f, 1.f, fw);                                                      // LOCAL_SPACE -> world at application time
        sfor<3>([&](auto K) MI_LAMBDA { hv.rb_force[K * N + e] = f[K]; hv.obj_force[K * N + e] = fw[K]; });
    }
}

// gym.refresh_rigid_body_state_tensor (shadow_hand.py:440,456-457) for the five fingertip bodies: ONE THREAD PER (env, fingertip) -- blockIdx.y is
// the fingertip, so a wave walks one chain (wrist + one finger, 6 or 7 hinges) and 5 N / 64 waves fill the chip, where the post kernel's
// one lane per env walked all five chains in turn on 256 waves (latency-bound: 40 of its 61 us at 16384 envs).
template <class HT, int K>
__device__ __forceinline__ void hand_tip(const View& v, const HandView& hv, const HandParams& p, const int e) {
    constexpr int ND = HT::ND, tip = HT::M::sens_body[K];
    const int N = v.N;
    HandSim<typename HT::M> sim;
    sfor<3>([&](auto I_) MI_LAMBDA { sim.root[I_] = p.hand_pos[I_]; });
    sfor<4>([&](auto I_) MI_LAMBDA { sim.root[3 + I_] = p.hand_quat[I_]; });
    sfor<ND>([&](auto D) MI_LAMBDA {
        if constexpr (HandSim<typename HT::M>::is_ancestor_or_self(HT::M::dof_body[D], tip)) { sim.q[D] = v.dof[D * N + e]; sim.qd[D] = v.dof[(ND + D) * N + e]; }
    });
    float o[13];
    sim.template fingertip_state<K>(o);
    sfor<13>([&](auto I_) MI_LAMBDA { hv.fingertip[(K * 13 + I_) * N + e] = o[I_]; });
}
template <class HT>
__global__ __launch_bounds__(64) void hand_tips_kernel(View v, HandView hv, HandParams p) {
    MI_NO_CONTRACT
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= v.N) return;
    static_assert(HT::NTIPS == 5 || HT::NTIPS == 0, "five fingertips (ShadowHand) or none (AllegroHand: no fingertip observations)");
    if constexpr (HT::NTIPS == 5) {
        switch (blockIdx.y) {
            case 0: hand_tip<HT, 0>(v, hv, p, e); break;
            case 1: hand_tip<HT, 1>(v, hv, p, e); break;
            case 2: hand_tip<HT, 2>(v, hv, p, e); break;
            case 3: hand_tip<HT, 3>(v, hv, p, e); break;
            default: hand_tip<HT, 4>(v, hv, p, e); break;
        }
    }
}

// post_physics_step (shadow_hand.py:710-715): progress++, compute_observations (full_state), compute_reward; the fingertip states come
// from hand_tips_kernel
template <class HT>
__global__ __launch_bounds__(64) void hand_post_kernel(View v, HandView hv, HandParams p) {
    MI_NO_CONTRACT
    constexpr int ND = HT::ND;
    const int N = v.N;
    const int e0 = post_env_index<HandSim<typename HT::M>::LANES>(blockIdx.x, threadIdx.x, N);
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    float q[ND], qd[ND];
    sfor<ND>([&](auto K) MI_LAMBDA { q[K] = v.dof[K * N + e]; qd[K] = v.dof[(ND + K) * N + e]; });
    float tips[HT::NTIPS > 0 ? HT::NTIPS : 1][13];
    sfor<HT::NTIPS>([&](auto T_) MI_LAMBDA { sfor<13>([&](auto K) MI_LAMBDA { tips[T_][K] = hv.fingertip[(T_ * 13 + K) * N + e]; }); });
    float os[13], gp[7], act[HT::NACT], dff[ND], sns[HT::NTIPS > 0 ? 6 * HT::NTIPS : 1];
    // every input is loaded before the first observation is stored: the stores below may alias these arrays as far as the
    // compiler knows, and a load that has to wait for them is a fully exposed memory round trip for a lone wave
    sfor<ND>([&](auto K) MI_LAMBDA { dff[K] = v.dof_force[K * N + e]; });
    sfor<6 * HT::NTIPS>([&](auto K) MI_LAMBDA { sns[K] = v.sensor[K * N + e]; });
    const long long reset_in = v.reset[e], reset_goal_in = hv.reset_goal[e];
    const float successes_in = hv.successes[e];
    sfor<13>([&](auto K) MI_LAMBDA { os[K] = hv.object_state[K * N + e]; });
    sfor<7>([&](auto K) MI_LAMBDA { gp[K] = hv.goal_state[K * N + e]; });
    sfor<HT::NACT>([&](auto K) MI_LAMBDA { act[K] = v.actions[K * N + e]; });
    const long long progress_in = v.progress[e] + 1;               // :711
    // compute_full_state (:528-584)
    // obs_type 0: the vector IS obs_buf; otherwise it goes to full_state and hand_obs_select_kernel picks obs_buf's columns.
    // With asymmetric observations full_state (= states_buf) is written in both cases.
    const bool direct = p.obs_type == 0, to_full = !direct || p.asymmetric_obs != 0;
    // The 211 columns of an env are a row of the row-major obs tensors: written lane by lane, every store instruction of the wave
    // touches 64 rows (64 cache lines for 4 bytes each).  They are staged in LDS instead, column-major with a pad ([k][65]: the
    // lanes of a column and the columns of a lane both fall on distinct banks), and the wave writes the rows out together below,
    // 64 consecutive floats per store instruction.
    __shared__ float stage[HT::NFULL * 65];
    auto emit = [&](int k, float val) MI_LAMBDA { stage[k * 65 + (int)threadIdx.x] = val; };
    sfor<ND>([&](auto D) MI_LAMBDA {
        constexpr int d = D;
        emit(d, (2.0f * q[d] - HT::M::dof_upper[d] - HT::M::dof_lower[d]) / (HT::M::dof_upper[d] - HT::M::dof_lower[d]));   // unscale
        emit(ND + d, p.vel_obs_scale * qd[d]);
        emit(2 * ND + d, p.force_torque_obs_scale * dff[d]);
    });
    // layout (shadow_hand.py:528-584 with 24 dofs and 5 fingertips: 211 columns; allegro_hand.py:485-507 with 16 dofs and none: 88):
    // 3 ND | object pose 7, linvel 3, angvel 3 | goal pose 7, quat diff 4 | 13 NTIPS fingertip states | 6 NTIPS force-torques | actions
    constexpr int O_OBJ = 3 * ND, O_GOAL = O_OBJ + 13, O_TIPS = O_GOAL + 11, O_FT = O_TIPS + 13 * HT::NTIPS, O_ACT = O_FT + 6 * HT::NTIPS;
    static_assert(O_ACT + HT::NACT == HT::NFULL, "full_state width");
    sfor<7>([&](auto K) MI_LAMBDA { emit(O_OBJ + K, os[K]); });
    sfor<3>([&](auto K) MI_LAMBDA { emit(O_OBJ + 7 + K, os[7 + K]); emit(O_OBJ + 10 + K, p.vel_obs_scale * os[10 + K]); });
    sfor<7>([&](auto K) MI_LAMBDA { emit(O_GOAL + K, gp[K]); });
    {
        float conj[4], qd4[4];
        quat_conjugate(gp + 3, conj);
        quat_mul(os + 3, conj, qd4);
        sfor<4>([&](auto K) MI_LAMBDA { emit(O_GOAL + 7 + K, qd4[K]); });
    }
    sfor<HT::NTIPS>([&](auto T_) MI_LAMBDA {
        sfor<13>([&](auto K) MI_LAMBDA {
            emit(O_TIPS + T_ * 13 + K, tips[T_][K]);
        });
    });
    sfor<6 * HT::NTIPS>([&](auto K) MI_LAMBDA { emit(O_FT + K, p.force_torque_obs_scale * sns[K]); });
    sfor<HT::NACT>([&](auto K) MI_LAMBDA { emit(O_ACT + K, act[K]); });
    // compute_hand_reward (:746-800)
    float r, succ;
    long long rs, gr, prog;
    hand_reward(p.rew, os, os + 3, gp, gp + 3, act, HT::NACT, reset_in, reset_goal_in, progress_in, successes_in, &r, &rs, &gr, &prog, &succ);
    float nres = valid ? (float)rs : 0.f, fin = valid ? succ * (float)rs : 0.f;
    nres = wave_sum(nres); fin = wave_sum(fin);
    if ((threadIdx.x & 63) == 0 && nres > 0.f) { atomicAdd(hv.ws, nres); atomicAdd(hv.ws + 1, fin); }
    episode_stats(v, e, valid, r, rs, prog);
    __syncthreads();
    for (int row = 0; row < 64; ++row) {                      // wave-uniform: env of lane `row`
        const int er = __shfl(e0, row);
        if (er >= N) continue;
        float* ob = v.obs + (size_t)er * HT::NFULL;
        float* oc = v.obs_out + ((size_t)v.ring * N + er) * HT::NFULL;
        float* fs = hv.full_state + (size_t)er * HT::NFULL;
        if (direct && v.obs_noise.dist != 0) {
            // observation noise of the domain randomisation: on obs_buf only, after the reward was computed from the clean state
            // (vec_task.py:397-399); states_buf stays clean.  (Its own loop: as a select inside the common loop the compiler evaluated
            // the noise hash for every element whether it was wanted or not, +7 us on the kernel.)
            for (int k = (int)threadIdx.x; k < HT::NFULL; k += 64) {
                const float val = stage[k * 65 + row];
                const float nv = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + er), v.step, 0u, (uint32_t)k, val);
                ob[k] = nv;
                oc[k] = fminf(fmaxf(nv, -v.clip_obs), v.clip_obs);
                if (to_full) fs[k] = val;
            }
            continue;
        }
        for (int k = (int)threadIdx.x; k < HT::NFULL; k += 64) {
            const float val = stage[k * 65 + row];
            if (direct) { ob[k] = val; oc[k] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs); }
            if (to_full) fs[k] = val;
        }
    }
    if (!valid) return;
    v.rew[e] = r;
    v.reset[e] = rs;
    hv.reset_goal[e] = gr;
    v.progress[e] = prog;
    hv.successes[e] = succ;
    v.randomize[e] += 1;
    v.timeout[e] = (unsigned char)(((float)prog >= p.rew.max_episode_length - 1.f) && (rs != 0));      // vec_task.py:394
}
// observationType openai / full_no_vel / full (shadow_hand.py:472-526): column subsets of the full state
template <class HT>
__global__ void hand_obs_select_kernel(View v, HandView hv, HandParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int no = p.num_obs;
    if (i >= v.N * no) return;
    const int e = i / no, k = i - e * no;
    float val = hv.full_state[(size_t)e * HT::NFULL + p.obs_map[k]];
    if (v.obs_noise.dist != 0) {   // (a block of its own so that the hash is not evaluated speculatively)
        const float nv = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + e), v.step, 0u, (uint32_t)k, val);
        v.obs[(size_t)e * no + k] = nv;
        v.obs_out[((size_t)v.ring * v.N + e) * no + k] = fminf(fmaxf(nv, -v.clip_obs), v.clip_obs);
        return;
    }
    v.obs[(size_t)e * no + k] = val;
    v.obs_out[((size_t)v.ring * v.N + e) * no + k] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs);
}
template <class HT>
__global__ void hand_finalize_kernel(HandView hv, HandParams p) {
    // (folding this into the post kernel's last block -- a fence and a counter at the end of every block -- was measured at the end of round 3:
    //  the post kernel grew by what this launch costs, 41.3 -> 45.0 us; a one-thread kernel hides in its neighbours' launch shadow)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float num_resets = hv.ws[0], finished = hv.ws[1], cs = hv.cons[0];
        hv.cons[0] = (num_resets > 0.f) ? p.rew.av_factor * finished / num_resets + (1.0f - p.rew.av_factor) * cs : cs;
        hv.ws[0] = 0.f; hv.ws[1] = 0.f;   // last reader of the step's sums: re-zero them here instead of a memset before every post kernel
    }
}

// initial state: buffers as the reference's __init__ leaves them (reset_buf = 1 => everything is reset at the first
// pre_physics_step), hand at its default pose, cube and goal at their initial poses
template <class HT>
__global__ void hand_init_kernel(View v, HandView hv, HandParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    for (int d = 0; d < HT::ND; ++d) {
        v.dof[d * N + e] = 0.f; v.dof[(HT::ND + d) * N + e] = 0.f; v.laml[d * N + e] = 0.f; v.dof_force[d * N + e] = 0.f;
        hv.cur_targets[d * N + e] = 0.f; hv.prev_targets[d * N + e] = 0.f;
    }
    for (int k = 0; k < 13; ++k) hv.object_state[k * N + e] = (k < 3) ? p.object_init_pos[k] : (k == 6 ? 1.f : 0.f);
    for (int k = 0; k < 7; ++k) hv.goal_state[k * N + e] = (k < 3) ? p.goal_init_pos[k] : (k == 6 ? 1.f : 0.f);
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = (k < 3) ? p.hand_pos[k] : (k < 7 ? p.hand_quat[k - 3] : 0.f);
    for (int k = 0; k < 6 * HT::NTIPS; ++k) v.sensor[k * N + e] = 0.f;
    for (int k = 0; k < 13 * HT::NTIPS; ++k) hv.fingertip[k * N + e] = 0.f;
    for (int k = 0; k < HT::NACT; ++k) v.actions[k * N + e] = 0.f;
    const int no = p.num_obs;
    for (int k = 0; k < no; ++k) { v.obs[(size_t)e * no + k] = 0.f; v.obs_out[(size_t)e * no + k] = 0.f; v.obs_out[((size_t)N + e) * no + k] = 0.f; }
    for (int k = 0; k < HT::NFULL; ++k) hv.full_state[(size_t)e * HT::NFULL + k] = 0.f;
    for (int k = 0; k < 3; ++k) { hv.obj_force[k * N + e] = 0.f; hv.rb_force[k * N + e] = 0.f; }
    hv.force_prob[e] = hand_force_prob(p, uniform01(v.seed ^ 0x51ED27u, (uint32_t)(v.env_offset + e), 0u, 0u));
    hv.mu_env[e] = -1.f;
    for (int k = 0; k < HS_COLUMNS; ++k) hv.scale[k * N + e] = 1.f;
    for (int k = 0; k < 2 * HT::ND; ++k) hv.limit_shift[k * N + e] = 0.f;
    hv.successes[e] = 0.f; hv.reset_goal[e] = 1; hv.goal_count[e] = 0; hv.ncontact[e] = 0; hv.ndropped[e] = 0;
    v.rew[e] = 0.f; v.reset[e] = 1; v.progress[e] = 0; v.randomize[e] = 0; v.timeout[e] = 0; v.episode[e] = 0; v.ep_ret[e] = 0.f;
    if (e == 0) { hv.cons[0] = 0.f; hv.ws[0] = hv.ws[1] = hv.ws[2] = hv.ws[3] = 0.f; for (int k = 0; k < 8; ++k) v.stats[k] = 0.f; }
}

// the task's physics sub-steps: which kernel form / object shape runs is the task's business (kernels_<task>.hip)
template <class HT>
hipError_t hand_substeps(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);

template <class HT>
hipError_t launch_step_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const float* actions, int cfi,
                            unsigned step_counter, hipStream_t s) {
    hipLaunchKernelGGL(hand_pre_kernel<HT>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, hv, p, actions, step_counter);
    hipError_t e = hand_substeps<HT>(v, hv, P, p, cfi * P.substeps, s);
    if (e != hipSuccess) return e;
    if constexpr (HT::NTIPS > 0) hipLaunchKernelGGL(hand_tips_kernel<HT>, dim3((v.N + 63) / 64, HT::NTIPS), dim3(64), 0, s, v, hv, p);
    hipLaunchKernelGGL(hand_post_kernel<HT>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, hv, p);
    if (p.obs_type != 0) hipLaunchKernelGGL(hand_obs_select_kernel<HT>, dim3((v.N * p.num_obs + 255) / 256), dim3(256), 0, s, v, hv, p);
    hipLaunchKernelGGL(hand_finalize_kernel<HT>, dim3(1), dim3(64), 0, s, hv, p);
    return hipGetLastError();
}
template <class HT>
hipError_t launch_simulate_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, hipStream_t s) {
    if (hipError_t e = hand_substeps<HT>(v, hv, P, p, P.substeps, s); e != hipSuccess) return e;
    if constexpr (HT::NTIPS > 0) hipLaunchKernelGGL(hand_tips_kernel<HT>, dim3((v.N + 63) / 64, HT::NTIPS), dim3(64), 0, s, v, hv, p);   // refresh_rigid_body_state_tensor
    return hipGetLastError();
}
template <class HT>
hipError_t launch_init_hand(const View& v, const HandView& hv, const HandParams& p, hipStream_t s) {
    hipLaunchKernelGGL(hand_init_kernel<HT>, dim3((v.N + 127) / 128), dim3(128), 0, s, v, hv, p);
    return hipGetLastError();
}
template <class HT>
__global__ void hand_reset_ids_kernel(View v, HandView hv, HandParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i];
    if (e < 0 || e >= v.N) return;
    hand_reset_env<HT>(v, hv, p, e, (uint32_t)(v.env_offset + e));
}
template <class HT>
hipError_t launch_reset_hand(const View& v, const HandView& hv, const HandParams& p, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(hand_reset_ids_kernel<HT>, dim3((n + 127) / 128), dim3(128), 0, s, v, hv, p, ids, n);
    return hipGetLastError();
}

}  // namespace mi

// mw_kernels.hpp -- the multi-wave physics sub-step kernel (core/engine_mw.hpp) and its launcher.  Included only by the
// kernels_mw_<model>.hip translation units, which instantiate launch_substeps_mw for one model / ground pair each.
#pragma once
#include "step_kernels.hpp"
#include "core/engine_mw.hpp"
#include "tasks/anymal_step.hpp"      // anymal_netf_norm, anymal_knee_body: the AnymalTerrain launch's curriculum tail (mw_role_fused)

namespace mi {

// ------------------------------------------------------------------------------------------------ multi-wave sub-step
#if defined(MI_TIMING)
__device__ unsigned long long* g_mi_tstamp_mw = nullptr;      // tools/debug/mw_phases.py (debug builds only)
#endif
// One env's sub-step spread over the 4 waves of a workgroup (core/engine_mw.hpp): blockDim = (64, NROLE), wave y = role y, lanes
// 0 .. E-1 of every wave hold the same E envs (the other lanes retire at once); env -> workgroup through xcd_env_base (step_kernels.hpp).
struct DevBarrier {
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
template <class M, int E>
constexpr size_t mw_lds_bytes() { return (size_t)SimMW<M>::MW_SLOTS * E * sizeof(float); }
// post_x: non-null = the step's last sub-step with post_physics_step fused in: the role also leaves its part of the new state (own q / qd; the
// trunk role: the root state) in the LDS exchange area for the wave that runs loco_post_env
template <class M, class GND, int E, int R>
__device__ __forceinline__ void mw_role(const View& v, const SimParams& P, const ActParams& ap, const float* __restrict__ actions_in,
                                        const int src, const GND& gnd, float* lds_rows, const int e, const int lane, float* post_x = nullptr) {
    using S = SimMW<M>;
    constexpr int ND = M::ND;
    const int N = v.N;
    S sim;
    load_sim(sim, v, e);
    load_actor_scales(sim, v, e);
    float tau[M::NDA];
    if (src != ACT_STORED_TAU) {
        sfor<ND>([&](auto K) MI_LAMBDA {
            constexpr int k = K;
            constexpr bool mine = S::template owns_gi<R>(M::OFF + k);
            float t = 0.f;
            if (k < ap.nact) {
                float a;
                if (src == ACT_FROM_ACTIONS) {
                    a = actions_in[(size_t)e * ap.nact + k];
                    a = fminf(fmaxf(a, -ap.clip), ap.clip);
                    if constexpr (mine) v.actions[k * N + e] = a;
                } else {
                    a = v.actions[k * N + e];
                }
                if (ap.mode == 0) {
                    t = a * ap.gear[k] * ap.scale;
                } else {
                    const float u = ap.kp * (ap.scale * a + ap.gear[k] - sim.q[k]) - ap.kd * sim.qd[k];
                    t = fminf(fmaxf(u, -ap.torque_limit), ap.torque_limit);
                }
            }
            tau[k] = t;
            if constexpr (mine) v.tau[k * N + e] = t;
        });
    } else {
        sfor<ND>([&](auto K) MI_LAMBDA { tau[K] = v.tau[K * N + e]; });
    }
    {   // last sub-step's impulses of the own rows: HBM -> row-store slots, LDS-direct
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        float* slot0 = lds_rows;     // lane l lands at slot base + 4 l
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            if constexpr (M::dof_limited[d] && S::template owns_gi<R>(M::OFF + d)) {
                constexpr int o = Sim<M>::stage_slot_lim(d) * E;
                __builtin_amdgcn_global_load_lds((gptr_t)(v.laml + (size_t)d * N + e), (lptr_t)(slot0 + o), 4, 0, 0);
            }
        });
        sfor<3 * M::NSPH>([&](auto K) MI_LAMBDA {
            if constexpr (S::template owns_body<R>(M::sph_body[K / 3])) {
                constexpr int o = Sim<M>::stage_slot_con(K) * E;
                __builtin_amdgcn_global_load_lds((gptr_t)(v.lamc + (size_t)K * N + e), (lptr_t)(slot0 + o), 4, 0, 0);
            }
        });
    }
    const float h = P.dt / (float)P.substeps;
    const Strided lamc{v.lamc + e, N}, laml{v.laml + e, N}, sensor{v.sensor + e, N}, dof_force{v.dof_force + e, N};
    const Strided netf{GND::NETF ? v.netf + e : nullptr, N};
    const float mu_env = (GND::HEIGHTFIELD || v.friction != nullptr) ? v.friction[e] : -1.f;
    sim.template substep_role<R>(P, tau, h, RowStore<E>{lds_rows + lane}, lamc, laml, sensor, dof_force, gnd, mu_env, netf, true, DevBarrier{});
    sfor<ND>([&](auto K) MI_LAMBDA {
        if constexpr (S::template owns_gi<R>(M::OFF + K)) {
            v.dof[K * N + e] = sim.q[K];
            v.dof[(ND + K) * N + e] = sim.qd[K];
        }
    });
    if constexpr (R == M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = sim.root[K]; });
    if (post_x != nullptr) {     // [13 + 2 ND][E]
        sfor<ND>([&](auto K) MI_LAMBDA {
            if constexpr (S::template owns_gi<R>(M::OFF + K)) { post_x[(13 + K) * E] = sim.q[K]; post_x[(13 + ND + K) * E] = sim.qd[K]; }
        });
        if constexpr (R == M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { post_x[K * E] = sim.root[K]; });
    }
}
template <class M, class GND, int E>
__global__ __launch_bounds__(64 * M::NROLE) void substep_mw_kernel(View v, SimParams P, ActParams ap, const float* __restrict__ actions_in,
                                                                   int src, GND gnd) {
    extern __shared__ float lds_rows[];   // [MW_SLOTS][E]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
    const int lane = threadIdx.x;
    if (lane >= E) return;               // barriers count waves, not lanes: the upper lanes of every wave simply retire
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= v.N) return;                // every wave of the workgroup holds the same envs, so all four agree on this
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    switch (role) {
        case 0: mw_role<M, GND, E, 0>(v, P, ap, actions_in, src, gnd, lds_rows, e, lane); break;
        case 1: mw_role<M, GND, E, 1>(v, P, ap, actions_in, src, gnd, lds_rows, e, lane); break;
        case 2: mw_role<M, GND, E, 2>(v, P, ap, actions_in, src, gnd, lds_rows, e, lane); break;
        default: mw_role<M, GND, E, 3>(v, P, ap, actions_in, src, gnd, lds_rows, e, lane); break;
    }
}

// ------------------------------------------------------------------------------------------------ all sub-steps of a step in ONE launch
// Option "fused_sub": the n sub-steps of a control step run inside one launch of the limb-per-wave kernel.  Between two sub-steps nothing
// goes through HBM: every role keeps its own q / qd (and the efforts, the clamped actions) in registers, last sub-step's impulses of its
// own rows stay where the sweeps left them in the LDS row store (substep_role<R, KEEP>, stage 2), and the one thing the other roles need
// -- the new root state, integrated by the trunk role -- crosses through 13 LDS slots behind the exchange area, with one barrier.  Same
// arithmetic per sub-step as the one-launch-per-sub-step form (bit-identical state: tests/test_gpu_multi_wave.py); the per-launch costs
// (dispatch, kernarg + state loads at HBM latency, the LDS-direct warm-start loads, the drain of the stores) are paid once per control step.
// Efforts: sub-step 0 from `first`, the following ones from `rest`, the last `tail` from the efforts of the sub-step before them
// (AnymalTerrain: `decimation` sim steps with the PD torques re-evaluated on the current joint state, then the base class's simulate()
// with the last torques, anymal_terrain.py:443-451 + vec_task.py:379-382).
// The loop around the unrolled dynamics is what round 1 had to give up on the one-wave kernel (LLVM hoisted hundreds of model literals
// into SGPRs and spilled them, step_kernels.hpp); a role's body is a quarter of that, and native.build() still refuses SGPR spills.
template <class GND>
struct MwFusedArgs {
    View v;
    SimParams P;
    ActParams ap;
    const float* actions_in;
    int first, rest, n_sub, tail;
    GND gnd;
    // AnymalTerrain (height-field ground) only: the curriculum pre-pass of its post step -- sum of cx^2 + cy^2 over the envs that will reset this
    // step (tasks/anymal_step.hpp anymal_cmdnorm_env) -- evaluated by the trunk wave at the end of this launch instead of by a kernel of its own
    int cmdnorm = 0, cmdnorm_allow_knee = 0;
    float cmdnorm_max_len = 0.f;
};
template <class M, int E>
constexpr size_t mw_fused_lds_bytes() { return (size_t)(SimMW<M>::MW_SLOTS + 13) * E * sizeof(float); }
// ------------------------------------------------------------------------------------------------ ... and post_physics_step with them, on ALL FOUR waves
// Options "fused_sub" + "fused_post" (locomotion tasks on the plane): after the last sub-step every role wave does ITS share of post_physics_step
// on the state it holds in registers -- the in-kernel reset of its own dofs (the draws are a hash of (seed, env, episode, dof): separable), the
// observation columns of its own dofs / actions / foot sensor, the three reward terms of its own dofs (published through the dead tree-pass
// exchange area) -- and the trunk role the root part (heading / up projections, atan2), the reward and the flags.  The single-wave form below
// (substep_mw_post_kernel) put the whole post step behind the one wave every workgroup waits for and only paid where launches dominate; here
// the dof part runs on the three leg waves while the trunk wave does the quaternion part.  Same helpers and the same partial sums as
// loco_post_env (part[d & 3], two terms each, whichever role they come from): bit-identical buffers (tests/test_gpu_multi_wave.py).
template <class GND>
struct MwFusedPostArgs {
    MwFusedArgs<GND> f;
    LocoParams tp;
};
// S: the role's simulator (SimMW<M>; SimMWC<M> for the Humanoid, whose launch is its LAST sub-step's: mwc_kernels.hpp).  act: the clamped actions in
// the role's registers, or nullptr -- then they are read back from v.actions (stored by the step's first launch).  BAR_FIRST: a barrier before
// anything is written to xpost (an area some role may still be reading in its output phase).
template <class S, class M, bool HUM, int E, int R, bool BAR_FIRST = false>
__device__ __forceinline__ void loco_post_role(const View& v, const LocoParams& tp, S& sim, const float* act, const int e, float* xpost) {
    using T = Loco<M::ND, 6 * M::NSENS, HUM>;
    constexpr int ND = M::ND, NOBS = T::NOBS;
    const int N = v.N;
    if constexpr (BAR_FIRST) __syncthreads();
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    const bool do_reset = v.reset[e] != 0;
    int ep = v.episode[e];
    float* ob = v.obs + (size_t)e * NOBS;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * NOBS;
    const float c = v.clip_obs;
    auto put = [&](const int k, const float x) MI_LAMBDA { ob[k] = x; oc[k] = fminf(fmaxf(x, -c), c); };
    // ---- own dofs: reset, state write-out, observation columns, reward terms
    sfor<ND>([&](auto D) MI_LAMBDA {
        constexpr int d = D;
        if constexpr (S::template owns_gi<R>(M::OFF + d)) {
            if (do_reset) {
                T::reset_dof(tp, v.seed, genv, (uint32_t)ep, d, tp.initial_dof_pos[d], tp.dof_lower[d], tp.dof_upper[d], &sim.q[d], &sim.qd[d]);
                v.laml[d * N + e] = 0.f;
            }
            v.dof[d * N + e] = sim.q[d];
            v.dof[(ND + d) * N + e] = sim.qd[d];
            float ps, vs, fs;
            // (joint forces, like the sensors below: stored by this wave's output phase, read back by the same lane)
            T::obs_dof(tp, sim.q[d], sim.qd[d], HUM ? v.dof_force[d * N + e] : 0.f, tp.dof_lower[d], tp.dof_upper[d], &ps, &vs, &fs);
            const float a_d = (act != nullptr) ? act[d] : v.actions[d * N + e];
            put(T::COL_POS + d, ps);
            put(T::COL_VEL + d, vs);
            if constexpr (HUM) put(T::COL_FORCE + d, fs);
            put(T::COL_ACT + d, a_d);
            typename T::DofSums one;
            T::reward_dof(tp, a_d, ps, vs, tp.gear[d], one);
            xpost[(3 * d + 0) * E] = one.actions; xpost[(3 * d + 1) * E] = one.electricity; xpost[(3 * d + 2) * E] = one.at_limit;
        }
    });
    // ---- own force sensors (stored by this wave's P5; a load after the own store of the same address sees it)
    sfor<M::NSENS>([&](auto K_) MI_LAMBDA {
        constexpr int k = K_;
        if constexpr (S::template owns_body<R>(M::sens_body[k]))
            sfor<6>([&](auto J) MI_LAMBDA { put(T::COL_SENS + 6 * k + J, v.sensor[(6 * k + J) * N + e] * tp.contact_force_scale); });
    });
    if (do_reset) sfor<3 * M::NSPH>([&](auto K) MI_LAMBDA { if constexpr (S::template owns_body<R>(M::sph_body[K / 3])) v.lamc[K * N + e] = 0.f; });
    if constexpr (R != M::TRUNK_ROLE) { __syncthreads(); return; }
    // ---- trunk role: root part, reward, flags
    long long progress = v.progress[e] + 1;
    float potentials = v.potentials[e], prev_potentials;
    if (do_reset) {
        float init_root[13];
        sfor<13>([&](auto K) MI_LAMBDA { init_root[K] = v.init_root[K * N + e]; sim.root[K] = init_root[K]; });
        const float pp = T::reset_potential(tp, init_root);
        potentials = pp;
        ep += 1;
        progress = 0;
        if constexpr (M::NPG > 0) { if (v.lamp) sfor<3 * M::NPG>([&](auto K) MI_LAMBDA { v.lamp[K * N + e] = 0.f; }); }
    }
    sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = sim.root[K]; });
    float up_vec[3], heading_vec[3], o12[12];
    T::obs_root(tp, sim.root, tp.targets, potentials, tp.inv_start_rot, tp.basis_vec0, tp.basis_vec1, o12, &potentials, &prev_potentials, up_vec, heading_vec);
    sfor<12>([&](auto K) MI_LAMBDA { put(K, o12[K]); });
    __syncthreads();
    typename T::DofSums part[4];
    sfor<ND>([&](auto D) MI_LAMBDA {
        constexpr int d = D;
        part[d & 3].actions += xpost[(3 * d + 0) * E]; part[d & 3].electricity += xpost[(3 * d + 1) * E]; part[d & 3].at_limit += xpost[(3 * d + 2) * E];
    });
    typename T::DofSums sm;
    sm.actions = (part[0].actions + part[1].actions) + (part[2].actions + part[3].actions);
    sm.electricity = (part[0].electricity + part[1].electricity) + (part[2].electricity + part[3].electricity);
    sm.at_limit = (part[0].at_limit + part[1].at_limit) + (part[2].at_limit + part[3].at_limit);
    float rew;
    long long reset;
    T::reward_total(tp, o12[0], o12[10], o12[11], sm, 0LL, progress, potentials, prev_potentials, &rew, &reset);
    episode_stats<E, true>(v, e, true, rew, reset, progress);
    v.randomize[e] += 1;
    v.episode[e] = ep;
    v.potentials[e] = potentials;
    v.prev_potentials[e] = prev_potentials;
    sfor<3>([&](auto K) MI_LAMBDA { v.up_vec[K * N + e] = up_vec[K]; v.heading_vec[K * N + e] = heading_vec[K]; });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)(((float)progress >= tp.max_episode_length - 1.f) && (reset != 0));   // vec_task.py:394
}

template <class M, class GND, int E, int R, bool POST = false, bool HUM = false>
__device__ __forceinline__ void mw_role_fused(const MwFusedArgs<GND>& a, float* lds_rows, const int e, const int lane, const LocoParams* tp = nullptr) {
    using S = SimMW<M>;
    constexpr int ND = M::ND;
    const View& v = a.v;
    const ActParams& ap = a.ap;
    const int N = v.N;
    S sim;
#if defined(MI_TIMING)
    // tools/debug/mw_phases.py: [32] stamps per (workgroup, role) from lane 0 -- 30: role entered, 8 s + 0 .. 6: sub-step s, 31: post step done
    unsigned long long* const tstamp = (lane == 0 && g_mi_tstamp_mw != nullptr) ? g_mi_tstamp_mw + ((size_t)blockIdx.x * 4 + R) * 32 : nullptr;
    MI_STAMP(30);
#endif
    load_sim(sim, v, e);
    load_actor_scales(sim, v, e);
    {   // last step's impulses of the own rows: HBM -> row-store slots, LDS-direct (as in mw_role)
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        float* slot0 = lds_rows;
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            if constexpr (M::dof_limited[d] && S::template owns_gi<R>(M::OFF + d)) {
                constexpr int o = Sim<M>::stage_slot_lim(d) * E;
                __builtin_amdgcn_global_load_lds((gptr_t)(v.laml + (size_t)d * N + e), (lptr_t)(slot0 + o), 4, 0, 0);
            }
        });
        sfor<3 * M::NSPH>([&](auto K) MI_LAMBDA {
            if constexpr (S::template owns_body<R>(M::sph_body[K / 3])) {
                constexpr int o = Sim<M>::stage_slot_con(K) * E;
                __builtin_amdgcn_global_load_lds((gptr_t)(v.lamc + (size_t)K * N + e), (lptr_t)(slot0 + o), 4, 0, 0);
            }
        });
    }
    const float h = a.P.dt / (float)a.P.substeps;
    const Strided lamc{v.lamc + e, N}, laml{v.laml + e, N}, sensor{v.sensor + e, N}, dof_force{v.dof_force + e, N};
    const Strided netf{GND::NETF ? v.netf + e : nullptr, N};
    const float mu_env = (GND::HEIGHTFIELD || v.friction != nullptr) ? v.friction[e] : -1.f;
    float* xroot = lds_rows + (size_t)S::MW_SLOTS * E + lane;     // [13][E] the new root state, trunk role -> the others
    float tau[M::NDA], act[M::NDA];
    sfor<ND>([&](auto K) MI_LAMBDA { tau[K] = 0.f; act[K] = 0.f; });
    for (int i = 0; i < a.n_sub; ++i) {
        const int src = (i >= a.n_sub - a.tail) ? (i == 0 ? (int)ACT_STORED_TAU : -1) : (i == 0 ? a.first : a.rest);   // -1: keep the efforts
        if (src == ACT_STORED_TAU) {
            if (i == 0) sfor<ND>([&](auto K) MI_LAMBDA { tau[K] = v.tau[K * N + e]; });
        } else if (src > 0) {
            // (only the efforts of the dofs this role sees matter to it; it stores the ones it owns)
            sfor<ND>([&](auto K) MI_LAMBDA {
                constexpr int k = K;
                constexpr bool mine = S::template owns_gi<R>(M::OFF + k);
                float t = 0.f;
                if (k < ap.nact) {
                    float x;
                    if (src == ACT_FROM_ACTIONS) {
                        x = a.actions_in[(size_t)e * ap.nact + k];
                        x = fminf(fmaxf(x, -ap.clip), ap.clip);
                        if constexpr (mine) v.actions[k * N + e] = x;
                        act[k] = x;
                    } else if (i == 0) {
                        x = v.actions[k * N + e];
                        act[k] = x;
                    } else {
                        x = act[k];
                    }
                    if (ap.mode == 0) {
                        t = x * ap.gear[k] * ap.scale;
                    } else {
                        float qk = sim.q[k], qdk = sim.qd[k];
                        if constexpr (GND::HEIGHTFIELD) {      // AnymalTerrain: the launch is one control step of the task, whose first PD evaluation reads
                            if (i == 0 && v.dof_api != nullptr) { qk = v.dof_api[k * N + e]; qdk = v.dof_api[(ND + k) * N + e]; }    // the dof-state tensor of its last refresh
                        }
                        const float u = ap.kp * (ap.scale * x + ap.gear[k] - qk) - ap.kd * qdk;
                        t = fminf(fmaxf(u, -ap.torque_limit), ap.torque_limit);
                    }
                }
                tau[k] = t;
                if constexpr (mine) v.tau[k * N + e] = t;
            });
        }
#if defined(MI_TIMING)
        sim.tstamp = (tstamp != nullptr && i < 3) ? tstamp + 8 * i : nullptr;
#endif
        sim.template substep_role<R, true>(a.P, tau, h, RowStore<E>{lds_rows + lane}, lamc, laml, sensor, dof_force, a.gnd, mu_env, netf,
                                           i == 0 ? 1 : 2, DevBarrier{});
        if constexpr (GND::HEIGHTFIELD) {
            if (i == a.n_sub - a.tail - 1 && v.dof_api != nullptr)       // (uniform) the task's refresh_dof_state_tensor at the end of its decimation loop
                sfor<ND>([&](auto K) MI_LAMBDA {
                    if constexpr (S::template owns_gi<R>(M::OFF + K)) { v.dof_api[K * N + e] = sim.q[K]; v.dof_api[(ND + K) * N + e] = sim.qd[K]; }
                });
        }
        if (i + 1 < a.n_sub) {
            if constexpr (R == M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { xroot[K * E] = sim.root[K]; });
            __syncthreads();
            if constexpr (R != M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = xroot[K * E]; });
        }
    }
    if constexpr (GND::HEIGHTFIELD && !POST) {
        if (a.cmdnorm != 0) {                  // (uniform) every role's net contact forces of the last sub-step are in memory behind this barrier
            __syncthreads();
            if constexpr (R == M::TRUNK_ROLE) {
                bool rs = anymal_netf_norm(v, e, 0) > 1.f;
                if (a.cmdnorm_allow_knee == 0)
                    for (int k = 0; k < 4; ++k) rs = rs || (anymal_netf_norm(v, e, anymal_knee_body(k)) > 1.f);
                if (v.progress[e] + 1 >= (long long)a.cmdnorm_max_len - 1) rs = true;
                float acc = 0.f;
                if (rs) { const float cx = v.commands[e], cy = v.commands[N + e]; acc = cx * cx + cy * cy; }
                acc = wave_sum_live<E>(acc);
                if ((threadIdx.x & 63) == 0 && acc > 0.f) atomicAdd(v.ep_stats + 15, acc);
            }
        }
    }
    if constexpr (POST) {
        static_assert(3 * M::ND <= 16 * S::NLR, "the reward terms fit the (by now dead) tree-pass exchange area");
        loco_post_role<S, M, HUM, E, R>(v, *tp, sim, act, e, lds_rows + (size_t)S::X_LR * E + lane);
        MI_STAMP(31);
        return;
    }
    sfor<ND>([&](auto K) MI_LAMBDA {
        if constexpr (S::template owns_gi<R>(M::OFF + K)) {
            v.dof[K * N + e] = sim.q[K];
            v.dof[(ND + K) * N + e] = sim.qd[K];
        }
    });
    if constexpr (R == M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = sim.root[K]; });
}
// E = 8 (option multi_wave 8, the Ant): TWO workgroups per CU -- every SIMD holds two role waves (of different workgroups) and issues one while the
// other waits on LDS / memory; the kernel is compiled for two waves per SIMD (<= 256 registers per lane, VGPRs + AGPRs).
template <class M, class GND, int E, bool HUM>
__global__ __launch_bounds__(64 * M::NROLE, E == 8 ? 2 : 1) void substep_mw_fused_post_kernel(MwFusedPostArgs<GND> args_by_value) {
    extern __shared__ float lds_rows[];   // [MW_SLOTS + 13][E]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    const MwFusedPostArgs<GND>& a = *reinterpret_cast<const MwFusedPostArgs<GND>*>(__builtin_amdgcn_kernarg_segment_ptr());
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.f.v.N) return;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    switch (role) {
        case 0: mw_role_fused<M, GND, E, 0, true, HUM>(a.f, lds_rows, e, lane, &a.tp); break;
        case 1: mw_role_fused<M, GND, E, 1, true, HUM>(a.f, lds_rows, e, lane, &a.tp); break;
        case 2: mw_role_fused<M, GND, E, 2, true, HUM>(a.f, lds_rows, e, lane, &a.tp); break;
        default: mw_role_fused<M, GND, E, 3, true, HUM>(a.f, lds_rows, e, lane, &a.tp); break;
    }
#endif
}
template <class M, class GND, int E>
__global__ __launch_bounds__(64 * M::NROLE) void substep_mw_fused_kernel(MwFusedArgs<GND> args_by_value) {
    extern __shared__ float lds_rows[];   // [MW_SLOTS + 13][E]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    const MwFusedArgs<GND>& a = *reinterpret_cast<const MwFusedArgs<GND>*>(__builtin_amdgcn_kernarg_segment_ptr());
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.v.N) return;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    switch (role) {
        case 0: mw_role_fused<M, GND, E, 0>(a, lds_rows, e, lane); break;
        case 1: mw_role_fused<M, GND, E, 1>(a, lds_rows, e, lane); break;
        case 2: mw_role_fused<M, GND, E, 2>(a, lds_rows, e, lane); break;
        default: mw_role_fused<M, GND, E, 3>(a, lds_rows, e, lane); break;
    }
#endif
}

template <class M, class GND>
hipError_t launch_substeps_mw(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                              hipStream_t s, const GND& gnd, int tail, const MwCmdNormTail* cn) {
    const dim3 block(64, M::NROLE);
    if (v.fused_sub != 0 && n_sub > 1) {
        static unsigned long long fconf16 = 0ull, fconf32 = 0ull;
        constexpr size_t flds16 = mw_fused_lds_bytes<M, 16>(), flds32 = mw_fused_lds_bytes<M, 32>();
        MwFusedArgs<GND> fa{v, P, ap, actions, first, rest, n_sub, tail, gnd};
        if (cn != nullptr) { fa.cmdnorm = 1; fa.cmdnorm_allow_knee = cn->allow_knee_contacts; fa.cmdnorm_max_len = cn->max_episode_length; }
        if (MI_MW_HAS16 && v.mw == 16) {
            auto kern = substep_mw_fused_kernel<M, GND, MI_MW_HAS16 ? 16 : 32>;
            if (hipError_t e = ensure_dynamic_lds((const void*)kern, flds16, &fconf16); e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(xcd_grid<16>(v.N)), block, flds16, s, fa);
        } else {
            auto kern = substep_mw_fused_kernel<M, GND, 32>;
            if (hipError_t e = ensure_dynamic_lds((const void*)kern, flds32, &fconf32); e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(xcd_grid<32>(v.N)), block, flds32, s, fa);
        }
        return hipGetLastError();
    }
    static unsigned long long conf16 = 0ull, conf32 = 0ull;
    constexpr size_t lds16 = mw_lds_bytes<M, 16>(), lds32 = mw_lds_bytes<M, 32>();
    // (one launch per sub-step; the last `tail` ones run on the stored efforts)
    auto src_of = [&](int i) { return i >= n_sub - tail ? (int)ACT_STORED_TAU : (i == 0 ? first : rest); };
    if (MI_MW_HAS16 && v.mw == 16) {
        auto kern = substep_mw_kernel<M, GND, MI_MW_HAS16 ? 16 : 32>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds16, &conf16); e != hipSuccess) return e;
        const dim3 grid(xcd_grid<16>(v.N));
        for (int i = 0; i < n_sub; ++i) hipLaunchKernelGGL(kern, grid, block, lds16, s, v, P, ap, actions, src_of(i), gnd);
    } else {
        auto kern = substep_mw_kernel<M, GND, 32>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds32, &conf32); e != hipSuccess) return e;
        const dim3 grid(xcd_grid<32>(v.N));
        for (int i = 0; i < n_sub; ++i) hipLaunchKernelGGL(kern, grid, block, lds32, s, v, P, ap, actions, src_of(i), gnd);
    }
    return hipGetLastError();
}

// The step's LAST sub-step with post_physics_step fused in (locomotion tasks on the limb-per-wave form): the four role waves finish the
// sub-step as above, leave the new root / dof state in the (by then dead) tree-pass exchange area, meet at one more barrier, and ONE wave
// runs loco_post_env for the workgroup's envs -- progress, in-kernel reset, observations, reward, write-out -- instead of a kernel of its
// own on 64 waves (Ant@4096: 10.3 us of a 42 us step).  Force sensors, joint forces and actions come from global memory (written by this
// or an earlier launch, never read before in this one: no stale L1 lines; __syncthreads orders the stores).
// (arguments as ONE struct read through the kernarg segment pointer: as by-value parameters the ~190 dwords of LocoParams + View would be
// preloaded into SGPRs and spill, see mwc_kernels.hpp MwcArgs)
template <class GND>
struct MwPostArgs {
    View v;
    SimParams P;
    ActParams ap;
    const float* actions_in;
    int src;
    GND gnd;
    LocoParams tp;
};
template <class M, class GND, int E, bool HUM>
__global__ __launch_bounds__(64 * M::NROLE) void substep_mw_post_kernel(MwPostArgs<GND> args_by_value) {
    extern __shared__ float lds_rows[];   // [MW_SLOTS][E]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
    static_assert(13 + 2 * M::ND <= 16 * SimMW<M>::NLR, "the new state fits the tree-pass exchange area");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    const MwPostArgs<GND>& a = *reinterpret_cast<const MwPostArgs<GND>*>(__builtin_amdgcn_kernarg_segment_ptr());
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.v.N) return;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    float* post_x = lds_rows + (size_t)SimMW<M>::X_LR * E + lane;
    switch (role) {
        case 0: mw_role<M, GND, E, 0>(a.v, a.P, a.ap, a.actions_in, a.src, a.gnd, lds_rows, e, lane, post_x); break;
        case 1: mw_role<M, GND, E, 1>(a.v, a.P, a.ap, a.actions_in, a.src, a.gnd, lds_rows, e, lane, post_x); break;
        case 2: mw_role<M, GND, E, 2>(a.v, a.P, a.ap, a.actions_in, a.src, a.gnd, lds_rows, e, lane, post_x); break;
        default: mw_role<M, GND, E, 3>(a.v, a.P, a.ap, a.actions_in, a.src, a.gnd, lds_rows, e, lane, post_x); break;
    }
    __syncthreads();
    if (role != M::TRUNK_ROLE) return;
    float root[13], q[M::NDA], qd[M::NDA];
    sfor<13>([&](auto K) MI_LAMBDA { root[K] = post_x[K * E]; });
    sfor<M::ND>([&](auto K) MI_LAMBDA { q[K] = post_x[(13 + K) * E]; qd[K] = post_x[(13 + M::ND + K) * E]; });
    loco_post_env<M, HUM, E, true, false>(a.v, a.tp, e, true, root, q, qd);     // (launch_loco_step takes this form only with the observation noise off)
#endif
}

template <class M, bool HUM>
hipError_t launch_substeps_mw_post(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                                   hipStream_t s, const LocoParams& tp) {
    if constexpr (!HUM) {
        if (v.fused_sub != 0 && n_sub > 1) {        // every sub-step AND the post step in one launch, the post step spread over the four role waves
            static unsigned long long pconf16 = 0ull, pconf32 = 0ull;
            constexpr size_t plds16 = mw_fused_lds_bytes<M, 16>(), plds32 = mw_fused_lds_bytes<M, 32>();
            const MwFusedPostArgs<PlaneGround> pa{MwFusedArgs<PlaneGround>{v, P, ap, actions, first, rest, n_sub, 0, PlaneGround{}}, tp};
            const dim3 block(64, M::NROLE);
            if (MI_MW_HAS8 && v.mw == 8) {
                static unsigned long long pconf8 = 0ull;
                constexpr size_t plds8 = mw_fused_lds_bytes<M, 8>();
                auto kern = substep_mw_fused_post_kernel<M, PlaneGround, MI_MW_HAS8 ? 8 : 32, HUM>;
                if (hipError_t e = ensure_dynamic_lds((const void*)kern, plds8, &pconf8); e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(xcd_grid<8>(v.N)), block, plds8, s, pa);
            } else if (MI_MW_HAS16 && v.mw == 16) {
                auto kern = substep_mw_fused_post_kernel<M, PlaneGround, MI_MW_HAS16 ? 16 : 32, HUM>;
                if (hipError_t e = ensure_dynamic_lds((const void*)kern, plds16, &pconf16); e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(xcd_grid<16>(v.N)), block, plds16, s, pa);
            } else {
                auto kern = substep_mw_fused_post_kernel<M, PlaneGround, 32, HUM>;
                if (hipError_t e = ensure_dynamic_lds((const void*)kern, plds32, &pconf32); e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(xcd_grid<32>(v.N)), block, plds32, s, pa);
            }
            return hipGetLastError();
        }
    }
    if (n_sub > 1) {
        if (hipError_t e = launch_substeps_mw<M, PlaneGround>(v, P, ap, actions, n_sub - 1, first, rest, s, PlaneGround{}, 0); e != hipSuccess) return e;
    }
    const int src = n_sub > 1 ? rest : first;
    static unsigned long long conf16 = 0ull, conf32 = 0ull;
    constexpr size_t lds16 = mw_lds_bytes<M, 16>(), lds32 = mw_lds_bytes<M, 32>();
    const dim3 block(64, M::NROLE);
    if (MI_MW_HAS16 && v.mw == 16) {
        auto kern = substep_mw_post_kernel<M, PlaneGround, MI_MW_HAS16 ? 16 : 32, HUM>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds16, &conf16); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(xcd_grid<16>(v.N)), block, lds16, s, MwPostArgs<PlaneGround>{v, P, ap, actions, src, PlaneGround{}, tp});
    } else {
        auto kern = substep_mw_post_kernel<M, PlaneGround, 32, HUM>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds32, &conf32); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(xcd_grid<32>(v.N)), block, lds32, s, MwPostArgs<PlaneGround>{v, P, ap, actions, src, PlaneGround{}, tp});
    }
    return hipGetLastError();
}

}  // namespace mi

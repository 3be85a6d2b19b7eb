// step_kernels.hpp -- the fused VecTask.step() kernels, templated on the generated robot model.
//
// Execution model: one environment per SIMD lane, 64 envs per wavefront, one wavefront per workgroup so that at
// the benchmark sizes (4096-8192 envs = 64-128 waves) every wave lands on its own CU and owns that CU's register
// file and LDS.  All persistent state is SoA [field][env] in the caller's arena => every state load/store of a
// wave is one fully coalesced 256-byte transaction.  A VecTask.step() (reference vec_task.py:360-408) is
// `substeps` physics launches (the first also clamps actions -> efforts) + one post launch (progress/reset ->
// observations -> reward -> timeout); the constraint rows of a sub-step are staged in LDS ([slot][lane]).
//
// Each robot model is instantiated in its own translation unit (kernels_<model>.hip) so the library builds in
// parallel; mi_engine.hip holds the C ABI and calls the launchers declared at the bottom of this file.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "core/engine.hpp"
#include "arena.hpp"
#include "tasks/locomotion.hpp"

namespace mi {

template <class M>
__device__ __forceinline__ void load_sim(Sim<M>& s, const View& v, int e) {
    const int N = v.N;
    sfor<13>([&](auto K) MI_LAMBDA { s.root[K] = v.root[K * N + e]; });
    sfor<M::ND>([&](auto K) MI_LAMBDA {
        s.q[K] = v.dof[K * N + e];
        s.qd[K] = v.dof[(M::ND + K) * N + e];
    });
}
// `actor_params` scale factors of this env (Ant / Humanoid; a null tensor or another model: the model's own constants)
template <class S>
__device__ __forceinline__ void load_actor_scales(S& s, const View& v, int e) {
    if constexpr (S::SCALED) {
        if (v.actor_scale != nullptr) s.actor_scale = Strided{v.actor_scale + e, v.N};
        if (v.limit_shift != nullptr) s.limit_shift = Strided{v.limit_shift + e, v.N};
    }
    if constexpr (S::FENCED) { if (v.alloc_fence != nullptr) s.fence = Strided{const_cast<float*>(v.alloc_fence) + e, v.N}; }   // never (Sim::alloc_fence)
}
template <class M>
__device__ __forceinline__ void store_sim(const Sim<M>& s, const View& v, int e) {
    const int N = v.N;
    sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = s.root[K]; });
    sfor<M::ND>([&](auto K) MI_LAMBDA {
        v.dof[K * N + e] = s.q[K];
        v.dof[(M::ND + K) * N + e] = s.qd[K];
    });
}

// ------------------------------------------------------------------------------------------------ episode statistics
// Per-step episode bookkeeping fused into the step kernel: finished-episode return/length sums are reduced across
// the 64 lanes of the wave with DPP/ds_swizzle shuffles and land in HBM with one atomic per wave and statistic.
// These five floats are the only thing ever all-reduced across GPUs (RCCL, parallel.py).
template <int ACTIVE = 64>      // lanes 0 .. ACTIVE-1 of the wave hold data (a 32-thread workgroup is a half-filled wave64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = ACTIVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same where lanes of the wave have already retired (the fused post step of the limb-per-wave kernels: the lanes of a partly filled last
// workgroup return at the top of the kernel): a retired lane's value is not defined, so it is masked by the wave's exec mask
template <int ACTIVE = 64>
__device__ __forceinline__ float wave_sum_live(float v) {
    const unsigned long long live = __builtin_amdgcn_ballot_w64(true);
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int o = ACTIVE / 2; o > 0; o >>= 1) {
        const float other = __shfl_xor(v, o, 64);
        v += ((live >> (lane ^ o)) & 1ull) ? other : 0.f;
    }
    return v;
}
// job statistics: fire-and-forget hardware float atomics (global_atomic_add_f32).  Plain atomicAdd(float*) compiles to a compare-and-swap loop
// here; every wave of a post kernel adds to the same five words, so the loops of 64-256 waves collided and retried.
__device__ __forceinline__ void stat_add(float* p, float x) { unsafeAtomicAdd(p, x); }
template <int ACTIVE = 64, bool LIVE_MASK = false>
__device__ __forceinline__ void episode_stats(const View& v, int e, bool valid, float rew, long long reset, long long progress) {
    float ret = 0.f, fin_ret = 0.f, fin_len = 0.f, fin = 0.f, r = 0.f, cnt = 0.f;
    if (valid) {
        ret = v.ep_ret[e] + rew;
        r = rew; cnt = 1.f;
        if (reset != 0) { fin_ret = ret; fin_len = (float)(progress + 1); fin = 1.f; ret = 0.f; }
        v.ep_ret[e] = ret;
    }
    if constexpr (LIVE_MASK) {
        fin_ret = wave_sum_live<ACTIVE>(fin_ret); fin_len = wave_sum_live<ACTIVE>(fin_len); fin = wave_sum_live<ACTIVE>(fin); r = wave_sum_live<ACTIVE>(r);
        cnt = wave_sum_live<ACTIVE>(cnt);
    } else {
        fin_ret = wave_sum<ACTIVE>(fin_ret); fin_len = wave_sum<ACTIVE>(fin_len); fin = wave_sum<ACTIVE>(fin); r = wave_sum<ACTIVE>(r); cnt = wave_sum<ACTIVE>(cnt);
    }
    if ((threadIdx.x & 63) == 0) {
        if (fin > 0.f) { stat_add(v.stats + 0, fin_ret); stat_add(v.stats + 1, fin_len); stat_add(v.stats + 2, fin); }
        stat_add(v.stats + 3, r);
        stat_add(v.stats + 4, cnt);
    }
}

// ------------------------------------------------------------------------------------------------ physics sub-step
// How a control step is cut into launches (all on the caller's stream, no host sync):
//     substep_kernel x (control_freq_inv * substeps)   -- gym.simulate(), reference vec_task.py:379-382
//     post kernel                                      -- post_physics_step + timeout mask, vec_task.py:389-394
// The first substep launch of a step also performs pre_physics_step (clamp actions -> efforts, ant.py:281-285).
// One sub-step per launch keeps the kernel body loop-free around the fully unrolled dynamics: with the sub-step
// loop inside, LLVM hoists hundreds of model literals (gfx9 VOP3 cannot encode literals, each needs an SGPR) out
// of it and spills SGPRs; that build was observed to return run-to-run different results on gfx950.
struct ActParams {   // pre_physics_step
    // mode 0 (effort, ant.py:281-285): tau[d] = clamp(a[d], +-clip) * gear[d] * scale for d < nact, else 0
    // mode 1 (PD position targets recomputed every sim step, anymal_terrain.py:443-446):
    //         tau[d] = clip(kp * (scale * a[d] + default_pos[d] - q[d]) - kd * qd[d], +-torque_limit)
    float clip, scale;
    int nact;
    int mode;
    float kp, kd, torque_limit;
    float gear[kMaxDof];         // mode 1: default_pos
};
// where a sub-step takes its efforts from
enum ActSource { ACT_STORED_TAU = 0,      // v.tau as left by an earlier launch / gym.set_dof_actuation_force_tensor
                 ACT_FROM_ACTIONS = 1,    // clamp the caller's row-major actions, store them in v.actions, derive tau
                 ACT_FROM_STORED_ACTIONS = 2 };  // PD mode: re-derive tau from v.actions and the current joint state

// XCD-aware env mapping of workgroups that hold E < 64 envs.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
// workgroup (x, j) = (id % 8, id / 8) takes the envs 64 * (8 * (j / SUBS) + x) + E * (j % SUBS) ...: every aligned group of 64 envs lives on
// one XCD, whichever kernel (sub-step with 16 / 32 envs per workgroup, multi-wave sub-step, 32- or 64-lane post kernel) touches it next.
// (A post kernel on another XCD than the sub-step made the next sub-step launch pull its state across L2s: +25 %, round 1.)
template <int E>
__device__ __forceinline__ int xcd_env_base(int wg) {
    if constexpr (E >= 64) {
        return wg * E;
    } else {
        constexpr int SUBS = 64 / E;
        const int x = wg & 7, j = wg >> 3;
        return 64 * (8 * (j / SUBS) + x) + E * (j % SUBS);
    }
}
template <int E>
inline int xcd_grid(int N) {           // workgroups, rounded up so that the (x, j) mapping covers every env
    if constexpr (E >= 64) {
        return (N + E - 1) / E;
    } else {
        constexpr int SUBS = 64 / E;
        return (((N + 63) / 64 + 7) / 8) * 8 * SUBS;
    }
}

// envs per workgroup (= per wave): 64, or 16 for models on the compact contact store (Sim<M>::COMPACT)
template <class M>
constexpr size_t lds_bytes() { return (size_t)Sim<M>::ROW_SLOTS * Sim<M>::LANES * sizeof(float); }
template <class M>
constexpr bool rows_fit_lds() { return lds_bytes<M>() <= 160 * 1024; }

// efforts of one env for this sub-step: from the policy's actions (first launch of a step: clamp, noise, gear / PD), from the stored
// clamped actions (PD re-evaluated every decimation sub-step) or the stored efforts
template <class M>
__device__ __forceinline__ void efforts_for_substep(const View& v, const ActParams& ap, const float* __restrict__ actions_in, const int src,
                                                    const int e, const Sim<M>& sim, float* tau) {
    constexpr int ND = M::ND;
    const int N = v.N;
    if (src != ACT_STORED_TAU) {         // uniform branch
        sfor<ND>([&](auto K) MI_LAMBDA {
            constexpr int k = K;
            float t = 0.f;
            if (k < ap.nact) {
                float a;
                if (src == ACT_FROM_ACTIONS) {
                    a = actions_in[(size_t)e * ap.nact + k];
                    a = fminf(fmaxf(a, -ap.clip), ap.clip);  // vec_task.py:374
                    v.actions[k * N + e] = a;
                } else {
                    a = v.actions[k * N + e];
                }
                if (ap.mode == 0) {
                    t = a * ap.gear[k] * ap.scale;
                } else {
                    // (the drive's gains are the dofs' `stiffness` / `damping` properties: their `actor_params` factors scale kp / kd, Anymal.yaml:146-158)
                    float kp = ap.kp, kd = ap.kd;
                    if constexpr (Sim<M>::SCALED) {
                        if (sim.actor_scale.p != nullptr) { kp *= sim.actor_scale(Sim<M>::AS_STIFF + k); kd *= sim.actor_scale(Sim<M>::AS_DAMP + k); }
                    }
                    const float u = kp * (ap.scale * a + ap.gear[k] - sim.q[k]) - kd * sim.qd[k];
                    t = fminf(fmaxf(u, -ap.torque_limit), ap.torque_limit);
                }
            }
            tau[k] = t;
            v.tau[k * N + e] = t;
        });
    } else {
        sfor<ND>([&](auto K) MI_LAMBDA { tau[K] = v.tau[K * N + e]; });
    }
}
// Action noise of the domain randomisation (vec_task.py:371-372, then the clamp of :374): a kernel of its own, launched only when the noise is
// on, that leaves the noisy clamped actions in v.actions; the step's first sub-step launch then takes ACT_FROM_STORED_ACTIONS.  Inlined into
// the sub-step kernels the never-taken noise block was two Box-Muller draws (logf / cosf with their argument reduction) per dof and role:
// 3 k (Ant) to 8 k (Humanoid) dead instructions at the top of every role's stream.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void act_noise_kernel(View v, const float* __restrict__ actions_in, float clip, int nact) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= v.N * nact) return;
    const int e = i / nact, k = i - e * nact;
    float a = apply_noise(v.act_noise, v.seed, (uint32_t)(v.env_offset + e), v.step, 1u, (uint32_t)k, actions_in[i]);
    v.actions[(size_t)k * v.N + e] = fminf(fmaxf(a, -clip), clip);
}
// to be called before the sub-step launches of a step: returns the effort source of the first launch
inline int prepare_actions(const View& v, const ActParams& ap, const float* actions, int first, hipStream_t s) {
    if (first != ACT_FROM_ACTIONS || v.act_noise.dist == 0) return first;
    hipLaunchKernelGGL(act_noise_kernel<0>, dim3((v.N * ap.nact + 255) / 256), dim3(256), 0, s, v, actions, ap.clip, ap.nact);
    return ACT_FROM_STORED_ACTIONS;
}
// Last sub-step's impulses go from HBM straight into their row-store slots with LDS-direct loads (global_load_lds_dword:
// lane l of the wave lands at slot base + 4 l, exactly the [slot][lane] layout): no VGPR holds them in flight, so the ~50
// (Ant) / ~130 (Humanoid) loads no longer push that many live values out of the register file at the top of the kernel.
template <class M, bool LIMITS = true>
__device__ __forceinline__ void prestage_warm_start(const View& v, const int e, float* lds_rows) {
    using S = Sim<M>;
    constexpr int LANES = S::LANES;
    const int N = v.N;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    if constexpr (LIMITS) sfor<M::ND>([&](auto D) MI_LAMBDA {
        constexpr int d = D;
        if constexpr (M::dof_limited[d])
            __builtin_amdgcn_global_load_lds((gptr_t)(v.laml + (size_t)d * N + e), (lptr_t)(lds_rows + S::stage_slot_lim(d) * LANES), 4, 0, 0);
    });
    sfor<3 * M::NSPH>([&](auto K) MI_LAMBDA {
        __builtin_amdgcn_global_load_lds((gptr_t)(v.lamc + (size_t)K * N + e), (lptr_t)(lds_rows + S::stage_slot_con(K) * LANES), 4, 0, 0);
    });
}

#if defined(MI_TIMING)
// debug builds only (tools/debug/phase_timing_live.py): per-workgroup s_memtime stamps of the sub-step phases, 16 slots per workgroup
__device__ unsigned long long* g_mi_tstamp = nullptr;
#endif
template <class M, class GND>
__global__ __launch_bounds__(64) void substep_kernel(View v, SimParams P, ActParams ap, const float* __restrict__ actions_in, int src,
                                                     GND gnd) {
    extern __shared__ float lds_rows[];  // [ROW_SLOTS][LANES] when the model's rows fit (else unused, size 0)
    constexpr int ND = M::ND, LANES = Sim<M>::LANES;
    const int e = xcd_env_base<LANES>(blockIdx.x) + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;                  // no cross-lane operation in here: tail lanes simply retire
    Sim<M> sim;
    load_sim(sim, v, e);
    load_actor_scales(sim, v, e);
#if defined(MI_TIMING)
    sim.tstamp = (threadIdx.x == 0 && g_mi_tstamp != nullptr) ? g_mi_tstamp + (size_t)blockIdx.x * 16 : nullptr;   // tools/debug/phase_timing_live.py
#endif
    float tau[M::NDA];
    efforts_for_substep<M>(v, ap, actions_in, src, e, sim, tau);
    constexpr bool PRESTAGE = rows_fit_lds<M>() && Sim<M>::STAGES_LAM;
    if constexpr (PRESTAGE) prestage_warm_start<M>(v, e, lds_rows);
    const float h = P.dt / (float)P.substeps;
    const Strided lamc{v.lamc + e, N}, laml{v.laml + e, N}, sensor{v.sensor + e, N}, dof_force{v.dof_force + e, N};
    const Strided netf{GND::NETF ? v.netf + e : nullptr, N};
    const float mu_env = (GND::HEIGHTFIELD || v.friction != nullptr) ? v.friction[e] : -1.f;   // per-env shape friction where the task has the tensor
    const SelfCol selfcol{Strided{v.lamp ? v.lamp + e : nullptr, N}, Strided{v.pairf ? v.pairf + e : nullptr, N}, v.dropped ? v.dropped + e : nullptr, N};
    const SelfCol* scp = (Sim<M>::NPG > 0 && v.lamp != nullptr) ? &selfcol : nullptr;   // uniform
    if constexpr (rows_fit_lds<M>()) {
        if constexpr (LANES == 64) sim.substep(P, tau, h, RowStore<64>(lds_rows + threadIdx.x), lamc, laml, sensor, dof_force, gnd, mu_env, netf, nullptr, PRESTAGE, scp);
        else sim.substep(P, tau, h, RowStore<LANES>{lds_rows + threadIdx.x}, lamc, laml, sensor, dof_force, gnd, mu_env, netf, nullptr, PRESTAGE, scp);
    } else {
        float rows[Sim<M>::ROW_SLOTS];
        sim.substep(P, tau, h, RowStore<1>{rows}, lamc, laml, sensor, dof_force, gnd, mu_env, netf);
    }
    store_sim(sim, v, e);
}

// ------------------------------------------------------------------------------------------------ multi-wave sub-step
// (kernels in mw_kernels.hpp, instantiated in their own translation units kernels_mw_<model>.hip: they take minutes to compile)
#ifndef MI_MW_HAS16
#define MI_MW_HAS16 1          // also build the 16-envs-per-workgroup variant (best at <= 4096 envs; 0 halves the compile time of kernels_mw_*.hip)
#endif
#ifndef MI_MW_HAS8
#define MI_MW_HAS8 0           // -DMI_MW_HAS8=1 (A/B builds): the 8-envs-per-workgroup, two-workgroups-per-CU variant of the Ant's one-launch kernel
#endif
template <class M, class GND>
constexpr bool mw_capable() {
    // (flat ground with net contact forces = the flat Anymal task: the plain model has the limb-per-wave form, kernels_mw_anymal.hip; its
    //  `actor_params` instantiation, Sim<Scaled<M>>, runs on one wave)
    return !Sim<M>::COMPACT && Sim<M>::LAM_IN_ROWS && !M::FIXED && M::NLIMB >= 3 && !(std::is_same<GND, PlaneGroundNF>::value && is_scaled<M>::value);
}
// launches n_sub sub-steps of a self-colliding robot on two waves per workgroup (sc2_kernels.hpp, instantiated in kernels_humanoid_sc2.hip)
template <class M>
hipError_t launch_substeps_sc2(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                               hipStream_t s);
// launches n_sub limb-per-wave sub-steps of a compact-store robot (mwc_kernels.hpp, instantiated in kernels_humanoid_mwc.hip)
template <class M>
hipError_t launch_substeps_mwc(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                               hipStream_t s);
// launches n_sub multi-wave sub-steps with v.mw envs per workgroup (defined for the model / ground pairs of kernels_mw_*.hip); the last
// `tail` of them run on the efforts of the sub-step before them; option "fused_sub": all of them in ONE launch (mw_kernels.hpp)
// `cn` (AnymalTerrain with the fused launch): the curriculum pre-pass of its post step runs on the trunk wave at the end of the launch
struct MwCmdNormTail { int allow_knee_contacts; float max_episode_length; };
template <class M, class GND>
hipError_t launch_substeps_mw(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                              hipStream_t s, const GND& gnd, int tail = 0, const MwCmdNormTail* cn = nullptr);

// XCD-aware env mapping of the 64-lane post kernels.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).
// A sub-step kernel with 32 envs per workgroup puts env e on XCD (e / 32) % 8; a post kernel that simply took envs
// 64 b .. 64 b + 63 in block b would put half of them on another XCD, and the next sub-step launch would then pull its
// whole state across L2s (measured: the first sub-step launch after the post kernel 256 us instead of 186 us, Humanoid@8192).
// Block b = 8 j + x therefore takes the two 32-env groups 16 j + x and 16 j + 8 + x, both of which live on XCD x.
template <int SUB_LANES>
__device__ __forceinline__ int post_env_index(int block, int lane, int N) {
    if constexpr (SUB_LANES == 64) {
        return block * 64 + lane;
    } else {
        static_assert(SUB_LANES == 32, "sub-step workgroups hold 32 or 64 envs");
        const int groups = (N + 31) / 32;                     // sub-step workgroups
        const int full = (groups / 16) * 8;                   // post blocks covered by the regular 16-group pattern
        if (block < full) {
            const int j = block >> 3, x = block & 7;
            const int g = 16 * j + x + ((lane >> 5) << 3);
            return g * 32 + (lane & 31);
        }
        return full * 64 + (block - full) * 64 + lane;        // ragged tail: plain mapping
    }
}

// ------------------------------------------------------------------------------------------------ post_physics_step
// Envs per post workgroup.  The Humanoid's post step stores 2 x 108 observation columns row-major, i.e. every store instruction of a full
// wave touches 64 cache lines; with 32-env workgroups (half-filled waves, one per sub-step workgroup, same XCD) each store touches 32 and
// twice as many CUs share the work: Humanoid step -0.6 % (fast box) to -2 % (slow box).  Ant (60 columns) showed no robust gain and keeps 64.
template <class M>
constexpr int post_lanes() { return Sim<M>::LANES < 64 ? 32 : 64; }
// post_physics_step of one env (ant.py:287-297): progress++, reset if flagged, observations, reward, write-out.  root / q / qd: the env's state
// after the last sub-step (the post kernel loads them; the fused form of the limb-per-wave sub-step hands them over through LDS).
// ACTIVE: lanes of the wave that hold envs (episode statistics reduction).
// What bounds this kernel (round 3, profiles/r3x_post_kernel_study.txt): not its instruction stream -- a form with FOUR lanes per env (a
// quarter of the instructions per lane, 4 x the waves, observation rows leaving through an LDS tile as whole cache lines, bit-identical
// buffers) ran 10.2 -> 11.8 us (Ant@4096) and 27.6 -> 29.6 us (Humanoid@8192); without the observation stores it takes 6.2 / 21.6 us,
// with non-temporal stores 13.0 / 32.7 us; hardware float atomics instead of the compare-and-swap loops of atomicAdd(float*) changed
// nothing measurable.  The one-lane form stays.
// NOISE: the kernel carries the observation noise of the domain randomisation.  It is a kernel of its own because the noise code is most
// of the kernel when it is there: two Box-Muller draws per observation column (logf / cosf with their full argument reduction, inlined per
// column) are 24 k of the Ant post kernel's 26 k vector instructions and 256 + 153 of its registers (Humanoid: 363 spilled registers) --
// never executed on a run without `task.randomize`, but fetched around and allocated for.
template <class M, bool HUM, int ACTIVE, bool LIVE_MASK = false, bool NOISE = true>
__device__ __forceinline__ void loco_post_env(const View& v, const LocoParams& tp, const int e, const bool valid, float (&root)[13], float (&q)[M::NDA],
                                              float (&qd)[M::NDA]) {
    using T = Loco<M::ND, 6 * M::NSENS, HUM>;
    constexpr int ND = M::ND, NOBS = T::NOBS;
    const int N = v.N;
    float dof_force[ND], sensor[6 * M::NSENS > 0 ? 6 * M::NSENS : 1], act[ND];
    sfor<ND>([&](auto K) MI_LAMBDA {
        dof_force[K] = v.dof_force[K * N + e];
        act[K] = v.actions[K * N + e];
    });
    sfor<6 * M::NSENS>([&](auto K) MI_LAMBDA { sensor[K] = v.sensor[K * N + e]; });
    // post_physics_step (ant.py:287-297): progress++, reset flagged envs, observations, reward
    long long progress = v.progress[e] + 1;
    float potentials = v.potentials[e], prev_potentials;
    int ep = v.episode[e];
    const bool do_reset = v.reset[e] != 0;
    if (do_reset) {
        float init_root[13];
        sfor<13>([&](auto K) MI_LAMBDA { init_root[K] = v.init_root[K * N + e]; });
        T::reset(tp, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, init_root, root, q, qd, &potentials, &prev_potentials);
        ep += 1;
        progress = 0;
        if (valid) {
            sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = root[K]; });
            sfor<ND>([&](auto K) MI_LAMBDA {
                v.dof[K * N + e] = q[K];
                v.dof[(ND + K) * N + e] = qd[K];
                v.laml[K * N + e] = 0.f;
            });
            sfor<3 * M::NSPH>([&](auto K) MI_LAMBDA { v.lamc[K * N + e] = 0.f; });
            if constexpr (M::NPG > 0) { if (v.lamp) sfor<3 * M::NPG>([&](auto K) MI_LAMBDA { v.lamp[K * N + e] = 0.f; }); }
        }
    }
    float up_vec[3], heading_vec[3];
    float rew;
    long long reset;
    if constexpr (!NOISE) {
        // Streaming form: every observation column is stored (raw into obs_buf, clamped into the ring slot) as soon as it exists, the
        // pass-through columns -- sensors, actions, then the dof columns -- first, so that the 2 x NOBS row-major stores (each instruction
        // touches one cache line per env) drain while the root part (quaternion products, atan2) is computed, instead of queueing up
        // behind it at the end of the kernel; nothing but the reward's partial sums stays live.  Same helpers, same sums as
        // Loco::observations / Loco::reward: bit-identical buffers.  (Lanes past the batch shadow the last env: their stores repeat its.)
        float* ob = v.obs + (size_t)e * NOBS;
        float* oc = v.obs_out + ((size_t)v.ring * N + e) * NOBS;
        const float c = v.clip_obs;
        auto put = [&](const int k, const float x) MI_LAMBDA { ob[k] = x; oc[k] = fminf(fmaxf(x, -c), c); };
        sfor<6 * M::NSENS>([&](auto K) MI_LAMBDA { put(T::COL_SENS + K, sensor[K] * tp.contact_force_scale); });
        sfor<ND>([&](auto D) MI_LAMBDA { put(T::COL_ACT + D, act[D]); });
        typename T::DofSums part[4];
        sfor<ND>([&](auto D) MI_LAMBDA {
            constexpr int d = D;
            float ps, vs, fs;
            T::obs_dof(tp, q[d], qd[d], HUM ? dof_force[d] : 0.f, tp.dof_lower[d], tp.dof_upper[d], &ps, &vs, &fs);
            put(T::COL_POS + d, ps);
            put(T::COL_VEL + d, vs);
            if constexpr (HUM) put(T::COL_FORCE + d, fs);
            T::reward_dof(tp, act[d], ps, vs, tp.gear[d], part[d & 3]);
        });
        float o12[12];
        T::obs_root(tp, root, tp.targets, potentials, tp.inv_start_rot, tp.basis_vec0, tp.basis_vec1, o12, &potentials, &prev_potentials, up_vec,
                    heading_vec);
        sfor<12>([&](auto K) MI_LAMBDA { put(K, o12[K]); });
        typename T::DofSums sm;
        sm.actions = (part[0].actions + part[1].actions) + (part[2].actions + part[3].actions);
        sm.electricity = (part[0].electricity + part[1].electricity) + (part[2].electricity + part[3].electricity);
        sm.at_limit = (part[0].at_limit + part[1].at_limit) + (part[2].at_limit + part[3].at_limit);
        T::reward_total(tp, o12[0], o12[10], o12[11], sm, 0LL, progress, potentials, prev_potentials, &rew, &reset);
        episode_stats<ACTIVE, LIVE_MASK>(v, e, valid, rew, reset, progress);
        if (!valid) return;
        v.randomize[e] += 1;
        v.episode[e] = ep;
        v.potentials[e] = potentials;
        v.prev_potentials[e] = prev_potentials;
        sfor<3>([&](auto K) MI_LAMBDA { v.up_vec[K * N + e] = up_vec[K]; v.heading_vec[K * N + e] = heading_vec[K]; });
        v.rew[e] = rew;
        v.reset[e] = reset;
        v.progress[e] = progress;
        v.timeout[e] = (unsigned char)(((float)progress >= tp.max_episode_length - 1.f) && (reset != 0));   // vec_task.py:394
        return;
    }
    float obs[NOBS];
    T::observations(tp, root, tp.targets, potentials, tp.inv_start_rot, q, qd, dof_force, tp.dof_lower, tp.dof_upper, sensor,
                    act, tp.basis_vec0, tp.basis_vec1, obs, &potentials, &prev_potentials, up_vec, heading_vec);
    T::reward(tp, obs, 0LL, progress, act, potentials, prev_potentials, &rew, &reset);
    // observation noise of the domain randomisation: the reference applies it to obs_buf after post_physics_step (vec_task.py:397-399),
    // i.e. the reward above saw the clean observations
    if constexpr (NOISE) {
        if (v.obs_noise.dist != 0)
            sfor<NOBS>([&](auto K) MI_LAMBDA { obs[K] = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + e), v.step, 0u, (uint32_t)K, obs[K]); });
    }
    episode_stats<ACTIVE, LIVE_MASK>(v, e, valid, rew, reset, progress);
    if (!valid) return;
    v.randomize[e] += 1;
    v.episode[e] = ep;
    v.potentials[e] = potentials;
    v.prev_potentials[e] = prev_potentials;
    sfor<3>([&](auto K) MI_LAMBDA { v.up_vec[K * N + e] = up_vec[K]; v.heading_vec[K * N + e] = heading_vec[K]; });
    float* ob = v.obs + (size_t)e * NOBS;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * NOBS;
    sfor<NOBS>([&](auto K) MI_LAMBDA {
        ob[K] = obs[K];
        oc[K] = fminf(fmaxf(obs[K], -v.clip_obs), v.clip_obs);
    });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    // vec_task.py:394
    v.timeout[e] = (unsigned char)(((float)progress >= tp.max_episode_length - 1.f) && (reset != 0));
}
template <class M, bool HUM, bool NOISE>
__global__ __launch_bounds__(post_lanes<M>()) void loco_post_kernel(View v, LocoParams tp) {
    constexpr int ND = M::ND, PL = post_lanes<M>();
    const int N = v.N;
    // xcd_grid rounds the grid up to whole XCD rounds: a workgroup entirely past the batch leaves here.  (It must not shadow env N - 1: its
    // loads of that env's reset / progress flags are not ordered against the real wave's end-of-kernel stores, and the streaming form below
    // stores as it goes -- a late shadow wave could overwrite the terminal observation with a post-reset one.)  Shadow lanes remain only
    // inside the partly filled wave, in lockstep with the real lane whose values they repeat.
    if (xcd_env_base<PL>(blockIdx.x) >= N) return;
    const int e0 = xcd_env_base<PL>(blockIdx.x) + threadIdx.x;
    const bool valid = e0 < N;           // tail lanes shadow the last env so wave reductions stay full
    const int e = valid ? e0 : N - 1;
    float root[13], q[M::NDA], qd[M::NDA];
    sfor<13>([&](auto K) MI_LAMBDA { root[K] = v.root[K * N + e]; });
    sfor<ND>([&](auto K) MI_LAMBDA {
        q[K] = v.dof[K * N + e];
        qd[K] = v.dof[(ND + K) * N + e];
    });
    loco_post_env<M, HUM, PL, false, NOISE>(v, tp, e, valid, root, q, qd);
}


template <class M>
__global__ __launch_bounds__(64) void cartpole_post_kernel(View v, CartpoleParams tp) {
    const int e0 = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    float q[2] = {v.dof[e], v.dof[N + e]}, qd[2] = {v.dof[2 * N + e], v.dof[3 * N + e]};
    long long progress = v.progress[e] + 1;  // cartpole.py:165-174
    int ep = v.episode[e];
    if (v.reset[e] != 0) {
        cartpole_reset(v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, q, qd);
        ep += 1;
        progress = 0;
        if (valid) {
            sfor<2>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = q[K]; v.dof[(2 + K) * N + e] = qd[K]; v.laml[K * N + e] = 0.f; });
        }
    }
    float obs[4] = {q[0], qd[0], q[1], qd[1]};  // cartpole.py:131-142
    float rew;
    long long reset;
    cartpole_reward(tp, obs[2], obs[3], obs[1], obs[0], 0LL, progress, &rew, &reset);
    if (v.obs_noise.dist != 0)
        sfor<4>([&](auto K) MI_LAMBDA { obs[K] = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + e), v.step, 0u, (uint32_t)K, obs[K]); });
    episode_stats(v, e, valid, rew, reset, progress);
    if (!valid) return;
    v.randomize[e] += 1;
    v.episode[e] = ep;
    float* ob = v.obs + (size_t)e * 4;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * 4;
    sfor<4>([&](auto K) MI_LAMBDA { ob[K] = obs[K]; oc[K] = fminf(fmaxf(obs[K], -v.clip_obs), v.clip_obs); });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)(((float)progress >= tp.max_episode_length - 1.f) && (reset != 0));
}

// ------------------------------------------------------------------------------------------------ indexed reset
template <class M, bool HUM>
__global__ void loco_reset_kernel(View v, LocoParams tp, const long long* __restrict__ ids, int n) {
    using T = Loco<M::ND, 6 * M::NSENS, HUM>;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i], N = v.N;
    if (e < 0 || e >= N) return;
    float init_root[13], root[13], q[M::ND], qd[M::ND], pot, prev;
    for (int k = 0; k < 13; ++k) init_root[k] = v.init_root[k * N + e];
    const int ep = v.episode[e];
    T::reset(tp, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, init_root, root, q, qd, &pot, &prev);
    v.episode[e] = ep + 1;
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
    for (int k = 0; k < M::ND; ++k) {
        v.dof[k * N + e] = q[k];
        v.dof[(M::ND + k) * N + e] = qd[k];
        v.laml[k * N + e] = 0.f;
    }
    for (int k = 0; k < 3 * M::NSPH; ++k) v.lamc[k * N + e] = 0.f;
    if (M::NPG > 0 && v.lamp) for (int k = 0; k < 3 * M::NPG; ++k) v.lamp[k * N + e] = 0.f;
    v.potentials[e] = pot;
    v.prev_potentials[e] = prev;
    v.progress[e] = 0;
    v.reset[e] = 0;
}
template <class M>
__global__ void cartpole_reset_kernel(View v, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i], N = v.N;
    if (e < 0 || e >= N) return;
    float q[2], qd[2];
    const int ep = v.episode[e];
    cartpole_reset(v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, q, qd);
    v.episode[e] = ep + 1;
    for (int k = 0; k < 2; ++k) { v.dof[k * N + e] = q[k]; v.dof[(2 + k) * N + e] = qd[k]; v.laml[k * N + e] = 0.f; }
    v.progress[e] = 0;
    v.reset[e] = 0;
}

// Kernels that need more than 48 KB of dynamic LDS must opt in once per device (hipFuncSetAttribute is a per-device setting; a
// process normally drives one GPU, but nothing here should break if it drives several).
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, unsigned long long* configured_mask) {
    if (bytes <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (*configured_mask & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) *configured_mask |= bit;
    return e;
}

// launch `n_sub` physics sub-steps.  `first`: effort source of the first launch, `rest`: of the following ones.
// the same on Sim<Scaled<M>>, the instantiation that reads the `actor_params` factor tensors (scaled_kernels.hpp; instantiated in
// kernels_scaled_<model>.hip for the models with M::ACTOR_SCALES)
template <class M, class GND>
hipError_t launch_substeps_scaled(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first,
                                  int rest, hipStream_t s, const GND& gnd);
template <class M, class GND = PlaneGround>
hipError_t launch_substeps(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first_in,
                           int rest, hipStream_t s, const GND& gnd = GND{}) {
    const int first = prepare_actions(v, ap, actions, first_in, s);
    if constexpr (M::ACTOR_SCALES != 0 && !is_scaled<M>::value) {
        // option actor_tensors: the arena holds per-env mass / joint-constant factors and limit shifts -> the kernels that read them
        if (v.actor_scale != nullptr || v.limit_shift != nullptr) return launch_substeps_scaled<M, GND>(v, P, ap, actions, n_sub, first, rest, s, gnd);
    }
    if constexpr (Sim<M>::NPG > 0 && std::is_same<GND, PlaneGround>::value) {
        // self-colliding robot on the compact store.  multi_wave 32 (default): one limb per wave, every wave sweeping its own rows
        // (kernels_<model>_mwc.hip, round 3); 2: round 2's form, the self-collision phase on a helper wave beside one main wave
        // (kernels_<model>_sc2.hip; kept for A/B runs and as the Gauss-Seidel-order reference on the GPU)
        if (v.mw == 2 && v.lamp != nullptr) return launch_substeps_sc2<M>(v, P, ap, actions, n_sub, first, rest, s);
        if (v.mw != 0 && v.mw != 2 && ap.mode == 0) return launch_substeps_mwc<M>(v, P, ap, actions, n_sub, first, rest, s);
    }
    if constexpr (mw_capable<M, GND>()) {
        if (v.mw != 0) return launch_substeps_mw<M, GND>(v, P, ap, actions, n_sub, first, rest, s, gnd);
    }
    constexpr size_t lds = rows_fit_lds<M>() ? lds_bytes<M>() : 0;
    constexpr int LANES = Sim<M>::LANES;
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)substep_kernel<M, GND>, lds, &configured); e != hipSuccess) return e;
    for (int i = 0; i < n_sub; ++i)
        hipLaunchKernelGGL((substep_kernel<M, GND>), dim3(xcd_grid<LANES>(v.N)), dim3(LANES), lds, s, v, P, ap, actions, i == 0 ? first : rest, gnd);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- launchers
// One set per task, defined in kernels_<task>.hip.  All enqueue on `stream` and return hipGetLastError().
// launch_step_*: the whole VecTask.step(); launch_simulate_*: one gym.simulate() with the efforts in v.tau.
hipError_t launch_step_cartpole(const View& v, const SimParams& P, const CartpoleParams& tp, const float* actions, int cfi, hipStream_t s);
hipError_t launch_simulate_cartpole(const View& v, const SimParams& P, hipStream_t s);
hipError_t launch_reset_cartpole(const View& v, const long long* ids, int n, hipStream_t s);
hipError_t launch_step_ant(const View& v, const SimParams& P, const LocoParams& tp, const float* actions, int cfi, hipStream_t s);
hipError_t launch_simulate_ant(const View& v, const SimParams& P, hipStream_t s);
hipError_t launch_reset_ant(const View& v, const LocoParams& tp, const long long* ids, int n, hipStream_t s);
hipError_t launch_step_humanoid(const View& v, const SimParams& P, const LocoParams& tp, const float* actions, int cfi, hipStream_t s);
hipError_t launch_simulate_humanoid(const View& v, const SimParams& P, hipStream_t s);
hipError_t launch_reset_humanoid(const View& v, const LocoParams& tp, const long long* ids, int n, hipStream_t s);

// the limb-per-wave sub-steps of a control step with post_physics_step FUSED into the last launch (mw_kernels.hpp; defined for the models of
// kernels_mw_*.hip that run a locomotion task): one wave of every workgroup runs loco_post_env once the sub-step's state is complete
template <class M, bool HUM>
hipError_t launch_substeps_mw_post(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                                   hipStream_t s, const LocoParams& tp);
template <class M>
constexpr bool mw_post_capable() { return mw_capable<M, PlaneGround>() && Sim<M>::NPG == 0 && !Sim<M>::COMPACT; }
// the same for a compact-store robot on limb waves (mwc_kernels.hpp; instantiated in kernels_humanoid_mwc.hip): the LAST sub-step launch of the
// step carries post_physics_step on its role waves
template <class M, bool HUM>
hipError_t launch_substeps_mwc_post(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                                    hipStream_t s, const LocoParams& tp);
template <class M>
constexpr bool mwc_post_capable() { return Sim<M>::COMPACT && M::NPG > 0 && !is_scaled<M>::value; }

template <class M, bool HUM>
hipError_t launch_loco_step(const View& v, const SimParams& P, const LocoParams& tp, const float* actions, int cfi, hipStream_t s) {
    ActParams ap;
    ap.clip = tp.clip_actions; ap.scale = tp.power_scale; ap.nact = M::ND;
    for (int d = 0; d < kMaxDof; ++d) ap.gear[d] = tp.gear[d];
    ap.mode = 0;
    if constexpr (mw_post_capable<M>()) {
        // (the fused form exists for the plain kernels only: with randomised actor parameters the two-launch form below runs)
        if (v.mw != 0 && v.fused_post != 0 && v.actor_scale == nullptr && v.limit_shift == nullptr && v.obs_noise.dist == 0 && v.act_noise.dist == 0) return launch_substeps_mw_post<M, HUM>(v, P, ap, actions, cfi * P.substeps, ACT_FROM_ACTIONS, ACT_STORED_TAU, s, tp);
    }
    if constexpr (mwc_post_capable<M>()) {
        if (v.mw != 0 && v.mw != 2 && v.fused_post != 0 && v.actor_scale == nullptr && v.limit_shift == nullptr && v.obs_noise.dist == 0 && v.act_noise.dist == 0) return launch_substeps_mwc_post<M, HUM>(v, P, ap, actions, cfi * P.substeps, ACT_FROM_ACTIONS, ACT_STORED_TAU, s, tp);
    }
    hipError_t e = launch_substeps<M>(v, P, ap, actions, cfi * P.substeps, ACT_FROM_ACTIONS, ACT_STORED_TAU, s);
    if (e != hipSuccess) return e;
    constexpr int PL = post_lanes<M>();
    if (v.obs_noise.dist != 0) hipLaunchKernelGGL((loco_post_kernel<M, HUM, true>), dim3(xcd_grid<PL>(v.N)), dim3(PL), 0, s, v, tp);
    else hipLaunchKernelGGL((loco_post_kernel<M, HUM, false>), dim3(xcd_grid<PL>(v.N)), dim3(PL), 0, s, v, tp);
    return hipGetLastError();
}
template <class M>
hipError_t launch_simulate(const View& v, const SimParams& P, hipStream_t s) {
    ActParams ap{};
    return launch_substeps<M>(v, P, ap, nullptr, P.substeps, ACT_STORED_TAU, ACT_STORED_TAU, s);
}

}  // namespace mi

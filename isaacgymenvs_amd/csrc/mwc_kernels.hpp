// mwc_kernels.hpp -- the limb-per-wave sub-step of a compact-store robot (core/engine_mwc.hpp: Humanoid) and its launcher.  Included only
// by kernels_<model>_mwc.hip.  blockDim = (64, NROLE): wave y = role y, lanes 0 .. 31 of every wave hold the same 32 envs (the upper lanes
// retire at once: barriers count waves), one workgroup per CU (the row store + exchange areas take 159 KB of its LDS).
#pragma once
#include "step_kernels.hpp"
#include "mw_kernels.hpp"          // DevBarrier
#include "core/engine_mwc.hpp"

namespace mi {

#if defined(MI_TIMING)
// debug builds only (tools/debug/mwc_phases.py): per workgroup and role 16 s_memtime stamps of the sub-step's phases
__device__ unsigned long long* g_mi_tstamp_mwc = nullptr;
#endif
template <class M>
constexpr size_t mwc_lds_bytes() { return (size_t)SimMWC<M>::MWC_SLOTS * SimMWC<M>::LANES * sizeof(float); }

template <class M, int R, bool POST = false, bool HUM = true>
__device__ __forceinline__ void mwc_role(const View& v, const SimParams& P, const ActParams& ap, const float* __restrict__ actions_in,
                                         const int src, float* lds_rows, const int e, const int lane, const LocoParams* tp = nullptr) {
    using S = SimMWC<M>;
    using MW = SimMW<M>;
    constexpr int ND = M::ND, E = S::LANES;
    const int N = v.N;
    S sim;
    load_sim(sim, v, e);
    load_actor_scales(sim, v, e);
#if defined(MI_TIMING)
    sim.tstamp = (lane == 0 && g_mi_tstamp_mwc != nullptr) ? g_mi_tstamp_mwc + ((size_t)blockIdx.x * 4 + R) * 32 : nullptr;   // tools/debug/mwc_phases.py
#endif
    float tau[M::NDA];
    if (src != ACT_STORED_TAU) {
        sfor<ND>([&](auto K) MI_LAMBDA {
            constexpr int k = K;
            constexpr bool mine = MW::template owns_gi<R>(M::OFF + k);
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
                t = a * ap.gear[k] * ap.scale;          // effort mode (humanoid.py:281-285)
            }
            tau[k] = t;
            if constexpr (mine) v.tau[k * N + e] = t;
        });
    } else {
        sfor<ND>([&](auto K) MI_LAMBDA { tau[K] = v.tau[K * N + e]; });
    }
    const float h = P.dt / (float)P.substeps;
    const Strided lamc{v.lamc + e, N}, laml{v.laml + e, N}, sensor{v.sensor + e, N}, dof_force{v.dof_force + e, N};
    const float mu_env = (v.friction != nullptr) ? v.friction[e] : -1.f;
    const SelfCol selfcol{Strided{v.lamp ? v.lamp + e : nullptr, N}, Strided{v.pairf ? v.pairf + e : nullptr, N}, v.dropped ? v.dropped + e : nullptr, N};
    const SelfCol* scp = (M::NPG > 0 && v.lamp != nullptr) ? &selfcol : nullptr;    // uniform
    if constexpr (S::HAS_PAIR_ROLE && R == S::PAIR_ROLE) {
        sim.substep_pair(P, h, RowStore<E>{lds_rows + lane}, scp, DevBarrier{});       // owns no dof: nothing to write back
    } else {
        sim.template substep_role_c<R>(P, tau, h, RowStore<E>{lds_rows + lane}, lamc, laml, sensor, dof_force, mu_env, scp, DevBarrier{});
    }
    if constexpr (POST) {
        // post_physics_step on the role waves of the step's last sub-step launch (mw_kernels.hpp loco_post_role): the f-terms cross through the
        // self-contact rows' region, dead once every role has left its output phase (hence the barrier first)
        static_assert(3 * M::ND <= S::KPAIR * S::P_CSZ, "the reward terms fit the self-contact rows' region");
        // (this launch spills 72 registers where the plain sub-step spills 8 -- a never-taken block between the two parts, Sim::alloc_fence, does not
        //  change that -- and still wins: the post kernel it replaces cost more)
        loco_post_role<S, M, HUM, E, R, true>(v, *tp, sim, nullptr, e, lds_rows + (size_t)S::P_B * E + lane);
        return;
    }
    sfor<ND>([&](auto K) MI_LAMBDA {
        if constexpr (MW::template owns_gi<R>(M::OFF + K)) {
            v.dof[K * N + e] = sim.q[K];
            v.dof[(ND + K) * N + e] = sim.qd[K];
        }
    });
    if constexpr (R == M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = sim.root[K]; });
}

// The kernel's arguments as ONE struct, read through the kernarg segment pointer where they are used instead of being taken as by-value
// parameters: ~40 pointers of the View + the action / sim parameters would otherwise be preloaded into SGPRs and stay live through the
// whole role body (the four role bodies together spilled 188 SGPRs to VGPR lanes); loads from the kernarg segment are invariant, so the
// compiler re-issues them (scalar cache hits) instead of keeping the values.
struct MwcArgs {
    View v;
    SimParams P;
    ActParams ap;
    const float* actions_in;
    int src;
};
template <class M>
__global__ __launch_bounds__(64 * M::NROLE) void substep_mwc_kernel(MwcArgs args_by_value) {
    extern __shared__ float lds_rows[];   // [MWC_SLOTS][32]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    const MwcArgs& a = *reinterpret_cast<const MwcArgs*>(__builtin_amdgcn_kernarg_segment_ptr());
#else
    const MwcArgs& a = args_by_value;       // (host pass of the compiler: never executed)
#endif
    constexpr int E = SimMWC<M>::LANES;
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.v.N) return;                 // all four waves hold the same envs and agree on this
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
#if defined(MI_MWC_ONLY_ROLE)     // tools/debug only: resource usage of one role's instruction stream
    if (role == MI_MWC_ONLY_ROLE) mwc_role<M, MI_MWC_ONLY_ROLE>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane);
#else
    switch (role) {
        case 0: mwc_role<M, 0>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane); break;
        case 1: mwc_role<M, 1>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane); break;
        case 2: mwc_role<M, 2>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane); break;
        default: mwc_role<M, 3>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane); break;
    }
#endif
}

// The step's LAST sub-step with post_physics_step on its role waves (option "fused_post"; mw_kernels.hpp loco_post_role): the leg roles and the
// trunk + arms role reset / observe / score their own dofs, the trunk role -- the one with slack in this kernel -- does the root part and the reward.
struct MwcPostArgs {
    MwcArgs a;
    LocoParams tp;
};
template <class M, bool HUM>
__global__ __launch_bounds__(64 * M::NROLE) void substep_mwc_post_kernel(MwcPostArgs args_by_value) {
    extern __shared__ float lds_rows[];   // [MWC_SLOTS][32]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    const MwcPostArgs& pa = *reinterpret_cast<const MwcPostArgs*>(__builtin_amdgcn_kernarg_segment_ptr());
    const MwcArgs& a = pa.a;
    constexpr int E = SimMWC<M>::LANES;
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.v.N) return;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
    switch (role) {
        case 0: mwc_role<M, 0, true, HUM>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane, &pa.tp); break;
        case 1: mwc_role<M, 1, true, HUM>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane, &pa.tp); break;
        case 2: mwc_role<M, 2, true, HUM>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane, &pa.tp); break;
        default: mwc_role<M, 3, true, HUM>(a.v, a.P, a.ap, a.actions_in, a.src, lds_rows, e, lane, &pa.tp); break;
    }
#endif
}
template <class M, bool HUM>
hipError_t launch_substeps_mwc_post(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                                    hipStream_t s, const LocoParams& tp) {
    static unsigned long long configured = 0ull, pconfigured = 0ull;
    constexpr size_t lds = mwc_lds_bytes<M>();
    constexpr int E = SimMWC<M>::LANES;
    const dim3 grid(xcd_grid<E>(v.N)), block(64, M::NROLE);
    auto kern = substep_mwc_kernel<M>;
    auto pkern = substep_mwc_post_kernel<M, HUM>;
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds, &configured); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)pkern, lds, &pconfigured); e != hipSuccess) return e;
    for (int i = 0; i + 1 < n_sub; ++i) hipLaunchKernelGGL(kern, grid, block, lds, s, MwcArgs{v, P, ap, actions, i == 0 ? first : rest});
    hipLaunchKernelGGL(pkern, grid, block, lds, s, MwcPostArgs{MwcArgs{v, P, ap, actions, n_sub == 1 ? first : rest}, tp});
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ all sub-steps of a step in ONE launch
// NOT PART OF THE PRODUCT BUILD (-DMI_MWC_FUSED builds it): measured and rejected in round 4, profiles/r4_humanoid_fused_substeps.txt.  The
// arithmetic is right (SimMWC::substeps_fused is bit-identical to one call per sub-step on the host, tests/test_self_collision.py), but every
// form of it -- a run-time loop, the same loop with the global pointers / strides / step size / parameter block made opaque per iteration,
// two straight-line copies -- lands in the heavy-spill regime native.build() refuses: 370-440 spilled SGPRs and 190-310 spilled VGPRs against
// 84 / 50 of the one-sub-step kernel (the trunk + arms role alone: 0 -> 80-155 spilled VGPRs).  The role bodies already use all 512
// registers; whatever is common to two sub-steps (row offsets k * N of the SoA tensors, the model's constant tables, expressions of the step
// size) is kept across them instead of being re-derived.
// Option "fused_sub" (round 4): SimMWC::substeps_fused loops over the control step's sub-steps inside the launch.  The efforts are derived
// once (effort mode: they do not depend on the state), every role keeps its state in registers between sub-steps, the pair role is handed
// the new pose through a dead half of the trunk exchange area (core/engine_mwc.hpp).  One launch instead of `n_sub`: the state is read and
// written once per control step, and the per-launch costs (dispatch, kernarg + state loads at HBM latency, store drain) are paid once.
struct MwcFusedArgs {
    View v;
    SimParams P;
    ActParams ap;
    const float* actions_in;
    int src, n_sub;
};
template <class M, int R>
__device__ __forceinline__ void mwc_role_fused(const MwcFusedArgs& a, float* lds_rows, const int e, const int lane) {
    using S = SimMWC<M>;
    using MW = SimMW<M>;
    constexpr int ND = M::ND, E = S::LANES;
    const View& v = a.v;
    const ActParams& ap = a.ap;
    const int N = v.N;
    S sim;
    load_sim(sim, v, e);
    load_actor_scales(sim, v, e);
    float tau[M::NDA];
    if (a.src != ACT_STORED_TAU) {
        sfor<ND>([&](auto K) MI_LAMBDA {
            constexpr int k = K;
            constexpr bool mine = MW::template owns_gi<R>(M::OFF + k);
            float t = 0.f;
            if (k < ap.nact) {
                float x;
                if (a.src == ACT_FROM_ACTIONS) {
                    x = a.actions_in[(size_t)e * ap.nact + k];
                    x = fminf(fmaxf(x, -ap.clip), ap.clip);
                    if constexpr (mine) v.actions[k * N + e] = x;
                } else {
                    x = v.actions[k * N + e];
                }
                t = x * ap.gear[k] * ap.scale;          // effort mode (humanoid.py:281-285)
            }
            tau[k] = t;
            if constexpr (mine) v.tau[k * N + e] = t;
        });
    } else {
        sfor<ND>([&](auto K) MI_LAMBDA { tau[K] = v.tau[K * N + e]; });
    }
    const float h = a.P.dt / (float)a.P.substeps;
    const Strided lamc{v.lamc + e, N}, laml{v.laml + e, N}, sensor{v.sensor + e, N}, dof_force{v.dof_force + e, N};
    const float mu_env = (v.friction != nullptr) ? v.friction[e] : -1.f;
    const SelfCol selfcol{Strided{v.lamp ? v.lamp + e : nullptr, N}, Strided{v.pairf ? v.pairf + e : nullptr, N}, v.dropped ? v.dropped + e : nullptr, N};
    const SelfCol* scp = (M::NPG > 0 && v.lamp != nullptr) ? &selfcol : nullptr;    // uniform
#ifndef MI_MWC_FUSED_NSUB
#define MI_MWC_FUSED_NSUB 2
#endif
    sim.template substeps_fused<R, MI_MWC_FUSED_NSUB>(a.P, tau, h, RowStore<E>{lds_rows + lane}, lamc, laml, sensor, dof_force, mu_env, scp, DevBarrier{}, a.n_sub);
    sfor<ND>([&](auto K) MI_LAMBDA {
        if constexpr (MW::template owns_gi<R>(M::OFF + K)) {
            v.dof[K * N + e] = sim.q[K];
            v.dof[(ND + K) * N + e] = sim.qd[K];
        }
    });
    if constexpr (R == M::TRUNK_ROLE) sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = sim.root[K]; });
}
#if defined(MI_MWC_FUSED)
template <class M>
__global__ __launch_bounds__(64 * M::NROLE) void substep_mwc_fused_kernel(MwcFusedArgs args_by_value) {
    extern __shared__ float lds_rows[];   // [MWC_SLOTS][32]
    static_assert(M::NROLE == 4, "four roles, one per SIMD of a CU");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    const MwcFusedArgs& a = *reinterpret_cast<const MwcFusedArgs*>(__builtin_amdgcn_kernarg_segment_ptr());
    constexpr int E = SimMWC<M>::LANES;
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.v.N) return;                 // all four waves hold the same envs and agree on this
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
#if defined(MI_MWC_ONLY_ROLE)     // tools/debug only: resource usage of one role's instruction stream
    if (role == MI_MWC_ONLY_ROLE) mwc_role_fused<M, MI_MWC_ONLY_ROLE>(a, lds_rows, e, lane);
#else
    switch (role) {
        case 0: mwc_role_fused<M, 0>(a, lds_rows, e, lane); break;
        case 1: mwc_role_fused<M, 1>(a, lds_rows, e, lane); break;
        case 2: mwc_role_fused<M, 2>(a, lds_rows, e, lane); break;
        default: mwc_role_fused<M, 3>(a, lds_rows, e, lane); break;
    }
#endif
#endif
}

#endif   // MI_MWC_FUSED

template <class M>
hipError_t launch_substeps_mwc(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                               hipStream_t s) {
    static unsigned long long configured = 0ull, fconfigured = 0ull;
    constexpr size_t lds = mwc_lds_bytes<M>();
    constexpr int E = SimMWC<M>::LANES;
    const dim3 grid(xcd_grid<E>(v.N)), block(64, M::NROLE);
#if defined(MI_MWC_FUSED)
    // the fused form keeps the first sub-step's efforts: that is what `rest` = ACT_STORED_TAU means for an effort-mode robot
    if (v.fused_sub != 0 && n_sub > 1 && rest == ACT_STORED_TAU) {
        auto fkern = substep_mwc_fused_kernel<M>;
        if (hipError_t e = ensure_dynamic_lds((const void*)fkern, lds, &fconfigured); e != hipSuccess) return e;
        hipLaunchKernelGGL(fkern, grid, block, lds, s, MwcFusedArgs{v, P, ap, actions, first, n_sub});
        return hipGetLastError();
    }
#else
    (void)fconfigured;
#endif
    auto kern = substep_mwc_kernel<M>;
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds, &configured); e != hipSuccess) return e;
    for (int i = 0; i < n_sub; ++i) hipLaunchKernelGGL(kern, grid, block, lds, s, MwcArgs{v, P, ap, actions, i == 0 ? first : rest});
    return hipGetLastError();
}

}  // namespace mi

// mw_kernels.hpp -- the multi-wave physics sub-step kernel (core/engine_mw.hpp) and its launcher.  Included only by the
// kernels_mw_<model>.hip translation units, which instantiate launch_substeps_mw for one model / ground pair each.
#pragma once
#include "step_kernels.hpp"
#include "core/engine_mw.hpp"

namespace mi {

// ------------------------------------------------------------------------------------------------ multi-wave sub-step
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
                    if (v.act_noise.dist != 0) a = apply_noise(v.act_noise, v.seed, (uint32_t)(v.env_offset + e), v.step, 1u, (uint32_t)k, a);
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

template <class M, class GND>
hipError_t launch_substeps_mw(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                              hipStream_t s, const GND& gnd) {
    static unsigned long long conf16 = 0ull, conf32 = 0ull;
    constexpr size_t lds16 = mw_lds_bytes<M, 16>(), lds32 = mw_lds_bytes<M, 32>();
    const dim3 block(64, M::NROLE);
    if (MI_MW_HAS16 && v.mw == 16) {
        auto kern = substep_mw_kernel<M, GND, MI_MW_HAS16 ? 16 : 32>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds16, &conf16); e != hipSuccess) return e;
        const dim3 grid(xcd_grid<16>(v.N));
        for (int i = 0; i < n_sub; ++i) hipLaunchKernelGGL(kern, grid, block, lds16, s, v, P, ap, actions, i == 0 ? first : rest, gnd);
    } else {
        auto kern = substep_mw_kernel<M, GND, 32>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds32, &conf32); e != hipSuccess) return e;
        const dim3 grid(xcd_grid<32>(v.N));
        for (int i = 0; i < n_sub; ++i) hipLaunchKernelGGL(kern, grid, block, lds32, s, v, P, ap, actions, i == 0 ? first : rest, gnd);
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
    loco_post_env<M, HUM, E, true>(a.v, a.tp, e, true, root, q, qd);
#endif
}

template <class M, bool HUM>
hipError_t launch_substeps_mw_post(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                                   hipStream_t s, const LocoParams& tp) {
    if (n_sub > 1) {
        if (hipError_t e = launch_substeps_mw<M, PlaneGround>(v, P, ap, actions, n_sub - 1, first, rest, s, PlaneGround{}); e != hipSuccess) return e;
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

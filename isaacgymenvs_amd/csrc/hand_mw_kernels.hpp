// hand_mw_kernels.hpp -- the finger-per-wave Shadow-Hand sub-step (core/hand_engine_mw.hpp) and its launcher, a template over the object's
// shape.  Included by kernels_shadow_hand_mw*.hip (one shape per translation unit: they compile in parallel).  blockDim = (64, NROLE):
// wave y = role y, lanes 0 .. E-1 of every wave hold the same E envs.  E = 32 (option multi_wave 32): the upper lanes retire at once
// (barriers count waves), two workgroups per CU (<= 80 KB of LDS each), two half-filled waves on every SIMD once 8192 envs are
// exceeded.  E = 64 (multi_wave 64): full waves, one workgroup per CU (160 KB of LDS), one wave per SIMD with 512 registers -- the
// sub-step is bound by the SIMDs' VALU issue (4 cycles per wave instruction whether 32 or 64 lanes are live), so at 16384 envs this
// form executes half the wave instructions per env.
#pragma once
#include "hand_kernels.hpp"
#include "mw_kernels.hpp"          // DevBarrier
#include "core/hand_engine_mw.hpp"

namespace mi {

// (a template over the hand task since the end of round 3: ShadowHandTask and AllegroHandTask -- the Allegro hand has no wrist dofs and no
//  tendons: its fingers couple through the object alone)
template <class HT, int E>
constexpr size_t hand_mw_lds_bytes() { return (size_t)HandSimMW<typename HT::M>::MW_SLOTS * E * sizeof(float); }

#if defined(MI_TIMING)
__device__ unsigned long long* g_mi_tstamp_hmw = nullptr;     // debug builds: per workgroup and role 16 s_memtime stamps
#endif

template <class HT, int SHAPE, int E, int R, bool PSENS = true>
__device__ __forceinline__ void hand_mw_role(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, float* lds_rows,
                                             const int e, const int lane) {
    using HM = typename HT::M;
    using HSW = HandSimMW<HM>;
    using MW = SimMW<HM>;
    constexpr int ND = HT::ND;
    const int N = v.N;
    HSW sim;
    sfor<3>([&](auto K) MI_LAMBDA { sim.root[K] = p.hand_pos[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { sim.root[3 + K] = p.hand_quat[K]; });
    sfor<6>([&](auto K) MI_LAMBDA { sim.root[7 + K] = 0.f; });
    float target[ND];
    sfor<ND>([&](auto K) MI_LAMBDA {            // the wrist and the own fingers
        if constexpr (MW::template sees_gi<R>(K)) {
            sim.q[K] = v.dof[K * N + e];
            sim.qd[K] = v.dof[(ND + K) * N + e];
            target[K] = hv.cur_targets[K * N + e];
        } else {
            sim.q[K] = 0.f; sim.qd[K] = 0.f; target[K] = 0.f;
        }
    });
    sfor<3>([&](auto K) MI_LAMBDA { sim.obj.pos[K] = hv.object_state[K * N + e]; sim.obj.vel[K] = hv.object_state[(7 + K) * N + e];
                                    sim.obj.angvel[K] = hv.object_state[(10 + K) * N + e]; });
    sfor<4>([&](auto K) MI_LAMBDA { sim.obj.quat[K] = hv.object_state[(3 + K) * N + e]; });
    ObjectParams OP{p.cube_half, p.cube_mass, p.cube_inertia, p.mu,
                    {hv.obj_force[e], hv.obj_force[N + e], hv.obj_force[2 * N + e]}};
    if constexpr (SHAPE != OBJ_BOX) sfor<3>([&](auto K) MI_LAMBDA { OP.dims[K] = p.object_dims[K]; OP.inertia3[K] = p.object_inertia[K]; });
    const float mu_e = hv.mu_env[e];
    if (mu_e >= 0.f) OP.mu = mu_e;
    OP.randomise(hv.scale[HS_OBJECT_MASS * N + e], hv.scale[HS_OBJECT_SCALE * N + e]);
    sim.actor_scale = Strided{hv.scale + e, N};
    sim.limit_shift = Strided{hv.limit_shift + e, N};
    sim.drive_clamp = hv.drive_clamp;
    sim.pair_k = hv.pair_k;
    sim.pair_sens = hv.pair_sens;
    if constexpr (is_scaled<typename HT::M>::value) { if (hv.body_mass != nullptr) sim.body_mass = Strided{hv.body_mass + e, N}; }
#if defined(MI_TIMING)
    sim.tstamp = (lane == 0 && g_mi_tstamp_hmw != nullptr) ? g_mi_tstamp_hmw + ((size_t)blockIdx.x * 4 + R) * 16 : nullptr;
#endif
    const float h = P.dt / (float)P.substeps;
    int nc = 0;
    sim.template substep_hand_role<R, E, SHAPE, PSENS>(P, OP, target, h, RowStore<E>{lds_rows + lane}, Strided{v.laml + e, N},
                                                         Strided{v.sensor + e, N}, Strided{v.dof_force + e, N}, &nc, DevBarrier{});
    sfor<ND>([&](auto K) MI_LAMBDA {
        if constexpr (MW::template owns_gi<R>(K)) { v.dof[K * N + e] = sim.q[K]; v.dof[(ND + K) * N + e] = sim.qd[K]; }
    });
    hv.npair[R * N + e] = sim.pair_active;
    if constexpr (R == HM::TRUNK_ROLE) {
        sfor<3>([&](auto K) MI_LAMBDA { hv.object_state[K * N + e] = sim.obj.pos[K]; hv.object_state[(7 + K) * N + e] = sim.obj.vel[K];
                                        hv.object_state[(10 + K) * N + e] = sim.obj.angvel[K]; });
        sfor<4>([&](auto K) MI_LAMBDA { hv.object_state[(3 + K) * N + e] = sim.obj.quat[K]; });
        hv.ncontact[e] = nc & 0xFFFF;
        if (nc >> 16) hv.ndropped[e] += nc >> 16;
    }
}

// the kernel's arguments as ONE struct read through the kernarg segment pointer (see mwc_kernels.hpp MwcArgs: by-value parameters would
// be preloaded into SGPRs and stay live through the role bodies)
struct HandMwArgs {
    View v;
    HandView hv;
    SimParams P;
    HandParams p;
};
template <class HT, int SHAPE, int E, bool PSENS = true>
__device__ __forceinline__ void hand_mw_body(float* lds_rows) {
    static_assert(HT::M::NROLE == 4, "four roles, one per SIMD of a CU");
#if defined(__HIP_DEVICE_COMPILE__)
    const HandMwArgs& a = *reinterpret_cast<const HandMwArgs*>(__builtin_amdgcn_kernarg_segment_ptr());
    const int lane = threadIdx.x;
    if (lane >= E) return;
    const int e = xcd_env_base<E>(blockIdx.x) + lane;
    if (e >= a.v.N) return;                 // all four waves hold the same envs and agree on this
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.y);
#if defined(MI_HMW_ONLY_ROLE)     // tools/debug only: resource usage of one role's instruction stream
    if (role == MI_HMW_ONLY_ROLE) hand_mw_role<HT, SHAPE, E, MI_HMW_ONLY_ROLE, PSENS>(a.v, a.hv, a.P, a.p, lds_rows, e, lane);
#else
    switch (role) {
        case 0: hand_mw_role<HT, SHAPE, E, 0, PSENS>(a.v, a.hv, a.P, a.p, lds_rows, e, lane); break;
        case 1: hand_mw_role<HT, SHAPE, E, 1, PSENS>(a.v, a.hv, a.P, a.p, lds_rows, e, lane); break;
        case 2: hand_mw_role<HT, SHAPE, E, 2, PSENS>(a.v, a.hv, a.P, a.p, lds_rows, e, lane); break;
        default: hand_mw_role<HT, SHAPE, E, 3, PSENS>(a.v, a.hv, a.P, a.p, lds_rows, e, lane); break;
    }
#endif
#else
    (void)lds_rows;
#endif
}
// (the kernel arguments are read through the kernarg segment pointer inside hand_mw_body; the host pass of the compiler never executes it)
// PSENS: with the pairs' forces on the fingertip sensors (the last sub-step launch of a call; core/hand_engine_mw.hpp substep_hand_role)
template <class HT, int SHAPE, bool PSENS = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void hand_substep_mw_kernel(HandMwArgs args_by_value) {
    extern __shared__ float lds_rows[];   // [MW_SLOTS][32]
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    hand_mw_body<HT, SHAPE, 32, PSENS>(lds_rows);
#endif
}
template <class HT, int SHAPE, bool PSENS = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void hand_substep_mw64_kernel(HandMwArgs args_by_value) {
    extern __shared__ float lds_rows[];   // [MW_SLOTS][64]
#if defined(__HIP_DEVICE_COMPILE__)
    (void)args_by_value;
    hand_mw_body<HT, SHAPE, 64, PSENS>(lds_rows);
#endif
}

template <class HT, int SHAPE>
inline hipError_t hand_substeps_mw_shape(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    static unsigned long long conf32 = 0ull, conf64 = 0ull;
    const dim3 block(64, HT::M::NROLE);
    // the sensor values of every launch but the last are overwritten unseen: only the last one carries the code that adds the pairs' forces to the
    // fingertip sensors (a model without pairs has one instantiation)
    constexpr bool SPLIT = HT::M::NHP > 0;
    static unsigned long long conf32n = 0ull, conf64n = 0ull;
    if (v.mw == 64) {
        constexpr size_t lds = hand_mw_lds_bytes<HT, 64>();
        static_assert(lds <= 160 * 1024, "one 64-env hand workgroup per CU");
        auto kern = hand_substep_mw64_kernel<HT, SHAPE, true>;
        auto kern_n = hand_substep_mw64_kernel<HT, SHAPE, !SPLIT>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds, &conf64); e != hipSuccess) return e;
        if constexpr (SPLIT) { if (hipError_t e = ensure_dynamic_lds((const void*)kern_n, lds, &conf64n); e != hipSuccess) return e; }
        const dim3 grid(xcd_grid<64>(v.N));
        for (int i = 0; i < n; ++i) { HandView hl = hv; hl.pair_sens = (i == n - 1) ? 1 : 0; hipLaunchKernelGGL((i == n - 1) ? kern : kern_n, grid, block, lds, s, HandMwArgs{v, hl, P, p}); }
    } else {
        constexpr size_t lds = hand_mw_lds_bytes<HT, 32>();
        auto kern = hand_substep_mw_kernel<HT, SHAPE, true>;
        auto kern_n = hand_substep_mw_kernel<HT, SHAPE, !SPLIT>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, lds, &conf32); e != hipSuccess) return e;
        if constexpr (SPLIT) { if (hipError_t e = ensure_dynamic_lds((const void*)kern_n, lds, &conf32n); e != hipSuccess) return e; }
        const dim3 grid(xcd_grid<32>(v.N));
        for (int i = 0; i < n; ++i) { HandView hl = hv; hl.pair_sens = (i == n - 1) ? 1 : 0; hipLaunchKernelGGL((i == n - 1) ? kern : kern_n, grid, block, lds, s, HandMwArgs{v, hl, P, p}); }
    }
    return hipGetLastError();
}

}  // namespace mi

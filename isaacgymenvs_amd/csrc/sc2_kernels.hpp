// sc2_kernels.hpp -- the sub-step of a self-colliding robot (Humanoid) on TWO waves per workgroup (gfx950).
//
// One workgroup = the same 32 envs as in the one-wave form, blockDim (64, 2): wave 0 is the main wave (Sim::substep role 0:
// everything except the self-collision phase), wave 1 the helper (role 1: tree pass, factorisation, then broad + narrow phase
// and the rows of the self contacts, written straight into the self-contact slots of the shared row store).  The two meet at
// ONE s_barrier, after the main wave has built its limit and ground-contact rows; the helper hands its bookkeeping (group ->
// slot map, contact point / normal / bodies / friction per slot: 25 dwords) over through the store and retires.  Both waves run
// the same code on the same inputs for the shared part, so L, S and the sphere centres agree bit for bit -- the result of the
// split is bit-identical to the one-wave sub-step (tests/test_self_collision.py on the host build, tests/test_gpu_parity.py).
// What it buys: the self-collision phase was 19.5 % of the one-wave sub-step; it now runs beside the ground rows on another SIMD.
#pragma once
#include "step_kernels.hpp"
#include "mw_kernels.hpp"   // DevBarrier

namespace mi {

// NR = 2: main + self-collision helper; NR = 3: + a third wave that builds the joint-limit rows
template <class M, int NR>
__global__ __launch_bounds__(64 * NR) void substep_sc2_kernel(View v, SimParams P, ActParams ap, const float* __restrict__ actions_in, int src) {
    extern __shared__ float lds_rows[];  // [ROW_SLOTS][LANES], shared by the two waves
    using S = Sim<M>;
    constexpr int LANES = S::LANES;
    static_assert(S::COMPACT && S::NPG > 0 && LANES <= 64 && rows_fit_lds<M>() && S::STAGES_LAM, "two-wave form: compact store with self-contact slots");
    const int lane = threadIdx.x, role = threadIdx.y;
    const int e = xcd_env_base<LANES>(blockIdx.x) + lane;
    const int N = v.N;
    if (lane >= LANES || e >= N) return;     // both waves of the workgroup hold the same envs, so they agree on who retires
    S sim;
    load_sim(sim, v, e);
    load_actor_scales(sim, v, e);
    float tau[M::NDA];
    if (role == 0) {
        efforts_for_substep<M>(v, ap, actions_in, src, e, sim, tau);
        prestage_warm_start<M, NR == 2>(v, e, lds_rows);       // with three waves the limit-row helper stages its own impulses
    } else {
        sfor<M::ND>([&](auto K) MI_LAMBDA { tau[K] = 0.f; });   // the helper never looks at the right-hand side
    }
    const float h = P.dt / (float)P.substeps;
    const Strided lamc{v.lamc + e, N}, laml{v.laml + e, N}, sensor{v.sensor + e, N}, dof_force{v.dof_force + e, N};
    const float mu_env = (v.friction != nullptr) ? v.friction[e] : -1.f;
    const SelfCol selfcol{Strided{v.lamp + e, N}, Strided{v.pairf ? v.pairf + e : nullptr, N}, v.dropped ? v.dropped + e : nullptr, N};
    sim.substep(P, tau, h, RowStore<LANES>{lds_rows + lane}, lamc, laml, sensor, dof_force, PlaneGround{}, mu_env, Strided{nullptr, N}, nullptr,
                role != 2, &selfcol, role, DevBarrier{}, NR);
    if (role == 0) store_sim(sim, v, e);
}

template <class M>
hipError_t launch_substeps_sc2(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first, int rest,
                               hipStream_t s) {
    constexpr size_t lds = lds_bytes<M>();
    constexpr int LANES = Sim<M>::LANES;
    // NR = 3 (a third wave for the joint-limit rows, Sim::substep role 2) was measured and is not instantiated: Humanoid@8192 0.444 ms
    // per step against 0.412 ms with two waves (one wave: 0.521 ms) -- three waves of this kernel on one CU cost each other more than the
    // 16 us of limit rows they take off the main wave.  The role logic stays (host-tested for 2 and 3 threads, tests/test_self_collision.py).
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)substep_sc2_kernel<M, 2>, lds, &configured); e != hipSuccess) return e;
    for (int i = 0; i < n_sub; ++i)
        hipLaunchKernelGGL((substep_sc2_kernel<M, 2>), dim3(xcd_grid<LANES>(v.N)), dim3(64, 2), lds, s, v, P, ap, actions, i == 0 ? first : rest);
    return hipGetLastError();
}

}  // namespace mi

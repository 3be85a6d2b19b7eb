// kernels_body_states.hip -- gym.refresh_rigid_body_state_tensor for every model: ONE THREAD PER (env, body), blockIdx.y = the body, so a
// wave walks one kinematic chain (Sim<M>::body_state) and N * NB / 64 waves fill the chip.  Off the step path: launched only by
// mi_engine_refresh_rigid_body_states (reference shadow_hand.py:150-175,440; the tasks of the benchmark read fused observation tensors).
#include "step_kernels.hpp"
#include "gen/model_ant.h"
#include "gen/model_cartpole.h"
#include "gen/model_humanoid.h"
#include "gen/model_anymal.h"
#include "gen/model_shadow_hand.h"
#include "gen/model_allegro_hand.h"
#include "gen/model_quadcopter.h"
#include "gen/model_ingenuity.h"
#include "gen/model_balance_bot.h"

namespace mi {

template <class M>
__global__ __launch_bounds__(64) void body_states_kernel(View v) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    const int body = blockIdx.y;
    sfor<M::NB>([&](auto B_) MI_LAMBDA {
        constexpr int b = B_;
        if (body == b) {         // wave-uniform
            Sim<M> sim;
            sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
            sfor<M::ND>([&](auto D) MI_LAMBDA {
                if constexpr (Sim<M>::on_chain(M::dof_body[D], b)) { sim.q[D] = v.dof[D * N + e]; sim.qd[D] = v.dof[(M::ND + D) * N + e]; }
            });
            float o[13];
            sim.template body_state<b>(o);
            sfor<13>([&](auto K) MI_LAMBDA { v.body_state[(b * 13 + K) * N + e] = o[K]; });
        }
    });
}
template <class M>
static hipError_t launch_bs(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(body_states_kernel<M>, dim3((v.N + 63) / 64, M::NB), dim3(64), 0, s, v);
    return hipGetLastError();
}
// gym.refresh_jacobian_tensors / refresh_mass_matrix_tensors (reference franka_cube_stack.py:388-392,551-552) into CALLER tensors, row-major
// like the simulator's: out_j [N][NB][6][NV], out_h [N][NV][NV] (Sim<M>::body_jacobian / mass_matrix).  Off the step path.
// One kernel per BODY (a template parameter, not blockIdx.y): with every body's unrolled chain in one kernel the constants of all of them
// competed for the scalar registers (530 spilled SGPRs for the Ant, 1510 for the Allegro hand -- the regime DESIGN.md warns about).
template <class M, int b>
__global__ __launch_bounds__(64) void jacobians_kernel(View v, float* __restrict__ out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    constexpr int NV = M::NV;
    Sim<M> sim;
    sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
    sfor<M::ND>([&](auto D) MI_LAMBDA {
        sim.qd[D] = 0.f;
        if constexpr (Sim<M>::on_chain(M::dof_body[D], b)) sim.q[D] = v.dof[D * N + e]; else sim.q[D] = 0.f;
    });
    float J[6 * NV];
    sim.template body_jacobian<b>(J);
    float* o = out + ((size_t)e * M::NB + b) * 6 * NV;
    sfor<6 * NV>([&](auto K) MI_LAMBDA { o[K] = J[K]; });
}
template <class M>
__global__ __launch_bounds__(64) void mass_matrix_kernel(View v, SimParams P, float* __restrict__ out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    constexpr int NV = M::NV;
    Sim<M> sim;
    sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
    sfor<M::ND>([&](auto D) MI_LAMBDA { sim.q[D] = v.dof[D * N + e]; sim.qd[D] = 0.f; });
    float H[NV * NV];
    sim.mass_matrix(P, H);
    float* o = out + (size_t)e * NV * NV;
    for (int k = 0; k < NV * NV; ++k) o[k] = H[k];
}
template <class M>
static hipError_t launch_kin(const View& v, const SimParams& P, float* out_j, float* out_h, hipStream_t s) {
    if (out_j) sfor<M::NB>([&](auto B_) { hipLaunchKernelGGL((jacobians_kernel<M, decltype(B_)::value>), dim3((v.N + 63) / 64), dim3(64), 0, s, v, out_j); });
    if (out_h) hipLaunchKernelGGL(mass_matrix_kernel<M>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, P, out_h);
    return hipGetLastError();
}
hipError_t launch_kinematics_views(int task, const View& v, const SimParams& P, float* out_j, float* out_h, hipStream_t s) {
    switch (task) {
        case 0: return launch_kin<ModelCartpole>(v, P, out_j, out_h, s);
        case 1: return launch_kin<ModelAnt>(v, P, out_j, out_h, s);
        case 2: return launch_kin<ModelHumanoid>(v, P, out_j, out_h, s);
        case 3: case 5: return launch_kin<ModelAnymal>(v, P, out_j, out_h, s);
        case 4: return launch_kin<ModelShadowHand>(v, P, out_j, out_h, s);
        case 6: return launch_kin<ModelQuadcopter>(v, P, out_j, out_h, s);
        case 7: return launch_kin<ModelIngenuity>(v, P, out_j, out_h, s);
        case 9: return launch_kin<ModelAllegroHand>(v, P, out_j, out_h, s);
        default: return launch_kin<ModelBalanceBot>(v, P, out_j, out_h, s);
    }
}

// task ids as in arena_layout.hpp
hipError_t launch_body_states(int task, const View& v, hipStream_t s) {
    switch (task) {
        case 0: return launch_bs<ModelCartpole>(v, s);
        case 1: return launch_bs<ModelAnt>(v, s);
        case 2: return launch_bs<ModelHumanoid>(v, s);
        case 3: case 5: return launch_bs<ModelAnymal>(v, s);
        case 4: return launch_bs<ModelShadowHand>(v, s);
        case 6: return launch_bs<ModelQuadcopter>(v, s);
        case 7: return launch_bs<ModelIngenuity>(v, s);
        case 9: return launch_bs<ModelAllegroHand>(v, s);
        default: return launch_bs<ModelBalanceBot>(v, s);
    }
}

}  // namespace mi

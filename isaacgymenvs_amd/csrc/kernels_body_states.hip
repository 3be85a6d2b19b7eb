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

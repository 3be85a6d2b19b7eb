// kernels_ball_balance.hip -- BallBalance (reference isaacgymenvs/tasks/ball_balance.py): a tray on three position-driven legs
// keeps a dropped ball in place.  pre kernel = pre_physics_step (deferred resets, position targets), sub-step kernel =
// gym.simulate on core/bbot_engine.hpp (attractor-pinned feet, ball <-> tray contact), post kernel = post_physics_step.
// One env per lane, 64 envs per wave; the sub-step keeps its 18 constraint rows in registers (no LDS).
#include "step_kernels.hpp"
#include "task_views.hpp"
#include "core/bbot_engine.hpp"
#include "gen/model_balance_bot.h"
#include "tasks/ball_balance.hpp"

namespace mi {

using BM = ModelBalanceBot;
static_assert(BM::ND == kBbotDof && BM::NSENS == kBbotSensors && BM::NB == 7, "balance bot model");
static_assert(sizeof(BbotPhys) == sizeof(BallBalanceParams) - offsetof(BallBalanceParams, pin_stiffness), "BbotPhys is the tail of BallBalanceParams");


static __device__ __forceinline__ const BbotPhys& phys_of(const BallBalanceParams& p) { return *reinterpret_cast<const BbotPhys*>(&p.pin_stiffness); }

static __device__ __forceinline__ void bbot_reset_env(const View& v, const BbotView& bv, const BallBalanceParams& p, int e) {
    const int N = v.N;
    const int ep = v.episode[e];
    float ball[13];
    bbot_reset_ball(p, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, ball);
    for (int k = 0; k < 13; ++k) {
        v.root[k * N + e] = v.init_root[k * N + e];                              // :353
        bv.ball[k * N + e] = ball[k];
    }
    for (int d = 0; d < kBbotDof; ++d) { v.dof[d * N + e] = 0.f; v.dof[(kBbotDof + d) * N + e] = 0.f; v.laml[d * N + e] = 0.f; }   // initial_dof_states (:387)
    for (int k = 0; k < 9; ++k) bv.lamp[k * N + e] = 0.f;
    v.episode[e] = ep + 1;
    v.reset[e] = 0;
    v.progress[e] = 0;
}

// pre_physics_step (:395-413)
__global__ void bbot_pre_kernel(View v, BbotView bv, BallBalanceParams p, const float* __restrict__ actions_in) {
    MI_NO_CONTRACT
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    const bool rs = v.reset[e] != 0;
    if (rs) bbot_reset_env(v, bv, p, e);
    float a[kBbotAct];
    for (int k = 0; k < kBbotAct; ++k) {
        a[k] = fminf(fmaxf(actions_in[(size_t)e * kBbotAct + k], -p.clip_actions), p.clip_actions);   // vec_task.py:374
        v.actions[k * N + e] = a[k];
    }
    for (int d = 0; d < kBbotDof; ++d) {
        float t = bv.targets[d * N + e];
        if (d & 1) t += p.dt * p.action_speed_scale * a[d >> 1];                 // actuated dofs 1, 3, 5 (:405)
        t = fmaxf(fminf(t, p.dof_upper[d]), p.dof_lower[d]);                     // tensor_clamp (:406)
        bv.targets[d * N + e] = rs ? 0.f : t;                                    // :409
    }
}

// gym.simulate(): one physics sub-step
__global__ __launch_bounds__(64) void bbot_substep_kernel(View v, BbotView bv, SimParams P, BallBalanceParams p) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    BbotSim<BM> sim;
    load_sim(sim, v, e);
    float target[kBbotDof];
    sfor<kBbotDof>([&](auto K) MI_LAMBDA { target[K] = bv.targets[K * N + e]; });
    sfor<3>([&](auto K) MI_LAMBDA { sim.ball.pos[K] = bv.ball[K * N + e]; sim.ball.vel[K] = bv.ball[(7 + K) * N + e]; sim.ball.angvel[K] = bv.ball[(10 + K) * N + e]; });
    sfor<4>([&](auto K) MI_LAMBDA { sim.ball.quat[K] = bv.ball[(3 + K) * N + e]; });
    int nc;
    sim.substep(P, phys_of(p), P.dt / (float)P.substeps, target, Strided{v.laml + e, N}, Strided{bv.lamp + e, N}, Strided{v.sensor + e, N}, &nc);
    bv.ncontact[e] = nc;
    store_sim(sim, v, e);
    sfor<3>([&](auto K) MI_LAMBDA { bv.ball[K * N + e] = sim.ball.pos[K]; bv.ball[(7 + K) * N + e] = sim.ball.vel[K]; bv.ball[(10 + K) * N + e] = sim.ball.angvel[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { bv.ball[(3 + K) * N + e] = sim.ball.quat[K]; });
}

// post_physics_step (:415-424): progress++, observations, reward
__global__ __launch_bounds__(64) void bbot_post_kernel(View v, BbotView bv, BallBalanceParams p) {
    const int e0 = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    float q[kBbotDof], qd[kBbotDof], ball[13], sens[18];
    sfor<kBbotDof>([&](auto K) MI_LAMBDA { q[K] = v.dof[K * N + e]; qd[K] = v.dof[(kBbotDof + K) * N + e]; });
    sfor<13>([&](auto K) MI_LAMBDA { ball[K] = bv.ball[K * N + e]; });
    sfor<18>([&](auto K) MI_LAMBDA { sens[K] = v.sensor[K * N + e]; });
    const long long progress = v.progress[e] + 1;
    float obs[kBbotObs], rew;
    long long reset;
    bbot_observations(q, qd, ball, sens, obs);
    bbot_reward(ball, ball + 7, p.ball_radius, v.reset[e], progress, p.max_episode_length, &rew, &reset);
    episode_stats(v, e, valid, rew, reset, progress);
    if (!valid) return;
    v.randomize[e] += 1;
    float* ob = v.obs + (size_t)e * kBbotObs;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * kBbotObs;
    sfor<kBbotObs>([&](auto K) MI_LAMBDA { ob[K] = obs[K]; oc[K] = fminf(fmaxf(obs[K], -v.clip_obs), v.clip_obs); });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)(((float)progress >= p.max_episode_length - 1.f) && (reset != 0));   // vec_task.py:394
}

// __init__ state (:88-112): bot at the tray height, ball at its spawn pose, zero targets, reset_buf = 1 (vec_task.py:318)
__global__ void bbot_init_kernel(View v, BbotView bv, BallBalanceParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    for (int k = 0; k < 13; ++k) {
        const float x = (k == 2) ? p.tray_height : (k == 6 ? 1.f : 0.f);
        v.root[k * N + e] = x; v.init_root[k * N + e] = x;
        bv.ball[k * N + e] = (k < 3) ? p.ball_init_pos[k] : (k == 6 ? 1.f : 0.f);
    }
    for (int d = 0; d < kBbotDof; ++d) bv.targets[d * N + e] = 0.f;
    for (int k = 0; k < 9; ++k) bv.lamp[k * N + e] = 0.f;
    bv.ncontact[e] = 0;
}
__global__ void bbot_reset_ids_kernel(View v, BbotView bv, BallBalanceParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i];
    if (e < 0 || e >= v.N) return;
    bbot_reset_env(v, bv, p, e);
    for (int d = 0; d < kBbotDof; ++d) bv.targets[d * v.N + e] = 0.f;
}

static hipError_t bbot_substeps(const View& v, const BbotView& bv, const SimParams& P, const BallBalanceParams& p, int n, hipStream_t s) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(bbot_substep_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, bv, P, p);
    return hipGetLastError();
}
hipError_t launch_step_ball_balance(const View& v, const BbotView& bv, const SimParams& P, const BallBalanceParams& p, const float* actions,
                                    int cfi, hipStream_t s) {
    hipLaunchKernelGGL(bbot_pre_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, bv, p, actions);
    hipError_t e = bbot_substeps(v, bv, P, p, cfi * P.substeps, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bbot_post_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, bv, p);
    return hipGetLastError();
}
hipError_t launch_simulate_ball_balance(const View& v, const BbotView& bv, const SimParams& P, const BallBalanceParams& p, hipStream_t s) {
    return bbot_substeps(v, bv, P, p, P.substeps, s);
}
hipError_t launch_init_ball_balance(const View& v, const BbotView& bv, const BallBalanceParams& p, hipStream_t s) {
    hipLaunchKernelGGL(bbot_init_kernel, dim3((v.N + 127) / 128), dim3(128), 0, s, v, bv, p);
    return hipGetLastError();
}
hipError_t launch_reset_ball_balance(const View& v, const BbotView& bv, const BallBalanceParams& p, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(bbot_reset_ids_kernel, dim3((n + 127) / 128), dim3(128), 0, s, v, bv, p, ids, n);
    return hipGetLastError();
}

}  // namespace mi

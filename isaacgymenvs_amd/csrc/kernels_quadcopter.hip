// kernels_quadcopter.hip -- Quadcopter (reference isaacgymenvs/tasks/quadcopter.py): free-flying chassis with four tilting
// rotors; position-driven rotor joints (DOF_MODE_POS) and thrust forces on the rotor bodies (apply_rigid_body_force_tensors,
// LOCAL_SPACE).  pre kernel = pre_physics_step (deferred resets, action integration), sub-step kernel = gym.simulate with the
// engine's Drive extras, post kernel = post_physics_step.  One env per lane, 64 envs per wave.
#include "step_kernels.hpp"
#include "task_views.hpp"
#include "gen/model_quadcopter.h"
#include "tasks/quadcopter.hpp"

namespace mi {

using QM = ModelQuadcopter;
static_assert(QM::ND == kQuadDof && QM::NSENS == kQuadRotors && QM::NB == 9, "quadcopter model");


// pre_physics_step (quadcopter.py:276-292)
__global__ void quad_pre_kernel(View v, QuadView qv, QuadcopterParams p, const float* __restrict__ actions_in) {
    MI_NO_CONTRACT
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    const bool rs = v.reset[e] != 0;
    float q[kQuadDof];
    if (rs) {   // reset_idx(reset_env_ids) (:279-281)
        float root[13], qd[kQuadDof];
        const int ep = v.episode[e];
        quadcopter_reset(p, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, root, q, qd);
        for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
        for (int d = 0; d < kQuadDof; ++d) { v.dof[d * N + e] = q[d]; v.dof[(kQuadDof + d) * N + e] = 0.f; v.laml[d * N + e] = 0.f; }
        v.episode[e] = ep + 1;
        v.reset[e] = 0;
        v.progress[e] = 0;
    }
    for (int d = 0; d < kQuadDof; ++d) {
        const float a = fminf(fmaxf(actions_in[(size_t)e * kQuadAct + d], -p.clip_actions), p.clip_actions);   // vec_task.py:374
        v.actions[d * N + e] = a;
        float t = qv.targets[d * N + e] + p.dt * p.dof_action_speed_scale * a;                                 // :284
        t = fmaxf(fminf(t, p.dof_upper[d]), p.dof_lower[d]);                                                   // tensor_clamp (:285)
        if (rs) t = q[d];                                                                                      // :297
        qv.targets[d * N + e] = t;
    }
    for (int k = 0; k < kQuadRotors; ++k) {
        const float a = fminf(fmaxf(actions_in[(size_t)e * kQuadAct + kQuadDof + k], -p.clip_actions), p.clip_actions);
        v.actions[(kQuadDof + k) * N + e] = a;
        float th = qv.thrusts[k * N + e] + p.dt * p.thrust_action_speed_scale * a;                             // :288
        th = fmaxf(fminf(th, p.max_thrust), 0.f);                                                              // :289
        // the reference fills `forces` from the thrusts BEFORE it clears both for the reset envs (:291-296)
        qv.thrusts[k * N + e] = rs ? 0.f : th;
        qv.forces[(3 * QM::sens_body[k] + 2) * N + e] = rs ? 0.f : th;
    }
}

// gym.simulate(): one physics sub-step with the position drives and the rotor thrusts
__global__ __launch_bounds__(64) void quad_substep_kernel(View v, QuadView qv, SimParams P, QuadcopterParams p) {
    extern __shared__ float lds_rows[];
    using S = Sim<QM>;
    static_assert(S::LANES == 64, "quadcopter uses the static row store");
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    S sim;
    load_sim(sim, v, e);
    float tau[kQuadDof], target[kQuadDof], fs[kQuadRotors][3];
    sfor<kQuadDof>([&](auto K) MI_LAMBDA { tau[K] = 0.f; target[K] = qv.targets[K * N + e]; });
    sfor<kQuadRotors>([&](auto K) MI_LAMBDA { fs[K][0] = 0.f; fs[K][1] = 0.f; fs[K][2] = qv.forces[(3 * QM::sens_body[K] + 2) * N + e]; });
    const Drive drv{p.drive_stiffness, p.drive_damping, target, &fs[0][0]};
    const float h = P.dt / (float)P.substeps;
    sim.substep(P, tau, h, RowStore<64>(lds_rows + threadIdx.x), Strided{v.lamc + e, N}, Strided{v.laml + e, N}, Strided{v.sensor + e, N},
                Strided{v.dof_force + e, N}, PlaneGround{}, -1.f, Strided{nullptr, N}, &drv);
    // asset_options.max_angular_velocity (:208): PhysX clamps the angular speed of the body
    {
        const float w2 = sim.root[10] * sim.root[10] + sim.root[11] * sim.root[11] + sim.root[12] * sim.root[12];
        const float lim = p.max_angular_velocity;
        if (w2 > lim * lim) {
            const float sc = lim * MI_RSQ(w2);
            sim.root[10] *= sc; sim.root[11] *= sc; sim.root[12] *= sc;
        }
    }
    store_sim(sim, v, e);
}

// post_physics_step (:294-302): progress++, observations, reward
__global__ __launch_bounds__(64) void quad_post_kernel(View v, QuadView qv, QuadcopterParams p) {
    const int e0 = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    float root[13], q[kQuadDof];
    sfor<13>([&](auto K) MI_LAMBDA { root[K] = v.root[K * N + e]; });
    sfor<kQuadDof>([&](auto K) MI_LAMBDA { q[K] = v.dof[K * N + e]; });
    const long long progress = v.progress[e] + 1;
    float obs[kQuadObs], rew;
    long long reset;
    quadcopter_observations(root, q, obs);
    quadcopter_reward(root, progress, p.max_episode_length, &rew, &reset);
    episode_stats(v, e, valid, rew, reset, progress);
    if (!valid) return;
    v.randomize[e] += 1;
    float* ob = v.obs + (size_t)e * kQuadObs;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * kQuadObs;
    sfor<kQuadObs>([&](auto K) MI_LAMBDA { ob[K] = obs[K]; oc[K] = fminf(fmaxf(obs[K], -v.clip_obs), v.clip_obs); });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)(((float)progress >= p.max_episode_length - 1.f) && (reset != 0));   // vec_task.py:394
}

// __init__ state (:62-101): craft at the default pose, zero targets / thrusts, reset_buf = 1 (vec_task.py:318)
__global__ void quad_init_kernel(View v, QuadView qv, QuadcopterParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    for (int k = 0; k < 13; ++k) { const float x = (k == 2) ? p.init_height : (k == 6 ? 1.f : 0.f); v.root[k * N + e] = x; v.init_root[k * N + e] = x; }
    for (int d = 0; d < kQuadDof; ++d) qv.targets[d * N + e] = 0.f;
    for (int k = 0; k < kQuadRotors; ++k) qv.thrusts[k * N + e] = 0.f;
    for (int k = 0; k < 3 * QM::NB; ++k) qv.forces[k * N + e] = 0.f;
}
__global__ void quad_reset_ids_kernel(View v, QuadView qv, QuadcopterParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i], N = v.N;
    if (e < 0 || e >= N) return;
    float root[13], q[kQuadDof], qd[kQuadDof];
    const int ep = v.episode[e];
    quadcopter_reset(p, v.seed, (uint32_t)(v.env_offset + e), (uint32_t)ep, root, q, qd);
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
    for (int d = 0; d < kQuadDof; ++d) { v.dof[d * N + e] = q[d]; v.dof[(kQuadDof + d) * N + e] = 0.f; v.laml[d * N + e] = 0.f; }
    v.episode[e] = ep + 1;
    v.reset[e] = 0;       // :273-274
    v.progress[e] = 0;
}

static hipError_t quad_substeps(const View& v, const QuadView& qv, const SimParams& P, const QuadcopterParams& p, int n, hipStream_t s) {
    constexpr size_t lds = lds_bytes<QM>();
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)quad_substep_kernel, lds, &configured); e != hipSuccess) return e;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(quad_substep_kernel, dim3((v.N + 63) / 64), dim3(64), lds, s, v, qv, P, p);
    return hipGetLastError();
}
hipError_t launch_step_quadcopter(const View& v, const QuadView& qv, const SimParams& P, const QuadcopterParams& p, const float* actions,
                                  int cfi, hipStream_t s) {
    // 64-thread blocks like the sub-step kernel: block b of every kernel of the step lands on the same XCD (own L2)
    hipLaunchKernelGGL(quad_pre_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, qv, p, actions);
    hipError_t e = quad_substeps(v, qv, P, p, cfi * P.substeps, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(quad_post_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, qv, p);
    return hipGetLastError();
}
hipError_t launch_simulate_quadcopter(const View& v, const QuadView& qv, const SimParams& P, const QuadcopterParams& p, hipStream_t s) {
    return quad_substeps(v, qv, P, p, P.substeps, s);
}
hipError_t launch_init_quadcopter(const View& v, const QuadView& qv, const QuadcopterParams& p, hipStream_t s) {
    hipLaunchKernelGGL(quad_init_kernel, dim3((v.N + 127) / 128), dim3(128), 0, s, v, qv, p);
    return hipGetLastError();
}
hipError_t launch_reset_quadcopter(const View& v, const QuadView& qv, const QuadcopterParams& p, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(quad_reset_ids_kernel, dim3((n + 127) / 128), dim3(128), 0, s, v, qv, p, ids, n);
    return hipGetLastError();
}

// stand-alone replacement of the jitted function (row-major contiguous tensors)
__global__ void quadcopter_reward_kernel(int n, const float* root_positions, const float* root_quats, const float* root_linvels,
                                         const float* root_angvels, const long long* progress_buf, float max_episode_length, float* rew,
                                         long long* reset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float root[13];
    for (int k = 0; k < 3; ++k) { root[k] = root_positions[3 * (size_t)e + k]; root[7 + k] = root_linvels[3 * (size_t)e + k]; root[10 + k] = root_angvels[3 * (size_t)e + k]; }
    for (int k = 0; k < 4; ++k) root[3 + k] = root_quats[4 * (size_t)e + k];
    quadcopter_reward(root, progress_buf[e], max_episode_length, rew + e, reset + e);
}
hipError_t launch_quadcopter_reward(int n, const float* root_positions, const float* root_quats, const float* root_linvels,
                                    const float* root_angvels, const long long* progress_buf, float max_episode_length, float* rew,
                                    long long* reset, hipStream_t s) {
    hipLaunchKernelGGL(quadcopter_reward_kernel, dim3((n + 127) / 128), dim3(128), 0, s, n, root_positions, root_quats, root_linvels,
                       root_angvels, progress_buf, max_episode_length, rew, reset);
    return hipGetLastError();
}

}  // namespace mi

// kernels_ingenuity.hip -- Ingenuity (reference isaacgymenvs/tasks/ingenuity.py): a free-flying chassis under Mars gravity with two
// coaxial rotors; the policy commands one thrust vector per rotor, applied on the rotor bodies in their local frames
// (apply_rigid_body_force_tensors, LOCAL_SPACE, :354).  pre kernel = pre_physics_step (periodic targets, deferred resets, thrusts),
// sub-step kernel = gym.simulate with the engine's Drive extras (zero gains: the joints are passive), post kernel =
// post_physics_step.  One env per lane, 64 envs per wave.
#include "step_kernels.hpp"
#include "task_views.hpp"
#include "gen/model_ingenuity.h"
#include "tasks/ingenuity.hpp"

namespace mi {

using IM = ModelIngenuity;
static_assert(IM::ND == kIngDof && IM::NSENS == kIngRotors && IM::NB + 1 == kIngBodies, "ingenuity model");


static __device__ __forceinline__ void ing_set_target(const View& v, const IngenuityView& iv, int e, const float* target) {
    const int N = v.N;
    for (int k = 0; k < 3; ++k) { iv.target[k * N + e] = target[k]; iv.marker[k * N + e] = target[k] + (k == 2 ? 0.4f : 0.f); }   // :288-290
}

// pre_physics_step (:321-354)
__global__ void ing_pre_kernel(View v, IngenuityView iv, IngenuityParams p, const float* __restrict__ actions_in) {
    MI_NO_CONTRACT
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    const int ep = v.episode[e];
    const long long progress = v.progress[e];
    float target[3];
    if (progress % p.target_period == 0) {                                          // :324-327
        ingenuity_target(v.seed, genv, (uint32_t)ep, 8u + 3u * (uint32_t)(progress / p.target_period), target);
        ing_set_target(v, iv, e, target);
    }
    const bool rs = v.reset[e] != 0;
    if (rs) {                                                                       // reset_idx(reset_env_ids) (:329-332)
        float root[13];
        ingenuity_target(v.seed, genv, (uint32_t)ep, 3u, target);
        ing_set_target(v, iv, e, target);
        ingenuity_reset_root(p, v.seed, genv, (uint32_t)ep, root);
        for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
        // set_dof_state_tensor_indexed pushes the tensor as it is: positions unchanged, the two visual rotors' speeds overwritten
        v.dof[(kIngDof + 1) * N + e] = -p.rotor_speed;
        v.dof[(kIngDof + 3) * N + e] = p.rotor_speed;
        for (int d = 0; d < kIngDof; ++d) v.laml[d * N + e] = 0.f;
        v.episode[e] = ep + 1;
        v.reset[e] = 0;
        v.progress[e] = 0;
    }
    float a[kIngAct];
    for (int k = 0; k < kIngAct; ++k) {
        a[k] = fminf(fmaxf(actions_in[(size_t)e * kIngAct + k], -p.clip_actions), p.clip_actions);   // vec_task.py:374
        v.actions[k * N + e] = a[k];
    }
    for (int r = 0; r < kIngRotors; ++r) {                                          // :337-348
        const float vertical = fminf(fmaxf(a[3 * r + 2] * p.thrust_action_speed_scale, -p.thrust_upper_limit), p.thrust_upper_limit);
        float th[3];
        th[2] = p.dt * vertical;
        for (int k = 0; k < 2; ++k) th[k] = th[2] * fminf(fmaxf(a[3 * r + k], -p.thrust_lateral_component), p.thrust_lateral_component);
        for (int k = 0; k < 3; ++k) {
            const float x = rs ? 0.f : th[k];                                       // cleared for the reset envs (:350-352)
            iv.thrusts[(3 * r + k) * N + e] = x;
            iv.forces[(3 * IM::sens_body[r] + k) * N + e] = x;
        }
    }
}

// gym.simulate(): one physics sub-step with the thrust vectors on the two rotor bodies
__global__ __launch_bounds__(64) void ing_substep_kernel(View v, IngenuityView iv, SimParams P, IngenuityParams p) {
    extern __shared__ float lds_rows[];
    using S = Sim<IM>;
    static_assert(S::LANES == 64, "ingenuity uses the static row store");
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    S sim;
    load_sim(sim, v, e);
    float tau[kIngDof], target[kIngDof], fs[kIngRotors][3];
    sfor<kIngDof>([&](auto K) MI_LAMBDA { tau[K] = 0.f; target[K] = 0.f; });
    sfor<kIngRotors>([&](auto R) MI_LAMBDA { sfor<3>([&](auto K) MI_LAMBDA { fs[R][K] = iv.forces[(3 * IM::sens_body[R] + K) * N + e]; }); });
    const Drive drv{0.f, 0.f, target, &fs[0][0]};                                   // stiffness = damping = 0 (:264-267)
    const float h = P.dt / (float)P.substeps;
    sim.substep(P, tau, h, RowStore<64>(lds_rows + threadIdx.x), Strided{v.lamc + e, N}, Strided{v.laml + e, N}, Strided{v.sensor + e, N},
                Strided{v.dof_force + e, N}, PlaneGround{}, -1.f, Strided{nullptr, N}, &drv);
    // asset_options.max_angular_velocity (:248): PhysX clamps the angular speed of the body
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

// post_physics_step (:356-365): progress++, observations, reward
__global__ __launch_bounds__(64) void ing_post_kernel(View v, IngenuityView iv, IngenuityParams p) {
    const int e0 = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    float root[13], target[3];
    sfor<13>([&](auto K) MI_LAMBDA { root[K] = v.root[K * N + e]; });
    sfor<3>([&](auto K) MI_LAMBDA { target[K] = iv.target[K * N + e]; });
    const long long progress = v.progress[e] + 1;
    float obs[kIngObs], rew;
    long long reset;
    ingenuity_observations(root, target, obs);
    ingenuity_reward(root, target, root + 3, root + 10, progress, p.max_episode_length, &rew, &reset);   // tasks/jit_twins.hpp
    episode_stats(v, e, valid, rew, reset, progress);
    if (!valid) return;
    v.randomize[e] += 1;
    float* ob = v.obs + (size_t)e * kIngObs;
    float* oc = v.obs_out + ((size_t)v.ring * N + e) * kIngObs;
    sfor<kIngObs>([&](auto K) MI_LAMBDA { ob[K] = obs[K]; oc[K] = fminf(fmaxf(obs[K], -v.clip_obs), v.clip_obs); });
    v.rew[e] = rew;
    v.reset[e] = reset;
    v.progress[e] = progress;
    v.timeout[e] = (unsigned char)(((float)progress >= p.max_episode_length - 1.f) && (reset != 0));   // vec_task.py:394
}

// __init__ state (:63-97): craft and marker at the default pose, target (0, 0, 1), zero thrusts, reset_buf = 1 (vec_task.py:318)
__global__ void ing_init_kernel(View v, IngenuityView iv, IngenuityParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    for (int k = 0; k < 13; ++k) {
        const float x = (k == 2) ? p.init_height : (k == 6 ? 1.f : 0.f);
        v.root[k * N + e] = x; v.init_root[k * N + e] = x; iv.marker[k * N + e] = x;
    }
    for (int k = 0; k < 3; ++k) iv.target[k * N + e] = (k == 2) ? 1.f : 0.f;
    for (int k = 0; k < 3 * kIngRotors; ++k) iv.thrusts[k * N + e] = 0.f;
    for (int k = 0; k < 3 * kIngBodies; ++k) iv.forces[k * N + e] = 0.f;
}
__global__ void ing_reset_ids_kernel(View v, IngenuityView iv, IngenuityParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i], N = v.N;
    if (e < 0 || e >= N) return;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    const int ep = v.episode[e];
    float root[13], target[3];
    ingenuity_target(v.seed, genv, (uint32_t)ep, 3u, target);
    ing_set_target(v, iv, e, target);
    ingenuity_reset_root(p, v.seed, genv, (uint32_t)ep, root);
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = root[k];
    v.dof[(kIngDof + 1) * N + e] = -p.rotor_speed;
    v.dof[(kIngDof + 3) * N + e] = p.rotor_speed;
    for (int d = 0; d < kIngDof; ++d) v.laml[d * N + e] = 0.f;
    v.episode[e] = ep + 1;
    v.reset[e] = 0;       // :316-317
    v.progress[e] = 0;
}

static hipError_t ing_substeps(const View& v, const IngenuityView& iv, const SimParams& P, const IngenuityParams& p, int n, hipStream_t s) {
    constexpr size_t lds = lds_bytes<IM>();
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)ing_substep_kernel, lds, &configured); e != hipSuccess) return e;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(ing_substep_kernel, dim3((v.N + 63) / 64), dim3(64), lds, s, v, iv, P, p);
    return hipGetLastError();
}
hipError_t launch_step_ingenuity(const View& v, const IngenuityView& iv, const SimParams& P, const IngenuityParams& p, const float* actions,
                                 int cfi, hipStream_t s) {
    hipLaunchKernelGGL(ing_pre_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, iv, p, actions);
    hipError_t e = ing_substeps(v, iv, P, p, cfi * P.substeps, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ing_post_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, iv, p);
    return hipGetLastError();
}
hipError_t launch_simulate_ingenuity(const View& v, const IngenuityView& iv, const SimParams& P, const IngenuityParams& p, hipStream_t s) {
    return ing_substeps(v, iv, P, p, P.substeps, s);
}
hipError_t launch_init_ingenuity(const View& v, const IngenuityView& iv, const IngenuityParams& p, hipStream_t s) {
    hipLaunchKernelGGL(ing_init_kernel, dim3((v.N + 127) / 128), dim3(128), 0, s, v, iv, p);
    return hipGetLastError();
}
hipError_t launch_reset_ingenuity(const View& v, const IngenuityView& iv, const IngenuityParams& p, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(ing_reset_ids_kernel, dim3((n + 127) / 128), dim3(128), 0, s, v, iv, p, ids, n);
    return hipGetLastError();
}

}  // namespace mi

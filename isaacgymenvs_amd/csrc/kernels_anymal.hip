// kernels_anymal.hip -- AnymalTerrain: ModelAnymal sub-step on the height field + the task's post_physics_step kernel.
#include "step_kernels.hpp"
#include "gen/model_anymal.h"
#include "tasks/anymal_step.hpp"

namespace mi {

static_assert(ModelAnymal::ND == kAnymalDof, "anymal dof count");

// The per-env bodies of the task's passes live in tasks/anymal_step.hpp (shared with the CPU backend, cpu/cpu_anymal.cpp); the kernels here
// supply the launch shapes and the cross-lane reductions: wave shuffles + one atomic per wave and statistic.
struct DevRed {
    __device__ __forceinline__ void cmdnorm(const View& v, float acc) const {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0 && acc > 0.f) atomicAdd(v.ep_stats + 15, acc);
    }
    // extras["episode"] partial sums: wave reduction, one atomic per wave and statistic (:421-425)
    __device__ __forceinline__ void extras(const View& v, const float (&st_sums)[kAnymalSums], float st_cnt, float lv) const {
        st_cnt = wave_sum(st_cnt);
        lv = wave_sum(lv);
        float red[kAnymalSums];
        sfor<kAnymalSums>([&](auto K) MI_LAMBDA { red[K] = wave_sum(st_sums[K]); });
        if ((threadIdx.x & 63) == 0) {
            if (st_cnt > 0.f) {
                sfor<kAnymalSums>([&](auto K) MI_LAMBDA { atomicAdd(v.ep_stats + K, red[K]); });
                atomicAdd(v.ep_stats + 13, st_cnt);
            }
            atomicAdd(v.ep_stats + 14, lv);
        }
    }
    __device__ __forceinline__ void episode(const View& v, int e, bool valid, float rew, long long reset, long long progress) const {
        episode_stats(v, e, valid, rew, reset, progress);
    }
};

// ------------------------------------------------------------------------------------------------ curriculum pre-pass (anymal_cmdnorm_env)
__global__ __launch_bounds__(64) void anymal_cmdnorm_kernel(View v, AnymalParams p) {
    anymal_cmdnorm_env(v, p, (int)(blockIdx.x * 64 + threadIdx.x), DevRed{});
}

// ------------------------------------------------------------------------------------------------ post_physics_step: one env per lane
template <bool OBS_ALL>
__global__ __launch_bounds__(64) void anymal_post_kernel(View v, AnymalParams p, AnymalTerrainDesc T, unsigned step_counter) {
    anymal_post_env<3 * ModelAnymal::NSPH, DevRed, OBS_ALL>(v, p, T, step_counter, (int)(blockIdx.x * 64 + threadIdx.x), DevRed{});
}

// extras["episode"] (:421-425): runs in block 0 of the height-scan kernel (the last kernel of the step: every post block has finished its
// atomics by then) and leaves ep_stats zeroed for the next step, which saves a memset and a one-block launch per step.
__device__ __forceinline__ void anymal_extras(const View& v, const AnymalParams& p, int k) {
    float mine = 0.f, cnt = 0.f;
    if (k < 16) { mine = v.ep_stats[k]; cnt = v.ep_stats[13]; }
    __syncthreads();                                                                    // all reads before the zeroing below
    anymal_extras_slot(v, p, k, mine, cnt);
}

// get_heights (:515-538) + the height columns of compute_observations (:311) + their noise (:481-482), one thread per
// (env, scan point): 573 k threads at 4096 envs instead of a 140-iteration gather loop in each of 64 waves.
// PLAIN (option fused_post): the threads also write the 39 observation columns whose inputs are in memory after the post pass (anymal_obs_column) --
// one thread per (env, column) with consecutive threads on consecutive columns of a row, where the post kernel's one lane per env wrote 78 more
// row-strided stores and drew 39 more noise values in its dependency chain (64 waves at 4096 envs).
template <bool PLAIN>
__global__ void anymal_heights_kernel(View v, AnymalParams p, AnymalTerrainDesc T, unsigned step_counter) {
    constexpr int PER_ENV = kAnymalHeightPts + (PLAIN ? kAnymalPlainCols : 0);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) anymal_extras(v, p, threadIdx.x);
    if (t >= v.N * PER_ENV) return;
    const int e = t / PER_ENV, k = t - e * PER_ENV;
    if constexpr (PLAIN) {
        // (row order: 9 .. 35 come before the scan's 36 .. 175, the actions after it)
        if (k < 3 + 2 * kAnymalDof) { anymal_obs_column(v, p, step_counter, e, k); return; }
        if (k >= 3 + 2 * kAnymalDof + kAnymalHeightPts) { anymal_obs_column(v, p, step_counter, e, k - kAnymalHeightPts); return; }
        anymal_height_point(v, p, T, step_counter, e, k - (3 + 2 * kAnymalDof));
    } else {
        anymal_height_point(v, p, T, step_counter, e, k);
    }
}

// ------------------------------------------------------------------------------------------------ init / explicit reset
__global__ void anymal_init_kernel(View v, AnymalParams p, AnymalTerrainDesc T, int max_init_level) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.N) return;
    anymal_init_env<3 * ModelAnymal::NB>(v, p, T, max_init_level, e);
}

// reset_idx(env_ids) outside step(), curriculum part: torch.norm(self.commands[env_ids, :2]) over the envs of THIS call (:431), left in
// ep_stats[15] for the reset kernel below (one block; mode 1 clears the slot again for the next step's own accumulation)
__global__ __launch_bounds__(256) void anymal_reset_cmdnorm_kernel(View v, const long long* __restrict__ ids, int n, int mode) {
    if (mode == 1) { if (threadIdx.x == 0) v.ep_stats[15] = 0.f; return; }
    __shared__ float part[4];
    const int N = v.N;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int e = (int)ids[i];
        if (e < 0 || e >= N) continue;
        const float cx = v.commands[e], cy = v.commands[N + e];
        acc += cx * cx + cy * cy;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) v.ep_stats[15] = (part[0] + part[1]) + (part[2] + part[3]);
}

// reset_idx(env_ids) (:384-425) outside step()
__global__ void anymal_reset_kernel(View v, AnymalParams p, AnymalTerrainDesc T, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i];
    if (e < 0 || e >= v.N) return;
    anymal_reset_env<3 * ModelAnymal::NSPH>(v, p, T, e);
}

static HeightfieldGround ground_of(const AnymalTerrainDesc& T) {
    return HeightfieldGround{T.hs, T.rows, T.cols, T.hscale, T.vscale, T.border,
                             T.slope_threshold > 0.f ? T.slope_threshold * T.hscale / T.vscale : 3.0e38f, T.walls};
}
static ActParams act_of(const AnymalParams& tp) {
    ActParams ap{};
    ap.clip = tp.clip_actions; ap.scale = tp.action_scale; ap.nact = kAnymalDof; ap.mode = 1;
    ap.kp = tp.kp; ap.kd = tp.kd; ap.torque_limit = tp.torque_limit;
    for (int d = 0; d < kAnymalDof; ++d) ap.gear[d] = tp.default_dof_pos[d];
    return ap;
}

// VecTask.step for AnymalTerrain: `decimation` sim steps with the PD torques recomputed before each (:443-451), then
// control_freq_inv more sim steps with the last torques (the base class simulates again, vec_task.py:379-382; the task
// YAML has no controlFrequencyInv => 1), then post_physics_step.
// The task's lagging dof-state tensor (View::dof_api) for the forms that launch once per sim sub-step (fused_sub 0, the one-wave form): the step's
// FIRST PD evaluation -- on the joint state of the task's last refresh, anymal_terrain.py:443-446 -- by this kernel (it also clamps and stores the
// actions, vec_task.py:374), so that the first sub-step launch runs on stored efforts; the refresh at the end of the decimation loop is a
// device-to-device copy.  (The one-launch form does both inside the kernel, mw_kernels.hpp mw_role_fused.)
__global__ __launch_bounds__(256) void anymal_lag_pd_kernel(View v, ActParams ap, const float* __restrict__ actions_in, int src) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    constexpr int ND = kAnymalDof;
    if (i >= v.N * ND) return;
    const int k = i / v.N, e = i - k * v.N;          // consecutive threads: consecutive envs of one dof (SoA)
    float a;
    if (src == ACT_FROM_ACTIONS) {
        a = fminf(fmaxf(actions_in[(size_t)e * ND + k], -ap.clip), ap.clip);
        v.actions[(size_t)k * v.N + e] = a;
    } else {
        a = v.actions[(size_t)k * v.N + e];
    }
    const float u = ap.kp * (ap.scale * a + ap.gear[k] - v.dof_api[(size_t)k * v.N + e]) - ap.kd * v.dof_api[(size_t)(ND + k) * v.N + e];
    v.tau[(size_t)k * v.N + e] = fminf(fmaxf(u, -ap.torque_limit), ap.torque_limit);
}

hipError_t launch_step_anymal(const View& v, const SimParams& P, const AnymalParams& tp, const AnymalTerrainDesc& T,
                              const float* actions, int cfi, unsigned step_counter, hipStream_t s) {
    const ActParams ap = act_of(tp);
    hipError_t e;
    bool cmdnorm_in_launch = false;
    const int n_dec = tp.decimation * P.substeps;
    const bool fused = v.mw != 0 && v.fused_sub != 0 && (tp.decimation + cfi) * P.substeps > 1;
    if (T.hs != nullptr && v.dof_api != nullptr && !fused) {
        // per-sub-step launches with the lagging tensor: [lag PD] sub-step 0 on stored efforts, the rest of the decimation loop with the PD law on the
        // current state, [refresh], the base class's simulate() calls on the last efforts
        const int first = prepare_actions(v, ap, actions, ACT_FROM_ACTIONS, s);
        hipLaunchKernelGGL(anymal_lag_pd_kernel, dim3((v.N * kAnymalDof + 255) / 256), dim3(256), 0, s, v, ap, actions, first);
        const size_t bytes = (size_t)2 * kAnymalDof * v.N * sizeof(float);
        if (v.mw != 0) {
            e = launch_substeps_mw<ModelAnymal, HeightfieldGround>(v, P, ap, actions, n_dec, ACT_STORED_TAU, ACT_FROM_STORED_ACTIONS, s, ground_of(T), 0, nullptr);
            if (e != hipSuccess) return e;
            if ((e = hipMemcpyAsync(v.dof_api, v.dof, bytes, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
            e = launch_substeps_mw<ModelAnymal, HeightfieldGround>(v, P, ap, nullptr, cfi * P.substeps, ACT_STORED_TAU, ACT_STORED_TAU, s, ground_of(T), 0, nullptr);
        } else {
            e = launch_substeps<ModelAnymal, HeightfieldGround>(v, P, ap, actions, n_dec, ACT_STORED_TAU, ACT_FROM_STORED_ACTIONS, s, ground_of(T));
            if (e != hipSuccess) return e;
            if ((e = hipMemcpyAsync(v.dof_api, v.dof, bytes, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
            e = launch_substeps<ModelAnymal, HeightfieldGround>(v, P, ap, nullptr, cfi * P.substeps, ACT_STORED_TAU, ACT_STORED_TAU, s, ground_of(T));
        }
    } else if (T.hs != nullptr) {
        if (v.mw != 0) {    // limb-per-wave form: one call (with option fused_sub: one LAUNCH) for the decimation steps + the base class's simulate()
            // (with the fused launch the curriculum pre-pass -- anymal_cmdnorm_env -- runs on the trunk wave at its end: one kernel less)
            cmdnorm_in_launch = tp.curriculum && v.fused_sub != 0 && (tp.decimation + cfi) * P.substeps > 1;
            const MwCmdNormTail cn{tp.allow_knee_contacts ? 1 : 0, (float)tp.max_episode_length};
            e = launch_substeps_mw<ModelAnymal, HeightfieldGround>(v, P, ap, actions, (tp.decimation + cfi) * P.substeps,
                                                                   prepare_actions(v, ap, actions, ACT_FROM_ACTIONS, s), ACT_FROM_STORED_ACTIONS, s,
                                                                   ground_of(T), cfi * P.substeps, cmdnorm_in_launch ? &cn : nullptr);
        } else {
            e = launch_substeps<ModelAnymal, HeightfieldGround>(v, P, ap, actions, tp.decimation * P.substeps, ACT_FROM_ACTIONS,
                                                                ACT_FROM_STORED_ACTIONS, s, ground_of(T));
            if (e != hipSuccess) return e;
            e = launch_substeps<ModelAnymal, HeightfieldGround>(v, P, ap, nullptr, cfi * P.substeps, ACT_STORED_TAU, ACT_STORED_TAU, s,
                                                                ground_of(T));
        }
    } else {
        return hipErrorInvalidValue;  // mi_engine_step refuses to run AnymalTerrain before mi_engine_set_terrain
    }
    if (e != hipSuccess) return e;
    if (tp.curriculum && !cmdnorm_in_launch) hipLaunchKernelGGL(anymal_cmdnorm_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, tp);
    if (v.fused_post != 0) {    // the observation columns that do not need pre-reset quantities: by the scan kernel's threads
        hipLaunchKernelGGL(anymal_post_kernel<false>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, tp, T, step_counter);
        hipLaunchKernelGGL(anymal_heights_kernel<true>, dim3((v.N * (kAnymalHeightPts + kAnymalPlainCols) + 255) / 256), dim3(256), 0, s, v, tp, T, step_counter);
    } else {
        hipLaunchKernelGGL(anymal_post_kernel<true>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, tp, T, step_counter);
        hipLaunchKernelGGL(anymal_heights_kernel<false>, dim3((v.N * kAnymalHeightPts + 255) / 256), dim3(256), 0, s, v, tp, T, step_counter);
    }
    return hipGetLastError();
}
hipError_t launch_simulate_anymal(const View& v, const SimParams& P, const AnymalTerrainDesc& T, hipStream_t s) {
    ActParams ap{};
    if (T.hs == nullptr) return hipErrorInvalidValue;
    return launch_substeps<ModelAnymal, HeightfieldGround>(v, P, ap, nullptr, P.substeps, ACT_STORED_TAU, ACT_STORED_TAU, s, ground_of(T));
}
hipError_t launch_init_anymal(const View& v, const AnymalParams& tp, const AnymalTerrainDesc& T, int max_init_level, hipStream_t s) {
    hipLaunchKernelGGL(anymal_init_kernel, dim3((v.N + 127) / 128), dim3(128), 0, s, v, tp, T, max_init_level);
    return hipGetLastError();
}
hipError_t launch_reset_anymal(const View& v, const AnymalParams& tp, const AnymalTerrainDesc& T, const long long* ids, int n, hipStream_t s) {
    const bool levels = tp.curriculum && T.hs != nullptr;
    if (levels) hipLaunchKernelGGL(anymal_reset_cmdnorm_kernel, dim3(1), dim3(256), 0, s, v, ids, n, 0);
    hipLaunchKernelGGL(anymal_reset_kernel, dim3((n + 127) / 128), dim3(128), 0, s, v, tp, T, ids, n);
    if (levels) hipLaunchKernelGGL(anymal_reset_cmdnorm_kernel, dim3(1), dim3(256), 0, s, v, ids, n, 1);
    return hipGetLastError();
}


// =================================================================================================== Anymal (flat ground)
// post_physics_step of isaacgymenvs/tasks/anymal.py:231-241 (anymal_flat_post_env)
__global__ __launch_bounds__(64) void anymal_flat_post_kernel(View v, AnymalFlatParams p) {
    anymal_flat_post_env<3 * ModelAnymal::NSPH>(v, p, (int)(blockIdx.x * 64 + threadIdx.x), DevRed{});
}

// constructor state (anymal.py:141-146): initial_root_states := base_init_state, then reset_idx(arange(num_envs))
__global__ void anymal_flat_init_kernel(View v, AnymalFlatParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.N) return;
    anymal_flat_init_env<3 * ModelAnymal::NB>(v, p, e);
}
__global__ void anymal_flat_reset_kernel(View v, AnymalFlatParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (i >= (ids ? n : N)) return;
    const int e = ids ? (int)ids[i] : i;
    if (e < 0 || e >= N) return;
    anymal_flat_reset_env<3 * ModelAnymal::NSPH>(v, p, e);
}
static ActParams act_of(const AnymalFlatParams& tp) {
    ActParams ap{};
    ap.clip = tp.clip_actions; ap.scale = tp.action_scale; ap.nact = kAnymalDof; ap.mode = 1;
    ap.kp = tp.kp; ap.kd = tp.kd; ap.torque_limit = tp.torque_limit;
    for (int d = 0; d < kAnymalDof; ++d) ap.gear[d] = tp.default_dof_pos[d];
    return ap;
}
// VecTask.step for Anymal: position targets = action_scale * a + default (anymal.py:226-229) held for the control step; the
// PD drive (stiffness / damping set on the dofs, :203-206) is evaluated at every physics sub-step like PhysX does.
hipError_t launch_step_anymal_flat(const View& v, const SimParams& P, const AnymalFlatParams& tp, const float* actions, int cfi,
                                   hipStream_t s) {
    // option actor_tensors: the kernel instantiation that reads the `actor_params` factors (kernels_scaled_anymal.hip)
    hipError_t e = v.actor_scale != nullptr
                       ? launch_substeps_scaled<ModelAnymal, PlaneGroundNF>(v, P, act_of(tp), actions, cfi * P.substeps,
                                                                            prepare_actions(v, act_of(tp), actions, ACT_FROM_ACTIONS, s), ACT_FROM_STORED_ACTIONS, s, PlaneGroundNF{})
                       : launch_substeps<ModelAnymal, PlaneGroundNF>(v, P, act_of(tp), actions, cfi * P.substeps, ACT_FROM_ACTIONS, ACT_FROM_STORED_ACTIONS, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(anymal_flat_post_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, tp);
    return hipGetLastError();
}
// gym.simulate() alone: the drive keeps tracking the targets of the last step (stored actions)
hipError_t launch_simulate_anymal_flat(const View& v, const SimParams& P, const AnymalFlatParams& tp, hipStream_t s) {
    if (v.actor_scale != nullptr)
        return launch_substeps_scaled<ModelAnymal, PlaneGroundNF>(v, P, act_of(tp), nullptr, P.substeps, ACT_FROM_STORED_ACTIONS, ACT_FROM_STORED_ACTIONS, s,
                                                                  PlaneGroundNF{});
    return launch_substeps<ModelAnymal, PlaneGroundNF>(v, P, act_of(tp), nullptr, P.substeps, ACT_FROM_STORED_ACTIONS,
                                                       ACT_FROM_STORED_ACTIONS, s);
}
hipError_t launch_init_anymal_flat(const View& v, const AnymalFlatParams& tp, hipStream_t s) {
    hipLaunchKernelGGL(anymal_flat_init_kernel, dim3((v.N + 127) / 128), dim3(128), 0, s, v, tp);
    hipLaunchKernelGGL(anymal_flat_reset_kernel, dim3((v.N + 127) / 128), dim3(128), 0, s, v, tp, (const long long*)nullptr, v.N);
    return hipGetLastError();
}
hipError_t launch_reset_anymal_flat(const View& v, const AnymalFlatParams& tp, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(anymal_flat_reset_kernel, dim3((n + 127) / 128), dim3(128), 0, s, v, tp, ids, n);
    return hipGetLastError();
}

// stand-alone replacements of the jitted functions (row-major contiguous tensors, reference argument meaning)
__global__ void anymal_obs_kernel(int n, AnymalFlatParams p, const float* root_states, const float* commands, const float* dof_pos,
                                  const float* dof_vel, const float* actions, float* obs) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float o[kAnymalFlatObs];
    anymal_flat_observations(p, root_states + 13 * (size_t)e, commands + 3 * (size_t)e, dof_pos + kAnymalDof * (size_t)e,
                             dof_vel + kAnymalDof * (size_t)e, actions + kAnymalDof * (size_t)e, o);
    for (int k = 0; k < kAnymalFlatObs; ++k) obs[(size_t)e * kAnymalFlatObs + k] = o[k];
}
__global__ void anymal_reward_kernel(int n, AnymalFlatParams p, const float* root_states, const float* commands, const float* torques,
                                     const float* contact_forces, int num_bodies, const long long* episode_lengths, float* rew,
                                     long long* reset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* cf = contact_forces + (size_t)e * num_bodies * 3;
    float knee[4][3];
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 3; ++j) knee[k][j] = cf[3 * anymal_knee_body(k) + j];
    anymal_flat_reward(p, root_states + 13 * (size_t)e, commands + 3 * (size_t)e, torques + kAnymalDof * (size_t)e, cf, knee,
                       episode_lengths[e], rew + e, reset + e);
}
hipError_t launch_anymal_obs(int n, const AnymalFlatParams& p, const float* root_states, const float* commands, const float* dof_pos,
                             const float* dof_vel, const float* actions, float* obs, hipStream_t s) {
    hipLaunchKernelGGL(anymal_obs_kernel, dim3((n + 127) / 128), dim3(128), 0, s, n, p, root_states, commands, dof_pos, dof_vel, actions, obs);
    return hipGetLastError();
}
hipError_t launch_anymal_reward(int n, const AnymalFlatParams& p, const float* root_states, const float* commands, const float* torques,
                                const float* contact_forces, int num_bodies, const long long* episode_lengths, float* rew,
                                long long* reset, hipStream_t s) {
    hipLaunchKernelGGL(anymal_reward_kernel, dim3((n + 127) / 128), dim3(128), 0, s, n, p, root_states, commands, torques, contact_forces,
                       num_bodies, episode_lengths, rew, reset);
    return hipGetLastError();
}

}  // namespace mi

// hand_task_kernels.hpp -- the in-hand manipulation tasks on core/hand_engine.hpp, a template over the task (hand_kernels.hpp: ShadowHandTask =
// reference isaacgymenvs/tasks/shadow_hand.py, AllegroHandTask = isaacgymenvs/tasks/allegro_hand.py; the two files share their control flow
// line by line -- the Allegro task has 16 dofs, all of them driven, no tendons, no fingertip states and no force sensors in its observations):
// pre_physics_step (deferred resets, actions -> position targets), hand + object physics sub-steps, post_physics_step (fingertip states,
// full_state observation, compute_hand_reward).  One env per lane; 32 envs per wave in the physics kernel (the compact contact store needs
// ~2.5 KB of LDS per env).  Instantiated in kernels_shadow_hand.hip and kernels_allegro_hand.hip.
#pragma once
#include "hand_kernels.hpp"

namespace mi {


// pre_physics_step (shadow_hand.py:670-698): hand_pre_env (tasks/hand_task.hpp) for the env of this lane
template <class HT>
__global__ __launch_bounds__(64) void hand_pre_kernel(View v, HandView hv, HandParams p, HandActLimits<HT> al, const float* __restrict__ actions_in,
                                                      unsigned step_counter) {
    const int e = post_env_index<HandSim<typename HT::M>::LANES>(blockIdx.x, threadIdx.x, v.N);   // same env -> XCD mapping as the sub-step kernel
    if (e >= v.N) return;
    hand_pre_env<HT>(v, hv, p, al, actions_in, step_counter, e);
}

// the same on four lanes per env (hand_pre_env_part): 4 N / 64 waves with a quarter of the loads and of the dependency chain each (round 3 / 4: 256
// waves at 16384 envs, 75 % of their cycles waiting).  Real block rb = 8 m + x takes quarter m % 4 of the 64-env block 8 (m / 4) + x, which keeps
// every env on the XCD (x) its sub-step workgroup runs on.
template <class HT>
__global__ __launch_bounds__(64) void hand_pre4_kernel(View v, HandView hv, HandParams p, HandActLimits<HT> al, const float* __restrict__ actions_in,
                                                       unsigned step_counter) {
    const int x = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int vb = 8 * (m >> 2) + x, t = (m & 3) * 16 + ((int)threadIdx.x >> 2);
    const int e = post_env_index<HandSim<typename HT::M>::LANES>(vb, t, v.N);
    if (e >= v.N) return;
    hand_pre_env_part<HT>(v, hv, p, al, actions_in, step_counter, e, (int)threadIdx.x & 3);
}

// gym.refresh_rigid_body_state_tensor (shadow_hand.py:440,456-457) for the five fingertip bodies: ONE THREAD PER (env, fingertip) -- blockIdx.y is
// the fingertip, so a wave walks one chain (wrist + one finger, 6 or 7 hinges) and 5 N / 64 waves fill the chip, where the post kernel's
// one lane per env walked all five chains in turn on 256 waves (latency-bound: 40 of its 61 us at 16384 envs).
template <class HT>
__global__ __launch_bounds__(64) void hand_tips_kernel(View v, HandView hv, HandParams p) {
    MI_NO_CONTRACT
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= v.N) return;
    static_assert(HT::NTIPS == 5 || HT::NTIPS == 0, "five fingertips (ShadowHand) or none (AllegroHand: no fingertip observations)");
    if constexpr (HT::NTIPS == 5) {
        switch (blockIdx.y) {
            case 0: hand_tip<HT, 0>(v, hv, p, e); break;
            case 1: hand_tip<HT, 1>(v, hv, p, e); break;
            case 2: hand_tip<HT, 2>(v, hv, p, e); break;
            case 3: hand_tip<HT, 3>(v, hv, p, e); break;
            default: hand_tip<HT, 4>(v, hv, p, e); break;
        }
    }
}

// cross-env sums of the post step on the device: wave shuffles, one atomic per wave and statistic
struct HandDevRed {
    __device__ __forceinline__ void successes(const HandView& hv, bool valid, long long rs, float succ) const {
        float nres = valid ? (float)rs : 0.f, fin = valid ? succ * (float)rs : 0.f;
        nres = wave_sum(nres); fin = wave_sum(fin);
        if ((threadIdx.x & 63) == 0 && nres > 0.f) { atomicAdd(hv.ws, nres); atomicAdd(hv.ws + 1, fin); }
    }
    __device__ __forceinline__ void episode(const View& v, int e, bool valid, float rew, long long reset, long long progress) const {
        episode_stats(v, e, valid, rew, reset, progress);
    }
};
// post_physics_step (shadow_hand.py:710-715): hand_post_env (tasks/hand_task.hpp), ONE WAVE PER (64 envs, column group) -- blockIdx.y is the
// group (HandCols: dof positions | velocities | joint forces | object / goal + the reward | force-torques | actions | one per fingertip: its wave
// walks the fingertip's chain itself).  Round 3 ran one lane per env over
// all 211 columns: 256 waves at 16384 envs, each with ~150 loads, 4.5 k vector instructions and a 55 KB staging tile (45 us, 61 % of it
// waiting); many more waves with a fraction of the chain each fill the chip's 1024 SIMDs.
template <class HT, int G>
__device__ __forceinline__ void hand_post_group(const View& v, const HandView& hv, const HandParams& p, float* stage) {
    using C = HandCols<HT>;
    constexpr int C0 = C::first(G), NC = C::count(G), LANES = HandSim<typename HT::M>::LANES;
    if constexpr (NC == 0) return;          // (the Allegro hand has no fingertip columns)
    const int N = v.N;
    const int e0 = post_env_index<LANES>(blockIdx.x, threadIdx.x, N);
    const bool valid = e0 < N;
    const int e = valid ? e0 : N - 1;
    // obs_type 0: the vector IS obs_buf; otherwise it goes to full_state and hand_obs_select_kernel picks obs_buf's columns.
    // With asymmetric observations full_state (= states_buf) is written in both cases.
    const bool direct = p.obs_type == 0, to_full = !direct || p.asymmetric_obs != 0;
    // The columns of an env are a row of the row-major obs tensors: written lane by lane, every store instruction of the wave would touch
    // 64 rows (64 cache lines for 4 bytes each).  They are staged in LDS instead, column-major with a pad ([k][65]: the lanes of a column and
    // the columns of a lane both fall on distinct banks), and written out below with consecutive lanes on consecutive columns of a row.
    const HandPostOut o = hand_post_env<HT, G>(v, hv, p, e, valid, [&](int k, float val) MI_LAMBDA { stage[(k - C0) * 65 + (int)threadIdx.x] = val; }, HandDevRed{});
    __syncthreads();
    const bool noisy = direct && v.obs_noise.dist != 0;         // (its own loop: as a select the noise hash was evaluated for every element, +7 us)
    for (int idx = (int)threadIdx.x; idx < 64 * NC; idx += 64) {
        const int row = idx / NC, k = idx - row * NC;
        const int er = post_env_index<LANES>(blockIdx.x, row, N);
        if (er >= N) continue;
        const float val = stage[k * 65 + row];
        if (noisy) hand_store_full_state_elem<HT::NFULL, true>(v, hv, er, C0 + k, val, direct, to_full);
        else hand_store_full_state_elem<HT::NFULL, false>(v, hv, er, C0 + k, val, direct, to_full);
    }
    if constexpr (G == 1) { if (valid) hand_post_store(v, hv, p, e, o); }
}
template <class HT, int G0 = 0>
__device__ __forceinline__ void hand_post_dispatch(const int g, const View& v, const HandView& hv, const HandParams& p, float* stage) {
    if constexpr (G0 < HandCols<HT>::NGROUPS) {
        if (g == G0) hand_post_group<HT, G0>(v, hv, p, stage);          // wave-uniform
        else hand_post_dispatch<HT, G0 + 1>(g, v, hv, p, stage);
    }
}
template <class HT>
__global__ __launch_bounds__(64) void hand_post_kernel(View v, HandView hv, HandParams p) {
    __shared__ float stage[HandCols<HT>::max_count() * 65];
    hand_post_dispatch<HT>((int)blockIdx.y, v, hv, p, stage);
}
// observationType openai / full_no_vel / full (shadow_hand.py:472-526): column subsets of the full state
template <class HT>
__global__ void hand_obs_select_kernel(View v, HandView hv, HandParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int no = p.num_obs;
    if (i >= v.N * no) return;
    const int e = i / no;
    hand_obs_select_elem<HT::NFULL>(v, hv, p, e, i - e * no);
}
template <class HT>
__global__ void hand_finalize_kernel(HandView hv, HandParams p) {
    // (folding this into the post kernel's last block -- a fence and a counter at the end of every block -- was measured at the end of round 3:
    //  the post kernel grew by what this launch costs, 41.3 -> 45.0 us; a one-thread kernel hides in its neighbours' launch shadow)
    if (threadIdx.x == 0 && blockIdx.x == 0) hand_finalize(hv, p);
}

// initial state (hand_init_env)
template <class HT>
__global__ void hand_init_kernel(View v, HandView hv, HandParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.N) return;
    hand_init_env<HT>(v, hv, p, e);
}

// the task's physics sub-steps: which kernel form / object shape runs is the task's business (kernels_<task>.hip)
template <class HT>
hipError_t hand_substeps(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);

template <class HT>
hipError_t launch_step_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const float* actions, int cfi,
                            unsigned step_counter, hipStream_t s) {
    if (hv.pre_parts == 4) {
        const int nvb = (v.N + 63) / 64;
        hipLaunchKernelGGL(hand_pre4_kernel<HT>, dim3(((nvb + 7) / 8) * 32), dim3(64), 0, s, v, hv, p, HandActLimits<HT>::of(p), actions, step_counter);
    } else {
        hipLaunchKernelGGL(hand_pre_kernel<HT>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, hv, p, HandActLimits<HT>::of(p), actions, step_counter);
    }
    hipError_t e = hand_substeps<HT>(v, hv, P, p, cfi * P.substeps, s);
    if (e != hipSuccess) return e;
    // (the fingertip states: by the post kernel's fingertip groups since round 4 -- hand_tips_kernel stays for gym.simulate below and for the A/B)
    if constexpr (HT::NTIPS > 0) { if (hv.tips_in_post == 0) hipLaunchKernelGGL(hand_tips_kernel<HT>, dim3((v.N + 63) / 64, HT::NTIPS), dim3(64), 0, s, v, hv, p); }
    hipLaunchKernelGGL(hand_post_kernel<HT>, dim3((v.N + 63) / 64, HandCols<HT>::NGROUPS), dim3(64), 0, s, v, hv, p);
    if (p.obs_type != 0) hipLaunchKernelGGL(hand_obs_select_kernel<HT>, dim3((v.N * p.num_obs + 255) / 256), dim3(256), 0, s, v, hv, p);
    hipLaunchKernelGGL(hand_finalize_kernel<HT>, dim3(1), dim3(64), 0, s, hv, p);
    return hipGetLastError();
}
template <class HT>
hipError_t launch_simulate_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, hipStream_t s) {
    if (hipError_t e = hand_substeps<HT>(v, hv, P, p, P.substeps, s); e != hipSuccess) return e;
    if constexpr (HT::NTIPS > 0) hipLaunchKernelGGL(hand_tips_kernel<HT>, dim3((v.N + 63) / 64, HT::NTIPS), dim3(64), 0, s, v, hv, p);   // refresh_rigid_body_state_tensor
    return hipGetLastError();
}
template <class HT>
hipError_t launch_init_hand(const View& v, const HandView& hv, const HandParams& p, hipStream_t s) {
    hipLaunchKernelGGL(hand_init_kernel<HT>, dim3((v.N + 127) / 128), dim3(128), 0, s, v, hv, p);
    return hipGetLastError();
}
template <class HT>
__global__ void hand_reset_ids_kernel(View v, HandView hv, HandParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)ids[i];
    if (e < 0 || e >= v.N) return;
    hand_reset_env<HT>(v, hv, p, e, (uint32_t)(v.env_offset + e));
}
template <class HT>
hipError_t launch_reset_hand(const View& v, const HandView& hv, const HandParams& p, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(hand_reset_ids_kernel<HT>, dim3((n + 127) / 128), dim3(128), 0, s, v, hv, p, ids, n);
    return hipGetLastError();
}

}  // namespace mi

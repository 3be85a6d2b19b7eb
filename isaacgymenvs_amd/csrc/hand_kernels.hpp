// hand_kernels.hpp -- what the ShadowHand translation units share: the arena view of the task and the physics sub-step kernel, a template
// over the object's shape.  Each shape is instantiated in its own translation unit (kernels_shadow_hand.hip: block,
// kernels_shadow_hand_pen.hip, kernels_shadow_hand_egg.hip) so that the three ~40 s compilations of this kernel run in parallel.
#pragma once
#include "step_kernels.hpp"
#include "tasks/hand_task.hpp"

namespace mi {

// the ShadowHand's names, as the finger-per-wave form (hand_mw_kernels.hpp, ShadowHand only) uses them
using HM = ModelShadowHand;
using HS = HandSim<HM>;
constexpr int kHandDof = 24, kHandAct = 20, kHandTips = 5, kHandObs = 211;

// gym.simulate(): one physics sub-step of hand + cube
template <class HT, int SHAPE>
__global__ __launch_bounds__(32) void hand_substep_kernel(View v, HandView hv, SimParams P, HandParams p) {
    extern __shared__ float lds_rows[];
    constexpr int LANES = HandSim<typename HT::M>::LANES;
    const int e = blockIdx.x * LANES + threadIdx.x;
    if (e >= v.N) return;
    unsigned long long* ts = nullptr;
#if defined(MI_TIMING)
    ts = (threadIdx.x == 0 && g_mi_tstamp != nullptr) ? g_mi_tstamp + (size_t)blockIdx.x * 16 : nullptr;   // tools/debug/phase_timing_live.py
#endif
    hand_substep_env<HT, SHAPE>(v, hv, P, p, e, RowStore<LANES>{lds_rows + threadIdx.x}, ts);
}

template <class HT, int SHAPE>
inline hipError_t hand_substeps_shape(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
#ifndef MI_HAND_LDS_PAD
#define MI_HAND_LDS_PAD 0      // residency experiments only (tools/hand_residency_ab.sh): extra LDS bytes requested per workgroup
#endif
    using HS = HandSim<typename HT::M>;
    constexpr size_t lds = (size_t)HS::ROW_SLOTS * HS::LANES * sizeof(float) + MI_HAND_LDS_PAD;
    static_assert(lds <= 160 * 1024, "hand row store must fit LDS");
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)hand_substep_kernel<HT, SHAPE>, lds, &configured); e != hipSuccess) return e;
    // (Two workgroups fit a CU.  Asking for more than half the LDS while CUs are spare makes no difference -- the dispatcher spreads
    // the workgroups over the CUs by itself: ShadowHand@8192 0.605 ms either way, round 2 A/B.)
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((hand_substep_kernel<HT, SHAPE>), dim3((v.N + HS::LANES - 1) / HS::LANES), dim3(HS::LANES), lds, s, v, hv, P, p);
    return hipGetLastError();
}

// the finger-per-wave form (hand_mw_kernels.hpp), one translation unit per shape: kernels_shadow_hand_mw.hip, _mw_pen.hip, _mw_egg.hip
hipError_t hand_substeps_mw_box(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// the ShadowHand's kernels on Sim<Scaled<M>> (per-body link-mass factors; kernels_scaled_shadow_hand_*.hip): one-wave and finger-per-wave form
hipError_t hand_substeps_box_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_pen_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_egg_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_box_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_pen_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_egg_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// the same for the Allegro hand (kernels_allegro_hand_mw*.hip)
hipError_t allegro_substeps_mw_box(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t allegro_substeps_mw_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t allegro_substeps_mw_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// defined in kernels_shadow_hand_pen.hip / kernels_shadow_hand_egg.hip
hipError_t hand_substeps_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// the AllegroHand's: kernels_allegro_hand_pen.hip / kernels_allegro_hand_egg.hip
hipError_t allegro_substeps_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t allegro_substeps_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);

}  // namespace mi

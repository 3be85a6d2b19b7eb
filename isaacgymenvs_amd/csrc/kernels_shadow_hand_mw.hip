// kernels_shadow_hand_mw.hip -- the finger-per-wave ShadowHand sub-step (hand_mw_kernels.hpp) instantiated for objectType "block".
#include "hand_mw_kernels.hpp"

namespace mi {
hipError_t hand_substeps_mw_box(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_mw_shape<ShadowHandTask, OBJ_BOX>(v, hv, P, p, n, s);
}
}  // namespace mi

#if defined(MI_TIMING)
extern "C" int mi_debug_set_tstamp_hmw(void* device_buffer) {   // debug builds only (tools/debug/hand_mw_phases.py)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi::g_mi_tstamp_hmw), &device_buffer, sizeof(void*));
}
#endif

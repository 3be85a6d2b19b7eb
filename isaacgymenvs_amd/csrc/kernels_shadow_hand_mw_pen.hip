// kernels_shadow_hand_mw_pen.hip -- the finger-per-wave ShadowHand sub-step (hand_mw_kernels.hpp) instantiated for objectType "pen".
#include "hand_mw_kernels.hpp"

namespace mi {
hipError_t hand_substeps_mw_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_mw_shape<ShadowHandTask, OBJ_CAPSULE>(v, hv, P, p, n, s);
}
}  // namespace mi

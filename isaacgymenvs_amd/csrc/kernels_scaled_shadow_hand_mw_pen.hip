// kernels_scaled_shadow_hand_mw_pen.hip -- the finger-per-wave ShadowHand sub-step on Sim<Scaled<M>> (per-env, per-BODY link-mass factors), objectType
// shape OBJ_CAPSULE; see kernels_scaled_shadow_hand_pen.hip.
#include "hand_mw_kernels.hpp"

namespace mi {
hipError_t hand_substeps_mw_pen_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_mw_shape<ScaledShadowHandTask, OBJ_CAPSULE>(v, hv, P, p, n, s);
}
}  // namespace mi

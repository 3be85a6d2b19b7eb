// kernels_shadow_hand_mw_egg.hip -- the finger-per-wave ShadowHand sub-step (hand_mw_kernels.hpp) instantiated for objectType "egg".
#include "hand_mw_kernels.hpp"

namespace mi {
hipError_t hand_substeps_mw_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_mw_shape<ShadowHandTask, OBJ_ELLIPSOID>(v, hv, P, p, n, s);
}
}  // namespace mi

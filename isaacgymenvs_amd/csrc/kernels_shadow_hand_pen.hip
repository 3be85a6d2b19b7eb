// kernels_shadow_hand_pen.hip -- the ShadowHand physics sub-step instantiated for objectType "pen" (hand_kernels.hpp); its own translation
// unit only so that it compiles in parallel with the block instantiation.
#include "hand_kernels.hpp"

namespace mi {
hipError_t hand_substeps_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_shape<ShadowHandTask, OBJ_CAPSULE>(v, hv, P, p, n, s);
}
}  // namespace mi

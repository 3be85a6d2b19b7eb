// kernels_allegro_hand_egg.hip -- the AllegroHand physics sub-step instantiated for objectType "egg" (hand_kernels.hpp); its own translation
// unit only so that it compiles in parallel with the block instantiation.
#include "hand_kernels.hpp"

namespace mi {
hipError_t allegro_substeps_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_shape<AllegroHandTask, OBJ_ELLIPSOID>(v, hv, P, p, n, s);
}
}  // namespace mi

// kernels_allegro_hand_mw.hip -- the finger-per-wave sub-step (hand_mw_kernels.hpp, core/hand_engine_mw.hpp) instantiated for the Allegro hand,
// objectType "block": four fingers = four waves, the palm's spheres with the first finger's wave; no wrist dofs, no tendons -- the fingers
// couple through the object's six coordinates alone.
#include "hand_mw_kernels.hpp"

namespace mi {
hipError_t allegro_substeps_mw_box(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_mw_shape<AllegroHandTask, OBJ_BOX>(v, hv, P, p, n, s);
}
}  // namespace mi

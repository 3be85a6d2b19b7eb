// kernels_allegro_hand_mw_pen.hip -- the Allegro hand's finger-per-wave sub-step (kernels_allegro_hand_mw.hip) for objectType "pen".
#include "hand_mw_kernels.hpp"

namespace mi {
hipError_t allegro_substeps_mw_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_mw_shape<AllegroHandTask, OBJ_CAPSULE>(v, hv, P, p, n, s);
}
}  // namespace mi

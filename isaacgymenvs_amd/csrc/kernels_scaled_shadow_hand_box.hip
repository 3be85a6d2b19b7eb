// kernels_scaled_shadow_hand_box.hip -- the one-wave ShadowHand physics sub-step on Sim<Scaled<M>> (per-env, per-BODY link-mass factors:
// `actor_params.hand.rigid_body_properties.mass`, reference vec_task.py:783-828), objectType shape OBJ_BOX; picked by hand_substeps<ShadowHandTask>
// while option "hand_body_mass" is on.  Its own translation unit: the plain kernels never see the factor code, and the units compile side by side.
#include "hand_kernels.hpp"

namespace mi {
hipError_t hand_substeps_box_scaled(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    return hand_substeps_shape<ScaledShadowHandTask, OBJ_BOX>(v, hv, P, p, n, s);
}
}  // namespace mi

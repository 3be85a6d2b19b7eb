// kernels_scaled_humanoid.hip -- the Humanoid's one-wave sub-step kernel with the `actor_params` factor tensors compiled in, gfx950.
#include "scaled_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {
template hipError_t launch_substeps_scaled<ModelHumanoid, PlaneGround>(const View&, const SimParams&, const ActParams&, const float*, int, int, int,
                                                                       hipStream_t, const PlaneGround&);
}  // namespace mi

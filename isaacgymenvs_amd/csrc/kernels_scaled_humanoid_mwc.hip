// kernels_scaled_humanoid_mwc.hip -- the Humanoid's limb-per-wave sub-step with the `actor_params` factor tensors compiled in, gfx950.
#include "mwc_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {
template hipError_t launch_substeps_mwc<Scaled<ModelHumanoid>>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t);
}  // namespace mi

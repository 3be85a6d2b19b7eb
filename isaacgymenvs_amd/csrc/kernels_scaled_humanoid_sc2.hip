// kernels_scaled_humanoid_sc2.hip -- the Humanoid's two-wave sub-step with the `actor_params` factor tensors compiled in, gfx950.
#include "sc2_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {
template hipError_t launch_substeps_sc2<Scaled<ModelHumanoid>>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t);
}  // namespace mi

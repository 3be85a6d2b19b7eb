// kernels_humanoid_mwc.hip -- limb-per-wave sub-step of the Humanoid on the compact contact store (core/engine_mwc.hpp), gfx950.
#include "mwc_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {
template hipError_t launch_substeps_mwc<ModelHumanoid>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t);
}  // namespace mi

// kernels_humanoid_sc2.hip -- two-wave sub-step of the Humanoid (self-collision phase on a helper wave), gfx950.  Its own translation
// unit: the kernel takes minutes to compile, like kernels_humanoid.hip next to which it builds in parallel.
#include "sc2_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {
template hipError_t launch_substeps_sc2<ModelHumanoid>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t);
}  // namespace mi

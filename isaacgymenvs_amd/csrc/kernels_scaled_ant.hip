// kernels_scaled_ant.hip -- the Ant's sub-step kernels with the `actor_params` factor tensors compiled in (one-wave and leg-per-wave), gfx950.
#include "scaled_kernels.hpp"
#include "mw_kernels.hpp"
#include "gen/model_ant.h"

namespace mi {
template hipError_t launch_substeps_scaled<ModelAnt, PlaneGround>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t,
                                                                  const PlaneGround&);
template hipError_t launch_substeps_mw<Scaled<ModelAnt>, PlaneGround>(const View&, const SimParams&, const ActParams&, const float*, int, int, int,
                                                                      hipStream_t, const PlaneGround&, int, const MwCmdNormTail*);
}  // namespace mi

// kernels_mw_anymal.hip -- multi-wave sub-step of ANYmal on the height field (AnymalTerrain) and on flat ground (Anymal): one leg per wave, gfx950.
#include "mw_kernels.hpp"
#include "gen/model_anymal.h"

namespace mi {
template hipError_t launch_substeps_mw<ModelAnymal, HeightfieldGround>(const View&, const SimParams&, const ActParams&, const float*, int, int, int,
                                                                       hipStream_t, const HeightfieldGround&, int, const MwCmdNormTail*);
// flat ground, net contact forces reported: the Anymal task (anymal.py)
template hipError_t launch_substeps_mw<ModelAnymal, PlaneGroundNF>(const View&, const SimParams&, const ActParams&, const float*, int, int, int,
                                                                   hipStream_t, const PlaneGroundNF&, int, const MwCmdNormTail*);
}  // namespace mi

// kernels_scaled_anymal.hip -- ANYmal's flat-ground sub-step kernel with the `actor_params` factor tensors compiled in (task Anymal:
// link masses per body, the position drives' gains per dof; reference Anymal.yaml:121-165, vec_task.py:752-828), gfx950.
#include "scaled_kernels.hpp"
#include "gen/model_anymal.h"

namespace mi {
template hipError_t launch_substeps_scaled<ModelAnymal, PlaneGroundNF>(const View&, const SimParams&, const ActParams&, const float*, int, int, int,
                                                                       hipStream_t, const PlaneGroundNF&);
}  // namespace mi

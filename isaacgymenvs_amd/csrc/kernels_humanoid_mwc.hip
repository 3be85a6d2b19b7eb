// kernels_humanoid_mwc.hip -- limb-per-wave sub-step of the Humanoid on the compact contact store (core/engine_mwc.hpp), gfx950.
#include "mwc_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {
template hipError_t launch_substeps_mwc<ModelHumanoid>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t);
template hipError_t launch_substeps_mwc_post<ModelHumanoid, true>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t,
                                                                  const LocoParams&);
}  // namespace mi

#if defined(MI_TIMING)
extern "C" int mi_debug_set_tstamp_mwc(void* device_buffer) {   // debug builds only (tools/debug/mwc_phases.py)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi::g_mi_tstamp_mwc), &device_buffer, sizeof(void*));
}
#endif

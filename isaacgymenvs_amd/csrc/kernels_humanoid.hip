// kernels_humanoid.hip -- ModelHumanoid instantiation of the sub-step / post / reset kernels (gfx950).
#include "step_kernels.hpp"
#include "gen/model_humanoid.h"

namespace mi {

hipError_t launch_step_humanoid(const View& v, const SimParams& P, const LocoParams& tp, const float* actions, int cfi, hipStream_t s) {
    return launch_loco_step<ModelHumanoid, true>(v, P, tp, actions, cfi, s);
}
hipError_t launch_simulate_humanoid(const View& v, const SimParams& P, hipStream_t s) { return launch_simulate<ModelHumanoid>(v, P, s); }
hipError_t launch_reset_humanoid(const View& v, const LocoParams& tp, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL((loco_reset_kernel<ModelHumanoid, true>), dim3((n + 127) / 128), dim3(128), 0, s, v, tp, ids, n);
    return hipGetLastError();
}

}  // namespace mi

#if defined(MI_TIMING)
extern "C" int mi_debug_set_tstamp(void* device_buffer) {   // debug builds only (tools/debug/phase_timing_live.py)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi::g_mi_tstamp), &device_buffer, sizeof(void*));
}
#endif

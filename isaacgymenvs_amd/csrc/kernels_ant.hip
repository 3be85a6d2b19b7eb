// kernels_ant.hip -- ModelAnt instantiation of the sub-step / post / reset kernels (gfx950).
#include "step_kernels.hpp"
#include "gen/model_ant.h"

namespace mi {

hipError_t launch_step_ant(const View& v, const SimParams& P, const LocoParams& tp, const float* actions, int cfi, hipStream_t s) {
    return launch_loco_step<ModelAnt, false>(v, P, tp, actions, cfi, s);
}
hipError_t launch_simulate_ant(const View& v, const SimParams& P, hipStream_t s) { return launch_simulate<ModelAnt>(v, P, s); }
hipError_t launch_reset_ant(const View& v, const LocoParams& tp, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL((loco_reset_kernel<ModelAnt, false>), dim3((n + 127) / 128), dim3(128), 0, s, v, tp, ids, n);
    return hipGetLastError();
}

}  // namespace mi

// kernels_cartpole.hip -- ModelCartpole instantiation of the sub-step / post / reset kernels (gfx950).
#include "step_kernels.hpp"
#include "gen/model_cartpole.h"

namespace mi {

hipError_t launch_step_cartpole(const View& v, const SimParams& P, const CartpoleParams& tp, const float* actions, int cfi, hipStream_t s) {
    ActParams ap{};
    ap.clip = tp.clip_actions; ap.scale = 1.f; ap.nact = 1;
    ap.gear[0] = tp.max_push_effort;  // cartpole.py:159-163: effort on DoF 0 only
    hipError_t e = launch_substeps<ModelCartpole>(v, P, ap, actions, cfi * P.substeps, ACT_FROM_ACTIONS, ACT_STORED_TAU, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cartpole_post_kernel<ModelCartpole>, dim3((v.N + 63) / 64), dim3(64), 0, s, v, tp);
    return hipGetLastError();
}
hipError_t launch_simulate_cartpole(const View& v, const SimParams& P, hipStream_t s) { return launch_simulate<ModelCartpole>(v, P, s); }
hipError_t launch_reset_cartpole(const View& v, const long long* ids, int n, hipStream_t s) {
    hipLaunchKernelGGL(cartpole_reset_kernel<ModelCartpole>, dim3((n + 127) / 128), dim3(128), 0, s, v, ids, n);
    return hipGetLastError();
}

}  // namespace mi

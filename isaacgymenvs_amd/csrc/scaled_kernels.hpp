// scaled_kernels.hpp -- the sub-step launchers on Sim<Scaled<M>>: the instantiation of every physics kernel that multiplies the model's link
// masses / inertias and joint constants by the per-env `actor_params` factors (reference vec_task.py:752-828) and shifts its joint limits.
// Included by the kernels_scaled_<model>*.hip translation units only, so the plain kernels (and their register allocation) never see the
// factor code and the two sets compile side by side.
#pragma once
#include "step_kernels.hpp"

namespace mi {

template <class M, class GND>
hipError_t launch_substeps_scaled(const View& v, const SimParams& P, const ActParams& ap, const float* actions, int n_sub, int first,
                                  int rest, hipStream_t s, const GND& gnd) {
    return launch_substeps<Scaled<M>, GND>(v, P, ap, actions, n_sub, first, rest, s, gnd);
}

}  // namespace mi

// kernels_shadow_hand.hip -- ShadowHand (reference isaacgymenvs/tasks/shadow_hand.py): the task kernels of hand_task_kernels.hpp instantiated
// for the Shadow Hand, the block instantiation of the one-wave physics sub-step, and the choice of the sub-step form.
#include "hand_task_kernels.hpp"

namespace mi {

template <>
hipError_t hand_substeps<ShadowHandTask>(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    if (hv.body_mass != nullptr) {     // option hand_body_mass: per-body link-mass factors -> the Sim<Scaled<M>> instantiations
        if (v.mw != 0) {
            if (p.object_shape == OBJ_ELLIPSOID) return hand_substeps_mw_egg_scaled(v, hv, P, p, n, s);
            if (p.object_shape == OBJ_CAPSULE) return hand_substeps_mw_pen_scaled(v, hv, P, p, n, s);
            return hand_substeps_mw_box_scaled(v, hv, P, p, n, s);
        }
        if (p.object_shape == OBJ_ELLIPSOID) return hand_substeps_egg_scaled(v, hv, P, p, n, s);
        if (p.object_shape == OBJ_CAPSULE) return hand_substeps_pen_scaled(v, hv, P, p, n, s);
        return hand_substeps_box_scaled(v, hv, P, p, n, s);
    }
    if (v.mw != 0) {     // option multi_wave: the finger-per-wave form (core/hand_engine_mw.hpp), block solver order
        if (p.object_shape == OBJ_ELLIPSOID) return hand_substeps_mw_egg(v, hv, P, p, n, s);
        if (p.object_shape == OBJ_CAPSULE) return hand_substeps_mw_pen(v, hv, P, p, n, s);
        return hand_substeps_mw_box(v, hv, P, p, n, s);
    }
    if (p.object_shape == OBJ_ELLIPSOID) return hand_substeps_egg(v, hv, P, p, n, s);
    if (p.object_shape == OBJ_CAPSULE) return hand_substeps_pen(v, hv, P, p, n, s);
    return hand_substeps_shape<ShadowHandTask, OBJ_BOX>(v, hv, P, p, n, s);   // mi_engine_create admits no other shape
}

hipError_t launch_step_shadow_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const float* actions, int cfi,
                                   unsigned step_counter, hipStream_t s) {
    return launch_step_hand<ShadowHandTask>(v, hv, P, p, actions, cfi, step_counter, s);
}
hipError_t launch_simulate_shadow_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, hipStream_t s) {
    return launch_simulate_hand<ShadowHandTask>(v, hv, P, p, s);
}
hipError_t launch_init_shadow_hand(const View& v, const HandView& hv, const HandParams& p, hipStream_t s) { return launch_init_hand<ShadowHandTask>(v, hv, p, s); }
hipError_t launch_reset_shadow_hand(const View& v, const HandView& hv, const HandParams& p, const long long* ids, int n, hipStream_t s) {
    return launch_reset_hand<ShadowHandTask>(v, hv, p, ids, n, s);
}

}  // namespace mi

#if defined(MI_TIMING)
extern "C" int mi_debug_set_tstamp_hand(void* device_buffer) {   // debug builds only (tools/debug/phase_timing_live.py)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi::g_mi_tstamp), &device_buffer, sizeof(void*));
}
#endif

// kernels_allegro_hand.hip -- AllegroHand (reference isaacgymenvs/tasks/allegro_hand.py): the task kernels of hand_task_kernels.hpp instantiated
// for the 16-dof Allegro hand (model from allegro_touch_sensor.urdf, mesh collision shapes sampled by spheres: assets/mesh.py) and the block
// instantiation of the one-wave physics sub-step (core/hand_engine.hpp); the finger-per-wave form lives in kernels_allegro_hand_mw*.hip.
#include "hand_task_kernels.hpp"

namespace mi {

template <>
hipError_t hand_substeps<AllegroHandTask>(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
    if (v.mw != 0) {     // option multi_wave: the finger-per-wave form (core/hand_engine_mw.hpp), block solver order
        if (p.object_shape == OBJ_ELLIPSOID) return allegro_substeps_mw_egg(v, hv, P, p, n, s);
        if (p.object_shape == OBJ_CAPSULE) return allegro_substeps_mw_pen(v, hv, P, p, n, s);
        return allegro_substeps_mw_box(v, hv, P, p, n, s);
    }
    if (p.object_shape == OBJ_ELLIPSOID) return allegro_substeps_egg(v, hv, P, p, n, s);
    if (p.object_shape == OBJ_CAPSULE) return allegro_substeps_pen(v, hv, P, p, n, s);
    return hand_substeps_shape<AllegroHandTask, OBJ_BOX>(v, hv, P, p, n, s);
}

hipError_t launch_step_allegro_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const float* actions, int cfi,
                                    unsigned step_counter, hipStream_t s) {
    return launch_step_hand<AllegroHandTask>(v, hv, P, p, actions, cfi, step_counter, s);
}
hipError_t launch_simulate_allegro_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, hipStream_t s) {
    return launch_simulate_hand<AllegroHandTask>(v, hv, P, p, s);
}
hipError_t launch_init_allegro_hand(const View& v, const HandView& hv, const HandParams& p, hipStream_t s) { return launch_init_hand<AllegroHandTask>(v, hv, p, s); }
hipError_t launch_reset_allegro_hand(const View& v, const HandView& hv, const HandParams& p, const long long* ids, int n, hipStream_t s) {
    return launch_reset_hand<AllegroHandTask>(v, hv, p, ids, n, s);
}

}  // namespace mi

// cpu_articulation.cpp -- CPU backend of the Articulation task: the run-time-compiled robot (gen/model_articulation.h) on the host, OpenMP over envs.
// The one CPU translation unit a run-time variant recompiles (assets/runtime.py).  Not the oracle.
#include "cpu_engine.hpp"
#include "../gen/model_articulation.h"
#include "../tasks/articulation.hpp"

using AM = ModelArticulation;
static_assert(sizeof(MiArticulationParams) == sizeof(ArticulationParams), "MiArticulationParams layout");

ArticulationMeta mi_articulation_meta() { return ArticulationMeta{AM::ND, AM::NB, AM::NSENS, AM::NSPH, AM::FIXED}; }

int cpu_articulation_reset(MiEngine* e, const int64_t* ids, int n) {
    const ArticulationParams& p = *reinterpret_cast<const ArticulationParams*>(e->artic);
    if (!ids) { for (int en = 0; en < e->v.N; ++en) articulation_reset_env<AM>(e->v, p, en); return 0; }
    for (int i = 0; i < n; ++i) if (ids[i] >= 0 && ids[i] < e->v.N) articulation_reset_env<AM>(e->v, p, (int)ids[i]);
    return 0;
}
template <class M>
static int scene_simulate(MiEngine* e, const ArticulationParams& p) {
    if constexpr (M::FIXED == 1) {
        const View& v = e->v;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
        for (int en = 0; en < v.N; ++en) {
            float rows[SceneRows<M>::value];
            for (int ss = 0; ss < e->P.substeps; ++ss)
                articulation_scene_substep_env<M>(v, e->P, p, en, RowStore<1>{rows}, Strided{v.scene_warm + en, v.N});
        }
        return 0;
    } else {
        return -1;
    }
}
int cpu_articulation_simulate(MiEngine* e) {
    const View& v = e->v;
    const ArticulationParams& p = *reinterpret_cast<const ArticulationParams*>(e->artic);
    if (articulation_has_scene(p)) return scene_simulate<AM>(e, p);     // free / static boxes beside the actor (core/scene_engine.hpp)
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < v.N; ++en) {
        float rows[Sim<AM>::ROW_SLOTS > 0 ? Sim<AM>::ROW_SLOTS : 1];
        for (int ss = 0; ss < e->P.substeps; ++ss) articulation_substep_env<AM>(v, e->P, p, en, RowStore<1>{rows}, false);
    }
    return 0;
}
int cpu_articulation_body_states(MiEngine* e) {
    const View& v = e->v;
    const int N = v.N;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        Sim<AM> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = v.root[k * N + en];
        for (int k = 0; k < AM::ND; ++k) { sim.q[k] = v.dof[k * N + en]; sim.qd[k] = v.dof[(AM::ND + k) * N + en]; }
        sfor<AM::NB>([&](auto B_) {
            constexpr int b = decltype(B_)::value;
            float o[13];
            sim.template body_state<b>(o);
            for (int k = 0; k < 13; ++k) v.body_state[(b * 13 + k) * N + en] = o[k];
        });
    }
    return 0;
}
int cpu_articulation_kinematics(MiEngine* e, float* out_j, float* out_h) {
    const View& v = e->v;
    const int N = v.N;
    constexpr int NV = AM::NV;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        Sim<AM> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = v.root[k * N + en];
        for (int k = 0; k < AM::ND; ++k) { sim.q[k] = v.dof[k * N + en]; sim.qd[k] = v.dof[(AM::ND + k) * N + en]; }
        if (out_j) sfor<AM::NB>([&](auto B_) { constexpr int b = decltype(B_)::value; sim.template body_jacobian<b>(out_j + ((size_t)en * AM::NB + b) * 6 * NV); });
        if (out_h) sim.mass_matrix(e->P, out_h + (size_t)en * NV * NV);
    }
    return 0;
}

// cpu_hand_physics.cpp -- gym.simulate() of the in-hand manipulation tasks on the host: hand_substep_env (csrc/tasks/hand_task.hpp, the body of
// hand_substep_kernel) = HandSim's one-wave sub-step (core/hand_engine.hpp) for every env, OpenMP over envs.  Compiled once per hand
// (-DMI_CPU_HAND=0 ShadowHand, 1 AllegroHand) and object shape (-DMI_CPU_SHAPE=0 block, 1 pen, 2 egg).
#include "cpu_hand.hpp"
#include "../tasks/hand_task.hpp"

#if MI_CPU_HAND == 0
// the host build always runs the Sim<Scaled<M>> instantiation: with option "hand_body_mass" off (HandView::body_mass == nullptr) every factor is
// the constant 1.0f, and x * 1.0f is x -- the results are the plain instantiation's, bit for bit (no second set of six translation units)
using HT = ScaledShadowHandTask;
#else
using HT = AllegroHandTask;
#endif
#define MI_CAT3(a, b, c, d) a##b##c##d
#define MI_FN(h, s) MI_CAT3(cpu_hand, h, _substeps_shape, s)

void MI_FN(MI_CPU_HAND, MI_CPU_SHAPE)(MiEngine* e, int n_sub) {
    using HS = HandSim<HT::M>;
    const View& v = e->v;
    const int N = v.N;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        float rows[HS::ROW_SLOTS];
        for (int ss = 0; ss < n_sub; ++ss) hand_substep_env<HT, MI_CPU_SHAPE>(v, e->hv, e->P, e->hand, en, RowStore<1>{rows});
    }
}

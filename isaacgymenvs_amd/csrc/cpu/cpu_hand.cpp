// cpu_hand.cpp -- CPU backend of the in-hand manipulation tasks (ShadowHand: reference shadow_hand.py, BASELINE config 5; AllegroHand:
// allegro_hand.py): the per-env bodies of the HIP kernels (csrc/tasks/hand_task.hpp, the same source hand_task_kernels.hpp wraps) in OpenMP
// loops over envs; physics in cpu_hand_physics.cpp.  Compiled once per hand (-DMI_CPU_HAND=0 ShadowHand, 1 AllegroHand).  Not the oracle.
#include "cpu_hand.hpp"
#include "../tasks/hand_task.hpp"

#if MI_CPU_HAND == 0
using HT = ShadowHandTask;
#else
using HT = AllegroHandTask;
#endif
#define MI_CAT2(a, b, c) a##b##c
#define MI_FN2(h, name) MI_CAT2(cpu_hand, h, name)
#include <cstring>
#define HFN(name) MI_FN2(MI_CPU_HAND, name)

namespace {
struct HandAcc { double nres = 0, fin = 0; StatAcc st; };
struct HostRed {
    HandAcc* a;
    void successes(const HandView&, bool valid, long long rs, float succ) const { if (valid) { a->nres += (double)rs; a->fin += (double)succ * (double)rs; } }
    void episode(const View& v, int e, bool valid, float rew, long long reset, long long progress) const { if (valid) episode_stats_env(v, e, rew, reset, progress, a->st); }
};
void substeps(MiEngine* e, int n) {
    switch (e->hand.object_shape) {
        case 1: HFN(_substeps_shape1)(e, n); break;
        case 2: HFN(_substeps_shape2)(e, n); break;
        default: HFN(_substeps_shape0)(e, n); break;
    }
}
template <int K>
void tips_all(MiEngine* e) {
    if constexpr (K < HT::NTIPS) {
        const View& v = e->v;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
        for (int en = 0; en < v.N; ++en) hand_tip<HT, K>(v, e->hv, e->hand, en);
        tips_all<K + 1>(e);
    }
}
}  // namespace

int HFN(_init)(MiEngine* e) {
    // the arena was laid out by mi_engine_cpu.cpp's task table: a run-time variant of the model that changes a SIZE (force sensors on a hand that has
    // none, assets/runtime.py) must have recompiled that unit too -- refuse to run against a layout of other sizes instead of writing past a tensor
    for (const MiTensorDesc& d : e->descs) {
        if (!strcmp(d.name, "force_sensor") && d.shape[1] != (HT::M::NSENS > 0 ? HT::M::NSENS : 1)) return -2;
        if (!strcmp(d.name, "dof_state") && d.shape[1] != HT::M::ND) return -2;
    }
    for (int en = 0; en < e->v.N; ++en) hand_init_env<HT>(e->v, e->hv, e->hand, en);
    return 0;
}
int HFN(_step)(MiEngine* e, const float* actions, bool simulate_only) {
    const View& v = e->v;
    const HandView& hv = e->hv;
    const HandParams& p = e->hand;
    const int N = v.N;
    if (!simulate_only) {
        const unsigned step_counter = (unsigned)(e->steps + 1);
        const HandActLimits<HT> al = HandActLimits<HT>::of(p);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
        for (int en = 0; en < N; ++en) hand_pre_env<HT>(v, hv, p, al, actions, step_counter, en);
    }
    substeps(e, (simulate_only ? 1 : e->control_freq_inv) * e->P.substeps);
    tips_all<0>(e);                                              // refresh_rigid_body_state_tensor for the fingertips
    if (simulate_only) return 0;
    const bool direct = p.obs_type == 0, to_full = !direct || p.asymmetric_obs != 0;
    std::vector<HandAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        float full[HT::NFULL];
        const HandPostOut o = hand_post_env<HT>(v, hv, p, en, true, [&](int k, float val) { full[k] = val; }, HostRed{&accs[omp_get_thread_num()]});
        if (direct && v.obs_noise.dist != 0) for (int k = 0; k < HT::NFULL; ++k) hand_store_full_state_elem<HT::NFULL, true>(v, hv, en, k, full[k], direct, to_full);
        else for (int k = 0; k < HT::NFULL; ++k) hand_store_full_state_elem<HT::NFULL, false>(v, hv, en, k, full[k], direct, to_full);
        hand_post_store(v, hv, p, en, o);
    }
    std::vector<StatAcc> st;
    for (const HandAcc& a : accs) { hv.ws[0] += (float)a.nres; hv.ws[1] += (float)a.fin; st.push_back(a.st); }
    flush_stats(v, st);
    if (p.obs_type != 0) {
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
        for (int en = 0; en < N; ++en)
            for (int k = 0; k < p.num_obs; ++k) hand_obs_select_elem<HT::NFULL>(v, hv, p, en, k);
    }
    hand_finalize(hv, p);
    return 0;
}
int HFN(_reset)(MiEngine* e, const int64_t* ids, int n) {
    for (int i = 0; i < n; ++i) {
        const int en = (int)ids[i];
        if (en >= 0 && en < e->v.N) hand_reset_env<HT>(e->v, e->hv, e->hand, en, (uint32_t)(e->v.env_offset + en));
    }
    return 0;
}
int HFN(_body_states)(MiEngine* e) {
    using M = HT::M;
    const View& v = e->v;
    const int N = v.N;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = v.root[k * N + en];
        for (int k = 0; k < M::ND; ++k) { sim.q[k] = v.dof[k * N + en]; sim.qd[k] = v.dof[(M::ND + k) * N + en]; }
        sfor<M::NB>([&](auto B_) {
            constexpr int b = decltype(B_)::value;
            float o[13];
            sim.template body_state<b>(o);
            for (int k = 0; k < 13; ++k) v.body_state[(b * 13 + k) * N + en] = o[k];
        });
    }
    return 0;
}
int HFN(_kinematics)(MiEngine* e, float* out_j, float* out_h) {
    using M = HT::M;
    const View& v = e->v;
    const int N = v.N;
    constexpr int NV = M::NV;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = v.root[k * N + en];
        for (int k = 0; k < M::ND; ++k) { sim.q[k] = v.dof[k * N + en]; sim.qd[k] = v.dof[(M::ND + k) * N + en]; }
        if (out_j) sfor<M::NB>([&](auto B_) { constexpr int b = decltype(B_)::value; sim.template body_jacobian<b>(out_j + ((size_t)en * M::NB + b) * 6 * NV); });
        if (out_h) sim.mass_matrix(e->P, out_h + (size_t)en * NV * NV);
    }
    return 0;
}

// cpu_anymal.cpp -- CPU backend of AnymalTerrain (reference isaacgymenvs/tasks/anymal_terrain.py, BASELINE config 4) and Anymal (anymal.py):
// the per-env bodies of the HIP kernels (csrc/tasks/anymal_step.hpp, the same source kernels_anymal.hip wraps) and the engine's own sub-step
// (core/engine.hpp: Sim<ModelAnymal> on the height field / on the plane, one-wave solver order) in OpenMP loops over envs.  Not the oracle.
#include "cpu_engine.hpp"
#include "../gen/model_anymal.h"
#include "../tasks/anymal_step.hpp"

using M = ModelAnymal;
static_assert(M::ND == kAnymalDof, "anymal dof count");

namespace {

// cross-env sums of the post pass: one accumulator per OpenMP thread, added up in thread order after the loop (the device adds per-wave sums
// with atomics in whatever order the waves finish: same quantities, another fp32 summation order)
struct AnymalAcc {
    double cmdnorm = 0, sums[kAnymalSums] = {0}, cnt = 0, lv = 0;
    StatAcc st;
};
struct HostRed {
    AnymalAcc* a;
    void cmdnorm(const View&, float acc) const { a->cmdnorm += acc; }
    void extras(const View&, const float (&st_sums)[kAnymalSums], float st_cnt, float lv) const {
        if (st_cnt > 0.f) {
            for (int k = 0; k < kAnymalSums; ++k) a->sums[k] += st_sums[k];
            a->cnt += st_cnt;
        }
        a->lv += lv;
    }
    void episode(const View& v, int e, bool valid, float rew, long long reset, long long progress) const {
        if (valid) episode_stats_env(v, e, rew, reset, progress, a->st);
    }
};

HeightfieldGround ground_of(const AnymalTerrainDesc& T) {
    return HeightfieldGround{T.hs, T.rows, T.cols, T.hscale, T.vscale, T.border,
                             T.slope_threshold > 0.f ? T.slope_threshold * T.hscale / T.vscale : 3.0e38f, T.walls};
}

// the PD drive of one sim step (anymal_terrain.py:443-446; anymal.py:203-206 through the dofs' stiffness / damping): what
// efforts_for_substep (step_kernels.hpp, ActParams mode 1) evaluates on the device
template <class S>
inline void pd_torques(const S& sim, float kp0, float kd0, float action_scale, float torque_limit, const float* default_pos, const float* act, float* tau,
                       const float* q_api = nullptr, const float* qd_api = nullptr) {
    // q_api / qd_api: the joint state of the task's last refresh instead of the sim's own (AnymalTerrain's first decimation iteration, View::dof_api)
    for (int k = 0; k < kAnymalDof; ++k) {
        float kp = kp0, kd = kd0;
        if constexpr (S::SCALED) {
            if (sim.actor_scale.p != nullptr) { kp *= sim.actor_scale(S::AS_STIFF + k); kd *= sim.actor_scale(S::AS_DAMP + k); }
        }
        const float qk = q_api ? q_api[k] : sim.q[k], qdk = qd_api ? qd_api[k] : sim.qd[k];
        const float u = kp * (action_scale * act[k] + default_pos[k] - qk) - kd * qdk;
        tau[k] = fminf(fmaxf(u, -torque_limit), torque_limit);
    }
}

// n_pd sim sub-steps with the drive re-evaluated before each, then n_hold more on the last torques (AnymalTerrain: the base class's extra
// simulate(), vec_task.py:379-382); act == nullptr: only the hold part, on the stored efforts (gym.simulate() by itself)
template <class MM, class GND>
void substeps_env(const MiEngine* e, const View& v, int en, const GND& gnd, const float* act, int n_pd, int n_hold, float kp, float kd, float action_scale,
                  float torque_limit, const float* default_pos) {
    const int N = v.N;
    Sim<MM> sim;
    for (int k = 0; k < 13; ++k) sim.root[k] = v.root[k * N + en];
    for (int k = 0; k < M::ND; ++k) { sim.q[k] = v.dof[k * N + en]; sim.qd[k] = v.dof[(M::ND + k) * N + en]; }
    if constexpr (Sim<MM>::SCALED) { if (v.actor_scale) sim.actor_scale = Strided{v.actor_scale + en, N}; }
    const float h = e->P.dt / (float)e->P.substeps;
    float tau[M::NDA], rows[Sim<MM>::ROW_SLOTS > 0 ? Sim<MM>::ROW_SLOTS : 1];
    for (int k = 0; k < M::ND; ++k) tau[k] = v.tau[k * N + en];
    const float mu_env = v.friction ? v.friction[en] : -1.f;
    const bool lagging = v.dof_api != nullptr && n_pd > 0;       // (AnymalTerrain's control step: step_kernels.hpp ActSource ACT_LAG / ACT_SNAP)
    for (int ss = 0; ss < n_pd + n_hold; ++ss) {
        if (ss < n_pd) {
            if (lagging && ss == 0) {
                float qa[M::NDA], qda[M::NDA];
                for (int k = 0; k < M::ND; ++k) { qa[k] = v.dof_api[k * N + en]; qda[k] = v.dof_api[(M::ND + k) * N + en]; }
                pd_torques(sim, kp, kd, action_scale, torque_limit, default_pos, act, tau, qa, qda);
            } else {
                pd_torques(sim, kp, kd, action_scale, torque_limit, default_pos, act, tau);
            }
            for (int k = 0; k < M::ND; ++k) v.tau[k * N + en] = tau[k];
        }
        sim.substep(e->P, tau, h, RowStore<1>{rows}, Strided{v.lamc + en, N}, Strided{v.laml + en, N}, Strided{v.sensor + en, N},
                    Strided{v.dof_force + en, N}, gnd, mu_env, Strided{v.netf + en, N});
        if (lagging && ss == n_pd - 1)      // the task's refresh_dof_state_tensor at the end of its decimation loop
            for (int k = 0; k < M::ND; ++k) { v.dof_api[k * N + en] = sim.q[k]; v.dof_api[(M::ND + k) * N + en] = sim.qd[k]; }
    }
    for (int k = 0; k < 13; ++k) v.root[k * N + en] = sim.root[k];
    for (int k = 0; k < M::ND; ++k) { v.dof[k * N + en] = sim.q[k]; v.dof[(M::ND + k) * N + en] = sim.qd[k]; }
}

void flush(const View& v, const std::vector<AnymalAcc>& accs, bool extras) {
    std::vector<StatAcc> st;
    for (const AnymalAcc& a : accs) {
        st.push_back(a.st);
        if (!extras) continue;
        for (int k = 0; k < kAnymalSums; ++k) v.ep_stats[k] += (float)a.sums[k];
        v.ep_stats[13] += (float)a.cnt;
        v.ep_stats[14] += (float)a.lv;
    }
    flush_stats(v, st);
}

// ------------------------------------------------------------------------------------------------ AnymalTerrain
int terrain_step(MiEngine* e, const float* actions, bool simulate_only) {
    const View& v = e->v;
    const AnymalParams& p = e->anymal;
    const AnymalTerrainDesc& T = e->terrain;
    const int N = v.N;
    const HeightfieldGround gnd = ground_of(T);
    const int n_pd = simulate_only ? 0 : p.decimation * e->P.substeps, n_hold = (simulate_only ? 1 : e->control_freq_inv) * e->P.substeps;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        float act[kAnymalDof] = {0};
        if (!simulate_only)
            for (int k = 0; k < kAnymalDof; ++k) {
                act[k] = fminf(fmaxf(actions[(size_t)en * kAnymalDof + k], -p.clip_actions), p.clip_actions);          // vec_task.py:374
                v.actions[k * N + en] = act[k];
            }
        substeps_env<M>(e, v, en, gnd, act, n_pd, n_hold, p.kp, p.kd, p.action_scale, p.torque_limit, p.default_dof_pos);
    }
    if (simulate_only) return 0;
    const unsigned step_counter = (unsigned)(e->steps + 1);       // common_step_counter is incremented before the push test (anymal_terrain.py:460-462)
    std::vector<AnymalAcc> accs(e->num_threads);
    if (p.curriculum) {
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
        for (int en = 0; en < N; ++en) anymal_cmdnorm_env(v, p, en, HostRed{&accs[omp_get_thread_num()]});
        for (AnymalAcc& a : accs) { v.ep_stats[15] += (float)a.cmdnorm; a.cmdnorm = 0; }
    }
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) anymal_post_env<3 * M::NSPH>(v, p, T, step_counter, en, HostRed{&accs[omp_get_thread_num()]});
    flush(v, accs, true);
    {   // extras["episode"] (:421-425): every slot is read before any is re-zeroed
        float mine[16];
        for (int k = 0; k < 16; ++k) mine[k] = v.ep_stats[k];
        for (int k = 0; k < 16; ++k) anymal_extras_slot(v, p, k, mine[k], mine[13]);
    }
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en)
        for (int k = 0; k < kAnymalHeightPts; ++k) anymal_height_point(v, p, T, step_counter, en, k);
    return 0;
}

void terrain_reset(MiEngine* e, const int64_t* ids, int n) {
    const View& v = e->v;
    const int N = v.N;
    const bool levels = e->anymal.curriculum && e->terrain.hs != nullptr;
    if (levels) {          // torch.norm(self.commands[env_ids, :2]) over the envs of THIS call (:431)
        float acc = 0.f;
        for (int i = 0; i < n; ++i) {
            const int en = (int)ids[i];
            if (en < 0 || en >= N) continue;
            const float cx = v.commands[en], cy = v.commands[N + en];
            acc += cx * cx + cy * cy;
        }
        v.ep_stats[15] = acc;
    }
    for (int i = 0; i < n; ++i) {
        const int en = (int)ids[i];
        if (en >= 0 && en < N) anymal_reset_env<3 * M::NSPH>(v, e->anymal, e->terrain, en);
    }
    if (levels) v.ep_stats[15] = 0.f;
}

// ------------------------------------------------------------------------------------------------ Anymal (flat ground)
template <class MM>
void flat_substeps_all(MiEngine* e, const float* actions, bool simulate_only) {
    const View& v = e->v;
    const AnymalFlatParams& p = e->anymal_flat;
    const int N = v.N;
    // position targets = action_scale * a + default (anymal.py:226-229) held for the control step; the PD drive is evaluated at every physics
    // sub-step like PhysX does.  gym.simulate() alone: the drive keeps tracking the targets of the last step (stored actions).
    const int n_pd = (simulate_only ? 1 : e->control_freq_inv) * e->P.substeps;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        float act[kAnymalDof];
        for (int k = 0; k < kAnymalDof; ++k) {
            if (!simulate_only) {
                act[k] = fminf(fmaxf(actions[(size_t)en * kAnymalDof + k], -p.clip_actions), p.clip_actions);
                v.actions[k * N + en] = act[k];
            } else {
                act[k] = v.actions[k * N + en];
            }
        }
        substeps_env<MM>(e, v, en, PlaneGroundNF{}, act, n_pd, 0, p.kp, p.kd, p.action_scale, p.torque_limit, p.default_dof_pos);
    }
}
int flat_step(MiEngine* e, const float* actions, bool simulate_only) {
    const View& v = e->v;
    if (v.actor_scale != nullptr) flat_substeps_all<Scaled<M>>(e, actions, simulate_only);     // option actor_tensors (kernels_scaled_anymal.hip on the device)
    else flat_substeps_all<M>(e, actions, simulate_only);
    if (simulate_only) return 0;
    std::vector<AnymalAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < v.N; ++en) anymal_flat_post_env<3 * M::NSPH>(v, e->anymal_flat, en, HostRed{&accs[omp_get_thread_num()]});
    flush(v, accs, false);
    return 0;
}

}  // namespace

int cpu_anymal_init(MiEngine* e, std::string* err) {
    const View& v = e->v;
    const int N = v.N;
    if (e->task == T_ANYMAL) {
        if (e->terrain.hs == nullptr) { *err = "mi_engine_init_state: AnymalTerrain needs mi_engine_set_terrain first"; return -1; }
        for (int en = 0; en < N; ++en) anymal_init_env<3 * M::NB>(v, e->anymal, e->terrain, e->max_init_level, en);
        std::vector<int64_t> ids(N);
        for (int i = 0; i < N; ++i) ids[i] = i;
        terrain_reset(e, ids.data(), N);      // the constructor's reset_idx(arange(num_envs)) (anymal_terrain.py:170)
    } else {
        for (int en = 0; en < N; ++en) anymal_flat_init_env<3 * M::NB>(v, e->anymal_flat, en);
        for (int en = 0; en < N; ++en) anymal_flat_reset_env<3 * M::NSPH>(v, e->anymal_flat, en);
    }
    return 0;
}
int cpu_anymal_step(MiEngine* e, const float* actions, bool simulate_only) {
    return e->task == T_ANYMAL ? terrain_step(e, actions, simulate_only) : flat_step(e, actions, simulate_only);
}
int cpu_anymal_reset(MiEngine* e, const int64_t* ids, int n) {
    if (e->task == T_ANYMAL) terrain_reset(e, ids, n);
    else
        for (int i = 0; i < n; ++i) if (ids[i] >= 0 && ids[i] < e->v.N) anymal_flat_reset_env<3 * M::NSPH>(e->v, e->anymal_flat, (int)ids[i]);
    return 0;
}
int cpu_anymal_body_states(MiEngine* e) {
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
int cpu_anymal_kinematics(MiEngine* e, float* out_j, float* out_h) {
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

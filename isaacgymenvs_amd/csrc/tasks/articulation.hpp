// articulation.hpp -- the Articulation task for ONE env, host + device: a run-time-compiled robot on the ground plane, driven through gym.simulate
// only (include/mi_engine.h MiArticulationParams).  No observation / reward code: the task that loads such a robot keeps its own.
#pragma once
#include "../arena.hpp"
#include "../core/engine.hpp"

namespace mi {

struct ArticulationParams {   // mirrors MiArticulationParams (include/mi_engine.h)
    float kp[kMaxDof], kd[kMaxDof];
    float max_angular_velocity;
    float init_root[13];
};

// gym.simulate(): one sub-step of env e.  Efforts from dof_actuation_force, position drives (per-dof gains) towards dof_position_targets.
template <class M, int RS>
MI_HD void articulation_substep_env(const View& v, const SimParams& P, const ArticulationParams& p, const int e, const RowStore<RS> rows, const bool prestaged) {
    constexpr int ND = M::ND;
    const int N = v.N;
    Sim<M> sim;
    sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
    float tau[M::NDA], target[M::NDA], kp[M::NDA], kd[M::NDA];
    sfor<ND>([&](auto K) MI_LAMBDA {
        sim.q[K] = v.dof[K * N + e]; sim.qd[K] = v.dof[(ND + K) * N + e];
        tau[K] = v.tau[K * N + e]; target[K] = v.targets[K * N + e];
        kp[K] = p.kp[K]; kd[K] = p.kd[K];
    });
    Drive drv{0.f, 0.f, target, nullptr};
    drv.kpv = kp; drv.kdv = kd;
    const float h = P.dt / (float)P.substeps;
    sim.substep(P, tau, h, rows, Strided{v.lamc + e, N}, Strided{v.laml + e, N}, Strided{v.sensor + e, N}, Strided{v.dof_force + e, N}, PlaneGroundNF{},
                v.friction ? v.friction[e] : -1.f, Strided{v.netf + e, N}, &drv, prestaged);
    if (!M::FIXED && p.max_angular_velocity > 0.f) {       // asset_options.max_angular_velocity: PhysX clamps the body's angular speed
        const float w2 = sim.root[10] * sim.root[10] + sim.root[11] * sim.root[11] + sim.root[12] * sim.root[12];
        const float lim = p.max_angular_velocity;
        if (w2 > lim * lim) {
            const float sc = lim * MI_RSQ(w2);
            sim.root[10] *= sc; sim.root[11] *= sc; sim.root[12] *= sc;
        }
    }
    sfor<13>([&](auto K) MI_LAMBDA { v.root[K * N + e] = sim.root[K]; });
    sfor<ND>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = sim.q[K]; v.dof[(ND + K) * N + e] = sim.qd[K]; });
}

// the state create_actor + prepare_sim leave: the actor at its start pose, joints at zero (clamped into their limits), targets = joint positions
template <class M>
MI_HD void articulation_reset_env(const View& v, const ArticulationParams& p, const int e) {
    const int N = v.N;
    for (int k = 0; k < 13; ++k) { v.root[k * N + e] = p.init_root[k]; v.init_root[k * N + e] = p.init_root[k]; }
    for (int d = 0; d < M::ND; ++d) {
        float q0 = 0.f;
        if (M::dof_limited[d]) q0 = fminf(fmaxf(0.f, fminf(M::dof_lower[d], M::dof_upper[d])), fmaxf(M::dof_lower[d], M::dof_upper[d]));
        v.dof[d * N + e] = q0; v.dof[(M::ND + d) * N + e] = 0.f;
        v.targets[d * N + e] = q0; v.laml[d * N + e] = 0.f;
    }
    for (int k = 0; k < 3 * M::NSPH; ++k) v.lamc[k * N + e] = 0.f;
    for (int k = 0; k < 3 * M::NB; ++k) v.netf[k * N + e] = 0.f;
    v.progress[e] = 0; v.reset[e] = 0;
}

}  // namespace mi

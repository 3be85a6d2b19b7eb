// articulation.hpp -- the Articulation task for ONE env, host + device: a run-time-compiled robot on the ground plane, driven through gym.simulate
// only (include/mi_engine.h MiArticulationParams).  No observation / reward code: the task that loads such a robot keeps its own.
#pragma once
#include "../arena.hpp"
#include "../core/engine.hpp"
#include "../core/scene_engine.hpp"
#include "../../../include/mi_engine.h"      // MI_SCENE_* sizes

namespace mi {

// one set of scene sizes for the C ABI (include/mi_engine.h), the arena layout, the kernels and the reset code
static_assert(kSceneMaxFree == MI_SCENE_MAX_FREE && kSceneMaxStatic == MI_SCENE_MAX_STATIC, "core/scene_engine.hpp and include/mi_engine.h agree on the scene sizes");

struct ArticulationParams {   // mirrors MiArticulationParams (include/mi_engine.h)
    float kp[kMaxDof], kd[kMaxDof];
    float max_angular_velocity;
    float init_root[13];
    SceneParams scene;        // free / static boxes beside the actor (MiScene); n_free = n_static = 0: the actor alone on the ground plane
    float drive_vmax[kMaxDof];   // scenes: the asset's velocity limit of each dof (<= 0: none): clamp of the solved joint velocity; bound of a position drive's approach speed
};

// gym.simulate() of an env that holds more than the actor (core/scene_engine.hpp): one sub-step of env e.  rows: the row store (device: LDS
// [slot][lane], host: a per-env array); warm: last sub-step's (feature, impulses) per contact slot, read and rewritten (device: staged in LDS by the
// kernel, host: the tensor itself).
static inline bool articulation_has_scene(const ArticulationParams& p) { return p.scene.n_free + p.scene.n_static > 0; }
// floats of the scene form's row store (1 for a floating-base robot, which has no scene form: SceneSim<M> is never instantiated for it)
template <class M, bool F = (M::FIXED == 1)> struct SceneRows { static constexpr int value = 1; };
template <class M> struct SceneRows<M, true> { static constexpr int value = SceneSim<M>::ROW_SLOTS; };
template <class M, int RS>
MI_HD void articulation_scene_substep_env(const View& v, const SimParams& P, const ArticulationParams& p, const int e, const RowStore<RS> rows, const Strided warm) {
    constexpr int ND = M::ND;
    const int N = v.N;
    SceneSim<M> sim;
    sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
    float tau[M::NDA], target[M::NDA], kp[M::NDA], kd[M::NDA], vmax[M::NDA];
    sfor<ND>([&](auto K) MI_LAMBDA {
        sim.q[K] = v.dof[K * N + e]; sim.qd[K] = v.dof[(ND + K) * N + e];
        tau[K] = v.tau[K * N + e]; target[K] = v.targets[K * N + e];
        kp[K] = p.kp[K]; kd[K] = p.kd[K]; vmax[K] = p.drive_vmax[K];
        // a velocity-limited drive: its position error is clamped to the error at which spring and damper balance at the limit speed
        // (kp e = kd vmax), so it approaches its target no faster than vmax and pushes with at most kd vmax
        // (only for a drive that HAS a damper: with kd = 0 the balance error is 0 and the target would collapse onto q -- such a drive is bounded by
        // the clamp of the solved joint velocity alone)
        if (p.drive_vmax[K] > 0.f && kp[K] > 0.f && kd[K] > 0.f) {
            const float emax = p.drive_vmax[K] * kd[K] / kp[K];
            target[K] = sim.q[K] + fminf(fmaxf(target[K] - sim.q[K], -emax), emax);
        }
    });
    const int nf = p.scene.n_free < kSceneMaxFree ? p.scene.n_free : kSceneMaxFree;
    for (int i = 0; i < nf; ++i)
        for (int k = 0; k < 13; ++k) sim.box[i][k] = v.scene[(size_t)(13 * i + k) * N + e];
    Drive drv{0.f, 0.f, target, nullptr};
    drv.kpv = kp; drv.kdv = kd;
    const float h = P.dt / (float)P.substeps;
    int nc = 0;
    sim.substep_scene(P, p.scene, tau, drv, h, rows, Strided{v.laml + e, N}, Strided{v.dof_force + e, N}, &nc, warm, vmax, Strided{v.netf + e, N});
    v.scene_nc[e] = nc & 0xFFFF;
    v.scene_nc[N + e] += nc >> 16;
    sfor<ND>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = sim.q[K]; v.dof[(ND + K) * N + e] = sim.qd[K]; });
    for (int i = 0; i < nf; ++i)
        for (int k = 0; k < 13; ++k) v.scene[(size_t)(13 * i + k) * N + e] = sim.box[i][k];
}

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
    for (int i = 0; i < kSceneMaxFree; ++i)
        for (int k = 0; k < 13; ++k) v.scene[(size_t)(13 * i + k) * N + e] = (i < p.scene.n_free && k < 7) ? p.scene.free_init[i][k] : (k == 6 ? 1.f : 0.f);
    v.scene_nc[e] = 0; v.scene_nc[N + e] = 0;
    for (int k = 0; k < 4 * MI_SCENE_WARM_SLOTS; ++k) v.scene_warm[(size_t)k * N + e] = 0.f;
    v.progress[e] = 0; v.reset[e] = 0;
}

}  // namespace mi

// hand_kernels.hpp -- what the ShadowHand translation units share: the arena view of the task and the physics sub-step kernel, a template
// over the object's shape.  Each shape is instantiated in its own translation unit (kernels_shadow_hand.hip: block,
// kernels_shadow_hand_pen.hip, kernels_shadow_hand_egg.hip) so that the three ~40 s compilations of this kernel run in parallel.
#pragma once
#include "step_kernels.hpp"
#include "core/hand_engine.hpp"
#include "gen/model_shadow_hand.h"
#include "gen/model_allegro_hand.h"
#include "tasks/shadow_hand.hpp"

namespace mi {

// the in-hand manipulation tasks built on HandSim (hand_task_kernels.hpp): model, driven dofs, fingertips with a state / force-torque block in
// the observations, width of compute_full_state's vector
struct ShadowHandTask {    // reference shadow_hand.py (24 dofs, 20 of them driven, 4 fixed tendons; :528-584: 211 columns)
    using M = ModelShadowHand;
    static constexpr int ND = 24, NACT = 20, NTIPS = 5, NFULL = 211;
};
struct AllegroHandTask {   // reference allegro_hand.py (16 dofs, all driven, :233-235; no fingertip / force-sensor columns, :485-507: 88 columns)
    using M = ModelAllegroHand;
    static constexpr int ND = 16, NACT = 16, NTIPS = 0, NFULL = 88;
};
static_assert(ShadowHandTask::M::ND == ShadowHandTask::ND && ShadowHandTask::M::NSENS == ShadowHandTask::NTIPS, "shadow hand model");
static_assert(AllegroHandTask::M::ND == AllegroHandTask::ND && AllegroHandTask::M::NSENS == AllegroHandTask::NTIPS, "allegro hand model");

// the ShadowHand's names, as the finger-per-wave form (hand_mw_kernels.hpp, ShadowHand only) uses them
using HM = ModelShadowHand;
using HS = HandSim<HM>;
constexpr int kHandDof = 24, kHandAct = 20, kHandTips = 5, kHandObs = 211;

// arena view of the task (device pointers, SoA [k][N] unless noted) -- same definition in mi_engine.hip
struct HandView {
    float* cur_targets;    // [24][N]
    float* prev_targets;   // [24][N]
    float* object_state;   // [13][N]  root state of the cube
    float* goal_state;     // [7][N]   goal pose (pos, quat)
    float* fingertip;      // [5*13][N] fingertip body states
    float* successes;      // [N]
    long long* reset_goal; // [N]
    int* goal_count;       // [N] number of goal resets so far (RNG counter)
    float* cons;           // [1] consecutive_successes (shadow_hand.py:795-798)
    float* ws;             // [2] per-step scratch of the cross-env sums
    int* ncontact;         // [N] object contacts of the last sub-step (diagnostic)
    float* full_state;     // [N][211] row-major: compute_full_state's vector when it is not obs_buf itself (states_buf, :584)
    float* obj_force;      // [3][N] world-frame force on the cube during this control step (apply_rigid_body_force_tensors)
    float* rb_force;       // [3][N] rb_forces[:, object] in the object's local frame (:201, 700-708)
    float* force_prob;     // [N] random_force_prob (:198-199)
    float* mu_env;         // [N] per-env hand-object contact friction for actor_params friction randomisation; negative = HandParams.mu
    float* scale;          // [8][N] per-env `actor_params` factors (core/hand_engine.hpp HS_*): hand link masses, joint damping, drive stiffness,
                           //        tendon limit stiffness / damping, object mass, object size; 1 = the model's own values
    float* limit_shift;    // [48][N] per-env shifts of the lower / upper joint limits (`actor_params.hand.dof_properties.lower / upper`)
    int* ndropped;         // [N] contacts refused since init because all KMAX slots of the env were taken (diagnostic; a manifold's 5th+ contact does not count)
};

// gym.simulate(): one physics sub-step of hand + cube
template <class HT, int SHAPE>
__global__ __launch_bounds__(32) void hand_substep_kernel(View v, HandView hv, SimParams P, HandParams p) {
    extern __shared__ float lds_rows[];
    using HS = HandSim<typename HT::M>;
    constexpr int ND = HT::ND, LANES = HS::LANES;
    const int e = blockIdx.x * LANES + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    HS sim;
    sfor<3>([&](auto K) MI_LAMBDA { sim.root[K] = p.hand_pos[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { sim.root[3 + K] = p.hand_quat[K]; });
    sfor<6>([&](auto K) MI_LAMBDA { sim.root[7 + K] = 0.f; });
    float target[ND];
    sfor<ND>([&](auto K) MI_LAMBDA {
        sim.q[K] = v.dof[K * N + e];
        sim.qd[K] = v.dof[(ND + K) * N + e];
        target[K] = hv.cur_targets[K * N + e];
    });
    sfor<3>([&](auto K) MI_LAMBDA { sim.obj.pos[K] = hv.object_state[K * N + e]; sim.obj.vel[K] = hv.object_state[(7 + K) * N + e];
                                    sim.obj.angvel[K] = hv.object_state[(10 + K) * N + e]; });
    sfor<4>([&](auto K) MI_LAMBDA { sim.obj.quat[K] = hv.object_state[(3 + K) * N + e]; });
    ObjectParams OP{p.cube_half, p.cube_mass, p.cube_inertia, p.mu,
                    {hv.obj_force[e], hv.obj_force[N + e], hv.obj_force[2 * N + e]}};
    if constexpr (SHAPE != OBJ_BOX) sfor<3>([&](auto K) MI_LAMBDA { OP.dims[K] = p.object_dims[K]; OP.inertia3[K] = p.object_inertia[K]; });
    const float mu_e = hv.mu_env[e];
    if (mu_e >= 0.f) OP.mu = mu_e;
    OP.randomise(hv.scale[HS_OBJECT_MASS * N + e], hv.scale[HS_OBJECT_SCALE * N + e]);
    sim.actor_scale = Strided{hv.scale + e, N};
    sim.limit_shift = Strided{hv.limit_shift + e, N};
#if defined(MI_TIMING)
    sim.tstamp = (threadIdx.x == 0 && g_mi_tstamp != nullptr) ? g_mi_tstamp + (size_t)blockIdx.x * 16 : nullptr;   // tools/debug/phase_timing_live.py
#endif
    const float h = P.dt / (float)P.substeps;
    int nc = 0;
    sim.template substep_hand<LANES, SHAPE>(P, OP, target, h, RowStore<LANES>{lds_rows + threadIdx.x}, Strided{v.laml + e, N}, Strided{v.sensor + e, N},
                                            Strided{v.dof_force + e, N}, &nc);
    sfor<ND>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = sim.q[K]; v.dof[(ND + K) * N + e] = sim.qd[K]; });
    sfor<3>([&](auto K) MI_LAMBDA { hv.object_state[K * N + e] = sim.obj.pos[K]; hv.object_state[(7 + K) * N + e] = sim.obj.vel[K];
                                    hv.object_state[(10 + K) * N + e] = sim.obj.angvel[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { hv.object_state[(3 + K) * N + e] = sim.obj.quat[K]; });
    hv.ncontact[e] = nc & 0xFFFF;
    if (nc >> 16) hv.ndropped[e] += nc >> 16;
}

template <class HT, int SHAPE>
inline hipError_t hand_substeps_shape(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s) {
#ifndef MI_HAND_LDS_PAD
#define MI_HAND_LDS_PAD 0      // residency experiments only (tools/hand_residency_ab.sh): extra LDS bytes requested per workgroup
#endif
    using HS = HandSim<typename HT::M>;
    constexpr size_t lds = (size_t)HS::ROW_SLOTS * HS::LANES * sizeof(float) + MI_HAND_LDS_PAD;
    static_assert(lds <= 160 * 1024, "hand row store must fit LDS");
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)hand_substep_kernel<HT, SHAPE>, lds, &configured); e != hipSuccess) return e;
    // (Two workgroups fit a CU.  Asking for more than half the LDS while CUs are spare makes no difference -- the dispatcher spreads
    // the workgroups over the CUs by itself: ShadowHand@8192 0.605 ms either way)
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((hand_substep_kernel<HT, SHAPE>), dim3((v.N + HS::LANES - 1) / HS::LANES), dim3(HS::LANES), lds, s, v, hv, P, p);
    return hipGetLastError();
}

// the finger-per-wave form (hand_mw_kernels.hpp), one translation unit per shape: kernels_shadow_hand_mw.hip, _mw_pen.hip, _mw_egg.hip
hipError_t hand_substeps_mw_box(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_mw_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// the same for the Allegro hand (kernels_allegro_hand_mw*.hip)
hipError_t allegro_substeps_mw_box(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t allegro_substeps_mw_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t allegro_substeps_mw_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// defined in kernels_shadow_hand_pen.hip / kernels_shadow_hand_egg.hip
hipError_t hand_substeps_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t hand_substeps_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
// the AllegroHand's: kernels_allegro_hand_pen.hip / kernels_allegro_hand_egg.hip
hipError_t allegro_substeps_pen(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);
hipError_t allegro_substeps_egg(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, int n, hipStream_t s);

}  // namespace mi

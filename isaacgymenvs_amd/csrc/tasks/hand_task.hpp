// hand_task.hpp -- the in-hand manipulation tasks (reference isaacgymenvs/tasks/shadow_hand.py, allegro_hand.py) for ONE env, host + device: what
// the kernels of hand_task_kernels.hpp run per lane and cpu/cpu_hand.cpp runs per loop iteration.  pre_physics_step (deferred resets, actions ->
// position targets, random object forces), one hand + object physics sub-step, the fingertip states, post_physics_step (compute_full_state,
// compute_hand_reward), initial state.  Cross-env sums (consecutive_successes, job statistics) go through a reduction policy RED supplied by
// the caller: red.successes(hv, valid, resets, successes) and red.episode(v, e, valid, rew, reset, progress).
#pragma once
#include "../arena.hpp"
#include "../hand_view.hpp"
#include "../core/hand_engine.hpp"
#include "../gen/model_shadow_hand.h"
#include "../gen/model_allegro_hand.h"
#include "shadow_hand.hpp"

namespace mi {

// the in-hand manipulation tasks built on HandSim: model, driven dofs, fingertips with a state / force-torque block in
// the observations, width of compute_full_state's vector
struct ShadowHandTask {    // reference shadow_hand.py (24 dofs, 20 of them driven, 4 fixed tendons; :528-584: 211 columns)
    using M = ModelShadowHand;
    static constexpr int ND = 24, NACT = 20, NTIPS = 5, NFULL = 211;
};
struct AllegroHandTask {   // reference allegro_hand.py (16 dofs, all driven, :233-235; no fingertip / force-sensor columns, :485-507: 88 columns)
    using M = ModelAllegroHand;
    static constexpr int ND = 16, NACT = 16, NTIPS = 0, NFULL = 88;
};
// the same task on Sim<Scaled<M>>: the instantiation whose tree pass multiplies every link's mass and inertia by its per-env factor
// (HandView::body_mass; core/engine.hpp body_scale).  Separate translation units (kernels_scaled_shadow_hand*.hip), picked by the launchers while
// option "hand_body_mass" is on, so that the plain kernels' register allocation never sees the factor code.
struct ScaledShadowHandTask : ShadowHandTask { using M = Scaled<ModelShadowHand>; };
static_assert(ShadowHandTask::M::ND == ShadowHandTask::ND && ShadowHandTask::M::NSENS == ShadowHandTask::NTIPS, "shadow hand model");
// (the Allegro task observes no fingertip force; a run-time variant of the model may still carry force sensors for gym.acquire_force_sensor_tensor --
//  the reference's dextreme task puts them on the four fingertips, tasks/dextreme/allegro_hand_dextreme.py:264-269)
static_assert(AllegroHandTask::M::ND == AllegroHandTask::ND && AllegroHandTask::NTIPS == 0, "allegro hand model");

MI_HD float hand_u(uint32_t seed, uint32_t genv, uint32_t ep, uint32_t k) { return 2.f * uniform01(seed, genv, ep, k) - 1.f; }
// random_force_prob (:198-199, 642-643): log-uniform in force_prob_range
MI_HD float hand_force_prob(const HandParams& p, float u) {
    MI_NO_CONTRACT
    return expf((logf(p.force_prob_range[0]) - logf(p.force_prob_range[1])) * u + logf(p.force_prob_range[1]));
}

// reset_target_pose (shadow_hand.py:586-602): new random goal orientation
MI_HD void hand_reset_goal(const View& v, const HandView& hv, const HandParams& p, int e, uint32_t genv) {
    const int N = v.N;
    const uint32_t gc = (uint32_t)hv.goal_count[e];
    const float r0 = hand_u(v.seed ^ 0x2545F491u, genv, gc, 0), r1 = hand_u(v.seed ^ 0x2545F491u, genv, gc, 1);
    const float xu[3] = {1.f, 0.f, 0.f}, yu[3] = {0.f, 1.f, 0.f};
    float q[4];
    randomize_rotation(r0, r1, xu, yu, q);
    sfor<3>([&](auto K) MI_LAMBDA { hv.goal_state[K * N + e] = p.goal_init_pos[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { hv.goal_state[(3 + K) * N + e] = q[K]; });
    hv.goal_count[e] = (int)gc + 1;
    hv.reset_goal[e] = 0;
}

// reset_idx (shadow_hand.py:604-668, allegro_hand.py:526-590) for one env, in two parts: everything but the hand's dofs ...
template <class HT>
MI_HD void hand_reset_env_object(const View& v, const HandView& hv, const HandParams& p, int e, uint32_t genv, const uint32_t ep) {
    MI_NO_CONTRACT
    const int N = v.N, ND = HT::ND;
    auto rf = [&](int k) MI_LAMBDA { return hand_u(v.seed, genv, ep, (uint32_t)k); };   // rand_floats[:, k], U(-1, 1)
    hand_reset_goal(v, hv, p, e, genv);
    // object: initial pose + position noise, random rotation, zero velocity
    hv.object_state[0 * N + e] = p.object_init_pos[0] + p.reset_position_noise * rf(0);
    hv.object_state[1 * N + e] = p.object_init_pos[1] + p.reset_position_noise * rf(1);
    hv.object_state[2 * N + e] = p.object_init_pos[2] + p.reset_position_noise * rf(2);
    const float xu[3] = {1.f, 0.f, 0.f}, yu[3] = {0.f, 1.f, 0.f};
    float q[4];
    randomize_rotation(rf(3), rf(4), xu, yu, q);
    if (p.object_shape == OBJ_CAPSULE) {                 // pen: randomize_rotation_pen with rand_angle_y = 0.3 (shadow_hand.py:626-629)
        const float zu[3] = {0.f, 0.f, 1.f};
        randomize_rotation_pen(rf(3), rf(4), 0.3f, xu, yu, zu, q);
    }
    sfor<4>([&](auto K) MI_LAMBDA { hv.object_state[(3 + K) * N + e] = q[K]; });
    sfor<6>([&](auto K) MI_LAMBDA { hv.object_state[(7 + K) * N + e] = 0.f; });
    sfor<3>([&](auto K) MI_LAMBDA { hv.rb_force[K * N + e] = 0.f; hv.obj_force[K * N + e] = 0.f; });       // :616
    hv.force_prob[e] = hand_force_prob(p, uniform01(v.seed, genv, ep, 5 + 2 * ND));                           // :642-643
    v.episode[e] = (int)ep + 1;
    v.progress[e] = 0;
    v.reset[e] = 0;
    hv.successes[e] = 0.f;
}
// ... and one dof: default pose (0) + noise * random point of the joint range (:642-651); returns the position (the new drive target)
template <class HT>
MI_HD float hand_reset_dof(const View& v, const HandView& hv, const HandParams& p, int e, uint32_t genv, const uint32_t ep, const int d, const float lower,
                           const float upper) {
    MI_NO_CONTRACT
    const int N = v.N, ND = HT::ND;
    auto rf = [&](int k) MI_LAMBDA { return hand_u(v.seed, genv, ep, (uint32_t)k); };
    const float delta_max = upper - 0.f, delta_min = lower - 0.f;
    const float rand_delta = delta_min + (delta_max - delta_min) * 0.5f * (rf(5 + d) + 1.f);
    const float pos = 0.f + p.reset_dof_pos_noise * rand_delta;
    v.dof[d * N + e] = pos;
    v.dof[(ND + d) * N + e] = 0.f + p.reset_dof_vel_noise * rf(5 + ND + d);
    hv.prev_targets[d * N + e] = pos;
    hv.cur_targets[d * N + e] = pos;
    v.laml[d * N + e] = 0.f;
    return pos;
}
template <class HT>
MI_HD void hand_reset_env(const View& v, const HandView& hv, const HandParams& p, int e, uint32_t genv) {
    const uint32_t ep = (uint32_t)v.episode[e];
    sfor<HT::ND>([&](auto D) MI_LAMBDA { hand_reset_dof<HT>(v, hv, p, e, genv, ep, (int)D, HT::M::dof_lower[D], HT::M::dof_upper[D]); });
    hand_reset_env_object<HT>(v, hv, p, e, genv, ep);
}

// joint limits of the ACTUATED dofs, in actuator order: looked up on the host (HandParams::actuated is a run-time table; indexing the model's
// constant tables with it on the device would either copy them to scratch or walk a 24-way select per action -- ~900 of this step's ~1000 vector /
// scalar instructions per wave were that walk)
template <class HT> struct HandActLimits {
    static constexpr int NX = HT::ND - HT::NACT, NXA = NX > 0 ? NX : 1;       // dofs without an actuator (the Shadow Hand's four coupled distal joints)
    float lo[HT::NACT], up[HT::NACT];
    int xdof[NXA];
    float xlo[NXA], xup[NXA];
    static HandActLimits of(const HandParams& p) {
        HandActLimits a{};
        bool driven[HT::ND] = {};
        for (int k = 0; k < HT::NACT; ++k) {
            const int d = p.actuated[k];
            a.lo[k] = (d >= 0 && d < HT::ND) ? HT::M::dof_lower[d] : 0.f;
            a.up[k] = (d >= 0 && d < HT::ND) ? HT::M::dof_upper[d] : 0.f;
            if (d >= 0 && d < HT::ND) driven[d] = true;
        }
        for (int d = 0, j = 0; d < HT::ND && j < NX; ++d)
            if (!driven[d]) { a.xdof[j] = d; a.xlo[j] = HT::M::dof_lower[d]; a.xup[j] = HT::M::dof_upper[d]; ++j; }
        return a;
    }
};
// random forces on the object (shadow_hand.py:700-708)
MI_HD void hand_random_force(const View& v, const HandView& hv, const HandParams& p, const unsigned step_counter, const int e, const uint32_t genv) {
    MI_NO_CONTRACT
    const int N = v.N;
    if (p.force_scale > 0.f) {   // random forces on the object (:700-708)
        const float decay = powf(p.force_decay, p.dt / p.force_decay_interval);
        float f[3];
        sfor<3>([&](auto K) MI_LAMBDA { f[K] = hv.rb_force[K * N + e] * decay; });
        const uint32_t sk = step_counter | 0x80000000u, sd = v.seed ^ 0x9E3779B9u;
        if (uniform01(sd, genv, sk, 0) < hv.force_prob[e]) {
            // torch.randn(3) * object mass * force_scale; Box-Muller on the engine's counter-based uniforms
            const float u1 = fmaxf(uniform01(sd, genv, sk, 1), 1e-7f), u2 = uniform01(sd, genv, sk, 2);
            const float u3 = fmaxf(uniform01(sd, genv, sk, 3), 1e-7f), u4 = uniform01(sd, genv, sk, 4);
            const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u3));
            const float k = p.cube_mass * p.force_scale;
            f[0] = r1 * cosf(6.283185307179586f * u2) * k;
            f[1] = r1 * sinf(6.283185307179586f * u2) * k;
            f[2] = r2 * cosf(6.283185307179586f * u4) * k;
        }
        float q[4], fw[3];
        sfor<4>([&](auto K) MI_LAMBDA { q[K] = hv.object_state[(3 + K) * N + e]; });
        quat_rotate_s(q, f, 1.f, fw);                                                      // LOCAL_SPACE -> world at application time
        sfor<3>([&](auto K) MI_LAMBDA { hv.rb_force[K * N + e] = f[K]; hv.obj_force[K * N + e] = fw[K]; });
    }
}

// pre_physics_step (shadow_hand.py:670-698): deferred resets, then actions -> targets
template <class HT>
MI_HD void hand_pre_env(const View& v, const HandView& hv, const HandParams& p, const HandActLimits<HT>& al, const float* __restrict__ actions_in,
                        const unsigned step_counter, const int e) {
    MI_NO_CONTRACT
    const int N = v.N;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    if (v.reset[e] != 0) hand_reset_env<HT>(v, hv, p, e, genv);           // also resets the goal (:615)
    else if (hv.reset_goal[e] != 0) hand_reset_goal(v, hv, p, e, genv);
    float raw_act[HT::NACT];
    sfor<HT::NACT>([&](auto A_) MI_LAMBDA { raw_act[A_] = actions_in[(size_t)e * HT::NACT + A_]; });
    if (v.act_noise.dist != 0)                                                                                     // vec_task.py:371-372 (one block: a real branch)
        sfor<HT::NACT>([&](auto A_) MI_LAMBDA { raw_act[A_] = apply_noise(v.act_noise, v.seed, genv, v.step, 1u, (uint32_t)A_, raw_act[A_]); });
    sfor<HT::NACT>([&](auto A_) MI_LAMBDA {
        constexpr int a = A_;
        const int d = p.actuated[a];
        const float act = fminf(fmaxf(raw_act[a], -p.clip_actions), p.clip_actions);                               // vec_task.py:374
        v.actions[a * N + e] = act;
        const float lo = al.lo[a], up = al.up[a];
        const float prev = hv.prev_targets[d * N + e];
        float t;
        if (p.use_relative_control) {
            t = prev + p.dof_speed_scale * p.dt * act;                                     // :686
        } else {
            t = 0.5f * (act + 1.0f) * (up - lo) + lo;                                      // scale(), torch_jit_utils.py:234-235
            t = p.act_moving_average * t + (1.0f - p.act_moving_average) * prev;            // :692-693
        }
        t = fmaxf(fminf(t, up), lo);                                                       // tensor_clamp
        hv.cur_targets[d * N + e] = t;
        hv.prev_targets[d * N + e] = t;                                                     // :697
    });
    hand_random_force(v, hv, p, step_counter, e, genv);
}

// pre_physics_step on FOUR LANES PER ENV (the device kernel since round 4): lane `part` of an env takes a quarter of the actuators -- their actions
// (consecutive floats of the env's row: the four lanes of an env and the envs of a wave read one contiguous stretch), their dofs' part of a
// deferred reset -- and, lane 0, everything that is not per dof (object, goal, flags, the random force).  Every flag is loaded before any lane
// stores one (one wave, program order).  Which actuators / limits a lane has is a select over the four parts of the host's tables (compile-time
// indices: the tables stay in scalar registers).  Same arithmetic per element as hand_pre_env: bit-identical buffers.
template <class T> MI_HD T hand_sel4(const int part, const T x0, const T x1, const T x2, const T x3) {
    return part == 0 ? x0 : part == 1 ? x1 : part == 2 ? x2 : x3;
}
template <class HT>
MI_HD void hand_pre_env_part(const View& v, const HandView& hv, const HandParams& p, const HandActLimits<HT>& al, const float* __restrict__ actions_in,
                             const unsigned step_counter, const int e, const int part) {
    MI_NO_CONTRACT
    constexpr int NACT = HT::NACT, APP = NACT / 4, NX = HandActLimits<HT>::NX;
    static_assert(NACT % 4 == 0 && NX <= 4, "a quarter of the actuators and at most one undriven dof per lane");
    const int N = v.N;
    const uint32_t genv = (uint32_t)(v.env_offset + e);
    const bool rst = v.reset[e] != 0;
    const bool rgoal = !rst && hv.reset_goal[e] != 0;
    const uint32_t ep = (uint32_t)v.episode[e];
    float raw_act[APP];
    sfor<APP>([&](auto K) MI_LAMBDA { raw_act[K] = actions_in[(size_t)e * NACT + part * APP + K]; });
    if (v.act_noise.dist != 0)
        sfor<APP>([&](auto K) MI_LAMBDA { raw_act[K] = apply_noise(v.act_noise, v.seed, genv, v.step, 1u, (uint32_t)(part * APP + K), raw_act[K]); });
    if constexpr (NX > 0) {
        if (rst && part < NX) {
            const int d = hand_sel4(part, al.xdof[0], al.xdof[NX > 1 ? 1 : 0], al.xdof[NX > 2 ? 2 : 0], al.xdof[NX > 3 ? 3 : 0]);
            const float lo = hand_sel4(part, al.xlo[0], al.xlo[NX > 1 ? 1 : 0], al.xlo[NX > 2 ? 2 : 0], al.xlo[NX > 3 ? 3 : 0]);
            const float up = hand_sel4(part, al.xup[0], al.xup[NX > 1 ? 1 : 0], al.xup[NX > 2 ? 2 : 0], al.xup[NX > 3 ? 3 : 0]);
            hand_reset_dof<HT>(v, hv, p, e, genv, ep, d, lo, up);
        }
    }
    sfor<APP>([&](auto K) MI_LAMBDA {
        constexpr int k = K;
        const int a = part * APP + k;
        const int d = hand_sel4(part, p.actuated[k], p.actuated[APP + k], p.actuated[2 * APP + k], p.actuated[3 * APP + k]);
        const float lo = hand_sel4(part, al.lo[k], al.lo[APP + k], al.lo[2 * APP + k], al.lo[3 * APP + k]);
        const float up = hand_sel4(part, al.up[k], al.up[APP + k], al.up[2 * APP + k], al.up[3 * APP + k]);
        float prev;
        if (rst) prev = hand_reset_dof<HT>(v, hv, p, e, genv, ep, d, lo, up);
        else prev = hv.prev_targets[d * N + e];
        const float act = fminf(fmaxf(raw_act[k], -p.clip_actions), p.clip_actions);                               // vec_task.py:374
        v.actions[a * N + e] = act;
        float t;
        if (p.use_relative_control) {
            t = prev + p.dof_speed_scale * p.dt * act;                                     // :686
        } else {
            t = 0.5f * (act + 1.0f) * (up - lo) + lo;                                      // scale(), torch_jit_utils.py:234-235
            t = p.act_moving_average * t + (1.0f - p.act_moving_average) * prev;            // :692-693
        }
        t = fmaxf(fminf(t, up), lo);                                                       // tensor_clamp
        hv.cur_targets[d * N + e] = t;
        hv.prev_targets[d * N + e] = t;                                                     // :697
    });
    if (part != 0) return;
    if (rst) hand_reset_env_object<HT>(v, hv, p, e, genv, ep);
    else if (rgoal) hand_reset_goal(v, hv, p, e, genv);
    hand_random_force(v, hv, p, step_counter, e, genv);
}

// gym.refresh_rigid_body_state_tensor (shadow_hand.py:440,456-457) for the five fingertip bodies: ONE THREAD PER (env, fingertip) -- blockIdx.y is
// the fingertip, so a wave walks one chain (wrist + one finger, 6 or 7 hinges) and 5 N / 64 waves fill the chip, where the post kernel's
// one lane per env walked all five chains in turn on 256 waves (latency-bound: 40 of its 61 us at 16384 envs).
template <class HT, int K>
MI_HD void hand_tip_state(const View& v, const HandParams& p, const int e, float (&o)[13]) {
    constexpr int ND = HT::ND, tip = HT::M::sens_body[K];
    const int N = v.N;
    HandSim<typename HT::M> sim;
    sfor<3>([&](auto I_) MI_LAMBDA { sim.root[I_] = p.hand_pos[I_]; });
    sfor<4>([&](auto I_) MI_LAMBDA { sim.root[3 + I_] = p.hand_quat[I_]; });
    sfor<ND>([&](auto D) MI_LAMBDA {
        if constexpr (HandSim<typename HT::M>::is_ancestor_or_self(HT::M::dof_body[D], tip)) { sim.q[D] = v.dof[D * N + e]; sim.qd[D] = v.dof[(ND + D) * N + e]; }
    });
    sim.template fingertip_state<K>(o);
}
template <class HT, int K>
MI_HD void hand_tip(const View& v, const HandView& hv, const HandParams& p, const int e) {
    const int N = v.N;
    float o[13];
    hand_tip_state<HT, K>(v, p, e, o);
    sfor<13>([&](auto I_) MI_LAMBDA { hv.fingertip[(K * 13 + I_) * N + e] = o[I_]; });
}
// post_physics_step (shadow_hand.py:710-715): progress++, compute_observations (full_state), compute_reward; the fingertip states come
// from hand_tip.  `emit(k, val)` receives column k of compute_full_state's vector (:528-584); what is returned goes to hand_post_store
// once the caller has written the observation rows out.
struct HandPostOut { float r, succ; long long rs, gr, prog; };
// column groups of the full-state vector: the device post kernel runs one wave per (64 envs, group) -- 8 x the waves of a one-lane-per-env kernel,
// each with an eighth of the loads, of the dependency chain and of the row write-out (hand_task_kernels.hpp); the host runs all of them at once
// (GROUP = -1).  Group 1 also computes the reward.
//   0: dof positions | 4: dof velocities | 5: joint forces | 1: object pose / velocities, goal pose, quaternion difference + compute_hand_reward |
//   3: fingertip force-torques | 2: the actions | 6 + t: the state of fingertip t -- on the device the wave of that group walks the fingertip's chain
//   itself (hand_tip_state: what hand_tips_kernel did in a launch of its own until round 4) and also writes the `fingertip` tensor
template <class HT> struct HandCols {
    static constexpr int ND = HT::ND, O_OBJ = 3 * ND, O_GOAL = O_OBJ + 13, O_TIPS = O_GOAL + 11, O_FT = O_TIPS + 13 * HT::NTIPS, O_ACT = O_FT + 6 * HT::NTIPS;
    static_assert(O_ACT + HT::NACT == HT::NFULL, "full_state width");
    static constexpr int G_TIP0 = 6, NGROUPS = G_TIP0 + HT::NTIPS;
    static constexpr int first(int g) {
        return g == 0 ? 0 : g == 4 ? ND : g == 5 ? 2 * ND : g == 1 ? O_OBJ : g == 3 ? O_FT : g == 2 ? O_ACT : O_TIPS + 13 * (g - G_TIP0);
    }
    static constexpr int count(int g) {
        return (g == 0 || g == 4 || g == 5) ? ND : g == 1 ? O_TIPS - O_OBJ : g == 3 ? 6 * HT::NTIPS : g == 2 ? HT::NACT : 13;
    }
    static constexpr int max_count() { int m = 0; for (int g = 0; g < NGROUPS; ++g) m = count(g) > m ? count(g) : m; return m; }
    static constexpr bool covers() { int n = 0; for (int g = 0; g < NGROUPS; ++g) n += count(g); return n == HT::NFULL; }
    static_assert(covers(), "the groups partition the full-state vector");
};
template <class HT, int GROUP = -1, class EMIT, class RED>
MI_HD HandPostOut hand_post_env(const View& v, const HandView& hv, const HandParams& p, const int e, const bool valid, const EMIT& emit, const RED& red) {
    MI_NO_CONTRACT
    constexpr int ND = HT::ND;
    using C = HandCols<HT>;
    constexpr bool ALL = GROUP < 0;
    constexpr bool GP = ALL || GROUP == 0, GV = ALL || GROUP == 4, GF = ALL || GROUP == 5, G1 = ALL || GROUP == 1, GS = ALL || GROUP == 3, GA = ALL || GROUP == 2;
    const int N = v.N;
    // every input is loaded before the first observation is stored: the stores below may alias these arrays as far as the
    // compiler knows, and a load that has to wait for them is a fully exposed memory round trip for a lone wave
    float q[ND], qd[ND], dff[ND];
    if constexpr (GP) sfor<ND>([&](auto K) MI_LAMBDA { q[K] = v.dof[K * N + e]; });
    if constexpr (GV) sfor<ND>([&](auto K) MI_LAMBDA { qd[K] = v.dof[(ND + K) * N + e]; });
    if constexpr (GF) sfor<ND>([&](auto K) MI_LAMBDA { dff[K] = v.dof_force[K * N + e]; });
    float tips[HT::NTIPS > 0 ? HT::NTIPS : 1][13];
    sfor<HT::NTIPS>([&](auto T_) MI_LAMBDA {
        constexpr int t = T_;
        if constexpr (ALL) sfor<13>([&](auto K) MI_LAMBDA { tips[t][K] = hv.fingertip[(t * 13 + K) * N + e]; });     // (the host ran hand_tip first)
        else if constexpr (GROUP == C::G_TIP0 + t) {
            if (hv.tips_in_post != 0) {
                hand_tip_state<HT, t>(v, p, e, tips[t]);
                if (valid) sfor<13>([&](auto K) MI_LAMBDA { hv.fingertip[(t * 13 + K) * N + e] = tips[t][K]; });
            } else {
                sfor<13>([&](auto K) MI_LAMBDA { tips[t][K] = hv.fingertip[(t * 13 + K) * N + e]; });
            }
        }
    });
    float os[13], gp[7], act[HT::NACT], sns[HT::NTIPS > 0 ? 6 * HT::NTIPS : 1];
    if constexpr (GS) sfor<6 * HT::NTIPS>([&](auto K) MI_LAMBDA { sns[K] = v.sensor[K * N + e]; });
    long long reset_in = 0, reset_goal_in = 0, progress_in = 0;
    float successes_in = 0.f;
    if constexpr (G1) {
        reset_in = v.reset[e]; reset_goal_in = hv.reset_goal[e];
        successes_in = hv.successes[e];
        sfor<13>([&](auto K) MI_LAMBDA { os[K] = hv.object_state[K * N + e]; });
        sfor<7>([&](auto K) MI_LAMBDA { gp[K] = hv.goal_state[K * N + e]; });
        progress_in = v.progress[e] + 1;               // :711
    }
    if constexpr (G1 || GA) sfor<HT::NACT>([&](auto K) MI_LAMBDA { act[K] = v.actions[K * N + e]; });
    // compute_full_state (:528-584)
    // layout (shadow_hand.py:528-584 with 24 dofs and 5 fingertips: 211 columns; allegro_hand.py:485-507 with 16 dofs and none: 88):
    // 3 ND | object pose 7, linvel 3, angvel 3 | goal pose 7, quat diff 4 | 13 NTIPS fingertip states | 6 NTIPS force-torques | actions
    sfor<ND>([&](auto D) MI_LAMBDA {
        constexpr int d = D;
        if constexpr (GP) emit(d, (2.0f * q[d] - HT::M::dof_upper[d] - HT::M::dof_lower[d]) / (HT::M::dof_upper[d] - HT::M::dof_lower[d]));   // unscale
        if constexpr (GV) emit(ND + d, p.vel_obs_scale * qd[d]);
        if constexpr (GF) emit(2 * ND + d, p.force_torque_obs_scale * dff[d]);
    });
    if constexpr (G1) {
        sfor<7>([&](auto K) MI_LAMBDA { emit(C::O_OBJ + K, os[K]); });
        sfor<3>([&](auto K) MI_LAMBDA { emit(C::O_OBJ + 7 + K, os[7 + K]); emit(C::O_OBJ + 10 + K, p.vel_obs_scale * os[10 + K]); });
        sfor<7>([&](auto K) MI_LAMBDA { emit(C::O_GOAL + K, gp[K]); });
        float conj[4], qd4[4];
        quat_conjugate(gp + 3, conj);
        quat_mul(os + 3, conj, qd4);
        sfor<4>([&](auto K) MI_LAMBDA { emit(C::O_GOAL + 7 + K, qd4[K]); });
    }
    sfor<HT::NTIPS>([&](auto T_) MI_LAMBDA {
        if constexpr (ALL || GROUP == C::G_TIP0 + T_) sfor<13>([&](auto K) MI_LAMBDA { emit(C::O_TIPS + T_ * 13 + K, tips[T_][K]); });
    });
    if constexpr (GS) sfor<6 * HT::NTIPS>([&](auto K) MI_LAMBDA { emit(C::O_FT + K, p.force_torque_obs_scale * sns[K]); });
    if constexpr (GA) sfor<HT::NACT>([&](auto K) MI_LAMBDA { emit(C::O_ACT + K, act[K]); });
    // compute_hand_reward (:746-800)
    HandPostOut o{0.f, 0.f, 0, 0, 0};
    if constexpr (G1) {
        hand_reward(p.rew, os, os + 3, gp, gp + 3, act, HT::NACT, reset_in, reset_goal_in, progress_in, successes_in, &o.r, &o.rs, &o.gr, &o.prog, &o.succ);
        red.successes(hv, valid, o.rs, o.succ);
        red.episode(v, e, valid, o.r, o.rs, o.prog);
    }
    return o;
}
// where column k (value `val`) of env er's full-state vector goes: obs_buf + the clamped ring slot when the vector IS the observation
// (`direct`: observationType full_state), states_buf when it is not, or under asymmetric observations (`to_full`).  NOISE: the observation
// noise of the domain randomisation, on obs_buf only, after the reward was computed from the clean state (vec_task.py:397-399).
template <int NFULL, bool NOISE>
MI_HD void hand_store_full_state_elem(const View& v, const HandView& hv, const int er, const int k, const float val, const bool direct, const bool to_full) {
    const int N = v.N;
    float* ob = v.obs + (size_t)er * NFULL;
    float* oc = v.obs_out + ((size_t)v.ring * N + er) * NFULL;
    float* fs = hv.full_state + (size_t)er * NFULL;
    if constexpr (NOISE) {
        const float nv = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + er), v.step, 0u, (uint32_t)k, val);
        ob[k] = nv;
        oc[k] = fminf(fmaxf(nv, -v.clip_obs), v.clip_obs);
        if (to_full) fs[k] = val;
    } else {
        if (direct) { ob[k] = val; oc[k] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs); }
        if (to_full) fs[k] = val;
    }
}
MI_HD void hand_post_store(const View& v, const HandView& hv, const HandParams& p, const int e, const HandPostOut& o) {
    v.rew[e] = o.r;
    v.reset[e] = o.rs;
    hv.reset_goal[e] = o.gr;
    v.progress[e] = o.prog;
    hv.successes[e] = o.succ;
    v.randomize[e] += 1;
    v.timeout[e] = (unsigned char)(((float)o.prog >= p.rew.max_episode_length - 1.f) && (o.rs != 0));      // vec_task.py:394
}
// observationType openai / full_no_vel / full (shadow_hand.py:472-526): column k of obs_buf is a column of the full state
template <int NFULL>
MI_HD void hand_obs_select_elem(const View& v, const HandView& hv, const HandParams& p, const int e, const int k) {
    const int no = p.num_obs;
    float val = hv.full_state[(size_t)e * NFULL + p.obs_map[k]];
    if (v.obs_noise.dist != 0) {   // (a block of its own so that the hash is not evaluated speculatively)
        const float nv = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + e), v.step, 0u, (uint32_t)k, val);
        v.obs[(size_t)e * no + k] = nv;
        v.obs_out[((size_t)v.ring * v.N + e) * no + k] = fminf(fmaxf(nv, -v.clip_obs), v.clip_obs);
        return;
    }
    v.obs[(size_t)e * no + k] = val;
    v.obs_out[((size_t)v.ring * v.N + e) * no + k] = fminf(fmaxf(val, -v.clip_obs), v.clip_obs);
}
// consecutive_successes moving average over the whole batch (shadow_hand.py:792-797); re-zeroes the step's sums
MI_HD void hand_finalize(const HandView& hv, const HandParams& p) {
    const float num_resets = hv.ws[0], finished = hv.ws[1], cs = hv.cons[0];
    hv.cons[0] = (num_resets > 0.f) ? p.rew.av_factor * finished / num_resets + (1.0f - p.rew.av_factor) * cs : cs;
    two_sum_acc(hv.ws[2], hv.ws[4], num_resets); two_sum_acc(hv.ws[3], hv.ws[5], finished);     // cumulative since init (compensated: [4], [5] the low parts): what a multi-GPU job all-reduces (parallel.py TaskExtrasReducer)
    hv.ws[0] = 0.f; hv.ws[1] = 0.f;   // last reader of the step's sums: re-zero them here instead of a memset before every post pass
}

// initial state: buffers as the reference's __init__ leaves them (reset_buf = 1 => everything is reset at the first
// pre_physics_step), hand at its default pose, cube and goal at their initial poses
template <class HT>
MI_HD void hand_init_env(const View& v, const HandView& hv, const HandParams& p, const int e) {
    const int N = v.N;
    for (int d = 0; d < HT::ND; ++d) {
        v.dof[d * N + e] = 0.f; v.dof[(HT::ND + d) * N + e] = 0.f; v.laml[d * N + e] = 0.f; v.dof_force[d * N + e] = 0.f;
        hv.cur_targets[d * N + e] = 0.f; hv.prev_targets[d * N + e] = 0.f;
    }
    for (int k = 0; k < 13; ++k) hv.object_state[k * N + e] = (k < 3) ? p.object_init_pos[k] : (k == 6 ? 1.f : 0.f);
    for (int k = 0; k < 7; ++k) hv.goal_state[k * N + e] = (k < 3) ? p.goal_init_pos[k] : (k == 6 ? 1.f : 0.f);
    for (int k = 0; k < 13; ++k) v.root[k * N + e] = (k < 3) ? p.hand_pos[k] : (k < 7 ? p.hand_quat[k - 3] : 0.f);
    for (int k = 0; k < 6 * HT::NTIPS; ++k) v.sensor[k * N + e] = 0.f;
    for (int k = 0; k < 13 * HT::NTIPS; ++k) hv.fingertip[k * N + e] = 0.f;
    for (int k = 0; k < HT::NACT; ++k) v.actions[k * N + e] = 0.f;
    const int no = p.num_obs;
    for (int k = 0; k < no; ++k) { v.obs[(size_t)e * no + k] = 0.f; v.obs_out[(size_t)e * no + k] = 0.f; v.obs_out[((size_t)N + e) * no + k] = 0.f; }
    for (int k = 0; k < HT::NFULL; ++k) hv.full_state[(size_t)e * HT::NFULL + k] = 0.f;
    for (int k = 0; k < 3; ++k) { hv.obj_force[k * N + e] = 0.f; hv.rb_force[k * N + e] = 0.f; }
    hv.force_prob[e] = hand_force_prob(p, uniform01(v.seed ^ 0x51ED27u, (uint32_t)(v.env_offset + e), 0u, 0u));
    hv.mu_env[e] = -1.f;
    for (int k = 0; k < HS_COLUMNS; ++k) hv.scale[k * N + e] = 1.f;
    for (int k = 0; k < HT::M::NB; ++k) hv.body_mass_arena[k * N + e] = 1.f;
    for (int k = 0; k < 2 * HT::ND; ++k) hv.limit_shift[k * N + e] = 0.f;
    hv.successes[e] = 0.f; hv.reset_goal[e] = 1; hv.goal_count[e] = 0; hv.ncontact[e] = 0; hv.ndropped[e] = 0;
    for (int k = 0; k < 4; ++k) hv.npair[k * N + e] = 0;
    v.rew[e] = 0.f; v.reset[e] = 1; v.progress[e] = 0; v.randomize[e] = 0; v.timeout[e] = 0; v.episode[e] = 0; v.ep_ret[e] = 0.f;
    if (e == 0) { hv.cons[0] = 0.f; for (int k = 0; k < 8; ++k) hv.ws[k] = 0.f; for (int k = 0; k < 8; ++k) v.stats[k] = 0.f; }
}

// gym.simulate(): one physics sub-step of hand + object for env e; `rows`: the env's row store (LDS [slot][lane] on the device, a plain
// array on the host)
template <class HT, int SHAPE, int RS>
MI_HD void hand_substep_env(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const int e, const RowStore<RS> rows,
                            unsigned long long* tstamp = nullptr) {
    using HS = HandSim<typename HT::M>;
    constexpr int ND = HT::ND;
    const int N = v.N;
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
    sim.drive_clamp = hv.drive_clamp;
    sim.pair_k = hv.pair_k;
    if constexpr (is_scaled<typename HT::M>::value) { if (hv.body_mass != nullptr) sim.body_mass = Strided{hv.body_mass + e, N}; }
#if defined(MI_TIMING)
    sim.tstamp = tstamp;
#else
    (void)tstamp;
#endif
    const float h = P.dt / (float)P.substeps;
    int nc = 0;
    sim.template substep_hand<RS, SHAPE>(P, OP, target, h, rows, Strided{v.laml + e, N}, Strided{v.sensor + e, N},
                                         Strided{v.dof_force + e, N}, &nc);
    sfor<ND>([&](auto K) MI_LAMBDA { v.dof[K * N + e] = sim.q[K]; v.dof[(ND + K) * N + e] = sim.qd[K]; });
    sfor<3>([&](auto K) MI_LAMBDA { hv.object_state[K * N + e] = sim.obj.pos[K]; hv.object_state[(7 + K) * N + e] = sim.obj.vel[K];
                                    hv.object_state[(10 + K) * N + e] = sim.obj.angvel[K]; });
    sfor<4>([&](auto K) MI_LAMBDA { hv.object_state[(3 + K) * N + e] = sim.obj.quat[K]; });
    hv.ncontact[e] = nc & 0xFFFF;
    if (nc >> 16) hv.ndropped[e] += nc >> 16;
    hv.npair[e] = sim.pair_active;
}

}  // namespace mi

// task_views.hpp -- arena views of the small tasks' extra tensors (device or host pointers, SoA [k][N]) and where they sit in the arena.
// Shared by the HIP library (mi_engine.hip, kernels_<task>.hip) and the CPU backend (cpu/mi_engine_cpu.cpp): one definition each.
#pragma once
#include "arena_layout.hpp"
#include "hand_view.hpp"

namespace mi {

struct QuadView {
    float* targets;        // [8][N] dof_position_targets
    float* thrusts;        // [4][N]
    float* forces;         // [27][N] forces[:, body, xyz] as the reference keeps them (rotor bodies' z = thrust)
};
struct IngenuityView {
    float* thrusts;        // [2][3][N]
    float* forces;         // [6][3][N]  forces[:, body, xyz] as the reference keeps them (bodies 1 and 3 carry the thrusts)
    float* target;         // [3][N]     target_root_positions
    float* marker;         // [13][N]    root state of the marker actor (vec_root_tensor[:, 1, :])
};
struct BbotView {
    float* targets;        // [6][N]  dof_position_targets
    float* ball;           // [13][N] root state of the ball actor (vec_root_tensor[:, 1, :])
    float* lamp;           // [9][N]  attractor impulses (warm start)
    int* ncontact;         // [N]     1 while the ball touches the tray
};

}  // namespace mi

// Quadcopter extras (quadcopter.py:90-97)
static inline void build_quad_layout(int N, Layout& L, QuadView* qv, char* base) {
    const int64_t n = N;
    auto P = [&](size_t o) { return base ? base + o : (char*)nullptr; };
    size_t o;
    o = L.add("dof_position_targets", MI_F32, {n, 8}, {1, n}, 8 * n); if (qv) qv->targets = (float*)P(o);
    o = L.add("thrusts", MI_F32, {n, 4}, {1, n}, 4 * n); if (qv) qv->thrusts = (float*)P(o);
    o = L.add("forces", MI_F32, {n, 9, 3}, {1, 3 * n, n}, 27 * n); if (qv) qv->forces = (float*)P(o);
    L.off = (L.off + 255) & ~size_t(255);
}
// Ingenuity extras (ingenuity.py:63-97): the marker actor's root state is the second row of the reference's [N, 2, 13] root tensor
static inline void build_ingenuity_layout(int N, Layout& L, IngenuityView* iv, char* base) {
    const int64_t n = N;
    auto P = [&](size_t o) { return base ? base + o : (char*)nullptr; };
    size_t o;
    o = L.add("thrusts", MI_F32, {n, 2, 3}, {1, 3 * n, n}, 6 * n); if (iv) iv->thrusts = (float*)P(o);
    o = L.add("forces", MI_F32, {n, 6, 3}, {1, 3 * n, n}, 18 * n); if (iv) iv->forces = (float*)P(o);
    o = L.add("target_root_positions", MI_F32, {n, 3}, {1, n}, 3 * n); if (iv) iv->target = (float*)P(o);
    o = L.add("marker_states", MI_F32, {n, 13}, {1, n}, 13 * n); if (iv) iv->marker = (float*)P(o);
    L.off = (L.off + 255) & ~size_t(255);
}
// BallBalance extras (ball_balance.py:88-112): the ball actor's root state is the second row of the reference's [N, 2, 13] root tensor
static inline void build_bbot_layout(int N, Layout& L, BbotView* bv, char* base) {
    const int64_t n = N;
    auto P = [&](size_t o) { return base ? base + o : (char*)nullptr; };
    size_t o;
    o = L.add("dof_position_targets", MI_F32, {n, 6}, {1, n}, 6 * n); if (bv) bv->targets = (float*)P(o);
    o = L.add("ball_states", MI_F32, {n, 13}, {1, n}, 13 * n); if (bv) bv->ball = (float*)P(o);
    o = L.add("attractor_impulse", MI_F32, {n, 3, 3}, {1, 3 * n, n}, 9 * n); if (bv) bv->lamp = (float*)P(o);
    o = L.add("ball_contact_count", MI_I32, {n}, {1}, n); if (bv) bv->ncontact = (int*)P(o);
    L.off = (L.off + 255) & ~size_t(255);
}
// ShadowHand extras (shadow_hand.py:150-222): object / goal root states, targets, fingertip body states, success counters
// (AllegroHand, allegro_hand.py:142-203: the same tensors for 16 dofs, no fingertips, an 88-wide full state)
static inline void build_hand_layout(int task, int N, Layout& L, HandView* hv, char* base) {
    const int64_t n = N, nd = kTasks[task].nd, nt = kTasks[task].nsens, nfull = kTasks[task].nobs;
    auto P = [&](size_t o) { return base ? base + o : (char*)nullptr; };
    size_t o;
    o = L.add("cur_targets", MI_F32, {n, nd}, {1, n}, nd * n); if (hv) hv->cur_targets = (float*)P(o);
    o = L.add("prev_targets", MI_F32, {n, nd}, {1, n}, nd * n); if (hv) hv->prev_targets = (float*)P(o);
    o = L.add("object_state", MI_F32, {n, 13}, {1, n}, 13 * n); if (hv) hv->object_state = (float*)P(o);
    o = L.add("goal_states", MI_F32, {n, 7}, {1, n}, 7 * n); if (hv) hv->goal_state = (float*)P(o);
    o = L.add("fingertip_state", MI_F32, {n, nt > 0 ? nt : 1, 13}, {1, 13 * n, n}, 13 * (nt > 0 ? nt : 1) * n); if (hv) hv->fingertip = (float*)P(o);
    o = L.add("successes", MI_F32, {n}, {1}, n); if (hv) hv->successes = (float*)P(o);
    o = L.add("reset_goal_buf", MI_I64, {n}, {1}, n); if (hv) hv->reset_goal = (long long*)P(o);
    o = L.add("goal_reset_count", MI_I32, {n}, {1}, n); if (hv) hv->goal_count = (int*)P(o);
    o = L.add("consecutive_successes", MI_F32, {1}, {1}, 1); if (hv) hv->cons = (float*)P(o);
    // [0], [1]: the step's sums for the consecutive-successes average (resets, successes of the envs that reset); [2], [3]: the same, cumulative,
    // [4], [5] the low parts of those two compensated sums
    o = L.add("reward_workspace", MI_F32, {8}, {1}, 8); if (hv) hv->ws = (float*)P(o);
    o = L.add("object_contact_count", MI_I32, {n}, {1}, n); if (hv) hv->ncontact = (int*)P(o);
    o = L.add("states_buf", MI_F32, {n, nfull}, {nfull, 1}, nfull * n); if (hv) hv->full_state = (float*)P(o);
    o = L.add("object_force", MI_F32, {n, 3}, {1, n}, 3 * n); if (hv) hv->obj_force = (float*)P(o);
    o = L.add("rb_forces_object", MI_F32, {n, 3}, {1, n}, 3 * n); if (hv) hv->rb_force = (float*)P(o);
    o = L.add("random_force_prob", MI_F32, {n}, {1}, n); if (hv) hv->force_prob = (float*)P(o);
    o = L.add("friction", MI_F32, {n}, {1}, n); if (hv) hv->mu_env = (float*)P(o);   // hand-object contact friction per env (negative: HandParams.mu)
    // `actor_params` factors of the hand and the object (core/hand_engine.hpp HS_*), 1 = the model's own values
    o = L.add("actor_scale", MI_F32, {n, 8}, {1, n}, 8 * n); if (hv) hv->scale = (float*)P(o);
    // `actor_params.hand.dof_properties.lower / upper`: shifts of the 24 lower, then the 24 upper joint limits of each env's hand
    o = L.add("dof_limit_shift", MI_F32, {n, 2 * nd}, {1, n}, 2 * nd * n); if (hv) hv->limit_shift = (float*)P(o);
    o = L.add("object_contact_dropped", MI_I32, {n}, {1}, n); if (hv) hv->ndropped = (int*)P(o);   // contacts refused since init: all KMAX slots taken
    // per-BODY factors of the hand's link masses (+ inertias) for `actor_params.hand.rigid_body_properties.mass` (reference vec_task.py:783-828 draws
    // one sample per body); read by the kernels only while option "hand_body_mass" is on (mi_engine_set_option)
    o = L.add("hand_body_mass_scale", MI_F32, {n, (int64_t)kTasks[task].nb}, {1, n}, (int64_t)kTasks[task].nb * n); if (hv) { hv->body_mass_arena = (float*)P(o); hv->body_mass = nullptr; }
    // sides of the asset's hand-to-hand contact pairs (MJCF <contact><pair>) pushed in the last sub-step, per role wave of the finger-per-wave form
    o = L.add("hand_pair_count", MI_I32, {n, 4}, {1, n}, 4 * n); if (hv) hv->npair = (int*)P(o);
    L.off = (L.off + 255) & ~size_t(255);
}


// task_views.hpp -- arena views of the small tasks' extra tensors (device or host pointers, SoA [k][N]) and where they sit in the arena.
// Shared by the HIP library (mi_engine.hip, kernels_<task>.hip) and the CPU backend (cpu/mi_engine_cpu.cpp): one definition each.
#pragma once
#include "arena_layout.hpp"

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

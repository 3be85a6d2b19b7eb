// cpu_hand.hpp -- names shared by the CPU backend's hand translation units (cpu_hand.cpp once per hand, cpu_hand_physics.cpp once per hand
// and object shape: the unrolled hand + object sub-step takes g++ about a minute per instantiation, so they compile side by side).
#pragma once
#include "cpu_engine.hpp"

// hand 0: ShadowHand (reference shadow_hand.py), hand 1: AllegroHand (allegro_hand.py); shape 0 block, 1 pen (capsule), 2 egg (ellipsoid)
#define MI_CPU_HAND_DECL(H)                                                   \
    int cpu_hand##H##_init(MiEngine* e);                                      \
    int cpu_hand##H##_step(MiEngine* e, const float* actions, bool simulate_only); \
    int cpu_hand##H##_reset(MiEngine* e, const int64_t* ids, int n);          \
    int cpu_hand##H##_body_states(MiEngine* e);                               \
    int cpu_hand##H##_kinematics(MiEngine* e, float* out_j, float* out_h);    \
    void cpu_hand##H##_substeps_shape0(MiEngine* e, int n_sub);               \
    void cpu_hand##H##_substeps_shape1(MiEngine* e, int n_sub);               \
    void cpu_hand##H##_substeps_shape2(MiEngine* e, int n_sub);
MI_CPU_HAND_DECL(0)
MI_CPU_HAND_DECL(1)

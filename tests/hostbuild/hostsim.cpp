// Host (g++) build of the specialised engine core -- TEST-ONLY.  Lets the CPU test-suite compare the exact
// code the HIP kernels run (csrc/core/engine.hpp + generated model tables) against the independent oracle
// without a GPU.  Never loaded by the product path.
#include <cstring>
// One library per robot model (-DHOSTSIM_<MODEL>): the unrolled cores take minutes to compile, so tests/hostbuild/hostsim.py
// builds the models in parallel and routes each call to the library that holds its model.
#include "../../isaacgymenvs_amd/csrc/core/engine.hpp"
#include "../../isaacgymenvs_amd/csrc/core/engine_mw.hpp"
#include "../../isaacgymenvs_amd/csrc/core/engine_mwc.hpp"
#include <pthread.h>
#include <thread>
#include <vector>
#ifdef HOSTSIM_CARTPOLE
#include "../../isaacgymenvs_amd/csrc/gen/model_cartpole.h"
#endif
#ifdef HOSTSIM_ANT
#include "../../isaacgymenvs_amd/csrc/gen/model_ant.h"
#endif
#ifdef HOSTSIM_ANYMAL
#include "../../isaacgymenvs_amd/csrc/gen/model_anymal.h"
#endif
#ifdef HOSTSIM_QUADCOPTER
#include "../../isaacgymenvs_amd/csrc/gen/model_quadcopter.h"
#endif
#ifdef HOSTSIM_HAND
#include "../../isaacgymenvs_amd/csrc/core/hand_engine.hpp"
#include "../../isaacgymenvs_amd/csrc/core/hand_engine_mw.hpp"
#include "../../isaacgymenvs_amd/csrc/gen/model_shadow_hand.h"
#endif
#ifdef HOSTSIM_HUMANOID
#include "../../isaacgymenvs_amd/csrc/gen/model_humanoid.h"
#endif
#ifdef HOSTSIM_BBOT
#include "../../isaacgymenvs_amd/csrc/core/bbot_engine.hpp"
#include "../../isaacgymenvs_amd/csrc/gen/model_balance_bot.h"
#endif

using namespace mi;

// state/out layouts identical to oracle/physics.c (AoS per env)
// selfcol != 0: per-env state additionally carries lamp[3*NPG] after laml, out carries 9 floats per group (first 3: world force on side a), the oracle's layout
template <class M>
static void run(const SimParams* P, int nenv, float* state, const float* tau, float* out, int selfcol = 0) {
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS, NPG = Sim<M>::NPG;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND + (selfcol ? 3 * NPG : 0), os = 6 * NSENS + ND + 3 * NSPH + (selfcol ? 9 * NPG : 0);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        // warm-start impulses and outputs are updated in place (same layout as oracle/physics.c)
        if (selfcol && NPG > 0) {
            float pf[3 * (NPG > 0 ? NPG : 1)];
            sim.step(*P, tau + (size_t)e * ND, s + 13 + 2 * ND, s + 13 + 2 * ND + 3 * NSPH, o, o + 6 * NSENS, s + 13 + 3 * ND + 3 * NSPH, pf);
            for (int g = 0; g < NPG; ++g) for (int k = 0; k < 3; ++k) o[6 * NSENS + ND + 3 * NSPH + 9 * g + k] = pf[3 * g + k];
        } else {
            sim.step(*P, tau + (size_t)e * ND, s + 13 + 2 * ND, s + 13 + 2 * ND + 3 * NSPH, o, o + 6 * NSENS);
        }
        for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        for (int k = 0; k < ND; ++k) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
    }
}

// ---- multi-wave sub-step (core/engine_mw.hpp) on the host: the four roles of an env run as four threads that meet at a
// pthread barrier where the GPU waves meet at s_barrier; the row store + exchange area is one plain array (stride 1).
struct HostBarrier {
    pthread_barrier_t* b;
    void operator()() const { pthread_barrier_wait(b); }
};
template <class M, class GND, int R>
static void mw_thread(const SimParams* P, float* s, const float* tau, float* o, float* rows, pthread_barrier_t* bar, const GND* gnd,
                      float mu_env, float* netf) {
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    using S = SimMW<M>;
    const float h = P->dt / (float)P->substeps;
    for (int ss = 0; ss < P->substeps; ++ss) {
        S sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        pthread_barrier_wait(bar);     // everybody has read the state of the previous sub-step
        sim.template substep_role<R>(*P, tau, h, RowStore<1>{rows}, Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 2 * ND + 3 * NSPH, 1},
                                     Strided{o, 1}, Strided{o + 6 * NSENS, 1}, *gnd, mu_env, Strided{netf, 1}, false, HostBarrier{bar});
        for (int k = 0; k < ND; ++k)
            if (S::role_of_gi(M::OFF + k) == R || (S::trunk_gi(M::OFF + k) && R == M::TRUNK_ROLE)) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
        if (R == M::TRUNK_ROLE) for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        pthread_barrier_wait(bar);     // the new state is complete
    }
}
template <class M, class GND>
static void run_mw(const SimParams* P, int nenv, float* state, const float* tau, float* out, const GND* gnd, const float* mu, float* netf) {
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND, os = 6 * NSENS + ND + 3 * NSPH;
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        float rows[SimMW<M>::MW_SLOTS];
        for (int k = 0; k < SimMW<M>::MW_SLOTS; ++k) rows[k] = 0.f;
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, 4);
        const float* t = tau + (size_t)e * ND;
        float* nf = netf ? netf + (size_t)e * 3 * M::NB : nullptr;
        const float m = mu ? mu[e] : -1.f;
        std::thread t0(mw_thread<M, GND, 0>, P, s, t, o, rows, &bar, gnd, m, nf), t1(mw_thread<M, GND, 1>, P, s, t, o, rows, &bar, gnd, m, nf),
            t2(mw_thread<M, GND, 2>, P, s, t, o, rows, &bar, gnd, m, nf), t3(mw_thread<M, GND, 3>, P, s, t, o, rows, &bar, gnd, m, nf);
        t0.join(); t1.join(); t2.join(); t3.join();
        pthread_barrier_destroy(&bar);
    }
}
// slope correction of the height-field ground for the terrain entries below: slope_threshold * hscale / vscale (raw units), <= 0 = off
static float g_slope_thr = 3.0e38f;
extern "C" void hs_set_slope_threshold(float thr_raw) { g_slope_thr = thr_raw > 0.f ? thr_raw : 3.0e38f; }
static int g_walls = 1;     // risers collide from the side (HeightfieldGround::contact)
extern "C" void hs_set_walls(int on) { g_walls = on; }
// the contact query by itself (tests/test_terrain.py)
extern "C" void hs_ground_contact(const short* hs, int rows, int cols, float hscale, float vscale, float border, float x, float y, float z, float r,
                                  float* dist, float* n) {
    const HeightfieldGround g{hs, rows, cols, hscale, vscale, border, g_slope_thr, g_walls};
    g.contact(x, y, z, r, dist, n);
}
#ifdef HOSTSIM_ANT
extern "C" int hs_step_mw_ant(const SimParams* P, int nenv, float* state, const float* tau, float* out) {
    const PlaneGround g{};
    run_mw<ModelAnt, PlaneGround>(P, nenv, state, tau, out, &g, nullptr, nullptr);
    return 0;
}
#endif
#ifdef HOSTSIM_ANYMAL
extern "C" int hs_step_mw_terrain(const SimParams* P, int nenv, float* state, const float* tau, float* out, const short* hs, int rows,
                                  int cols, float hscale, float vscale, float border, const float* mu, float* netf) {
    const HeightfieldGround g{hs, rows, cols, hscale, vscale, border, g_slope_thr, g_walls};
    run_mw<ModelAnymal, HeightfieldGround>(P, nenv, state, tau, out, &g, mu, netf);
    return 0;
}
#endif

#ifdef HOSTSIM_HUMANOID
extern "C" int hs_step_selfcol(const SimParams* P, int nenv, float* state, const float* tau, float* out) {
    run<ModelHumanoid>(P, nenv, state, tau, out, 1);
    return 0;
}
// the self-colliding sub-step on TWO waves (Sim::substep role 0 / 1): two threads per env share the row store and meet at a pthread
// barrier where the GPU waves meet at s_barrier.  Same state / out layout as hs_step_selfcol.
template <int ROLE>
static void sc2_thread(const SimParams* P, float* s, const float* tau, float* o, float* rows, float* pf, pthread_barrier_t* bar, int nroles) {
    using M = ModelHumanoid;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const float h = P->dt / (float)P->substeps;
    for (int ss = 0; ss < P->substeps; ++ss) {
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        pthread_barrier_wait(bar);     // both have read the state of the previous sub-step
        const SelfCol sc{Strided{s + 13 + 3 * ND + 3 * NSPH, 1}, Strided{pf, 1}};
        sim.substep(*P, tau, h, RowStore<1>{rows}, Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 2 * ND + 3 * NSPH, 1}, Strided{o, 1},
                    Strided{o + 6 * NSENS, 1}, PlaneGround{}, -1.f, Strided{nullptr, 1}, nullptr, false, &sc, ROLE, HostBarrier{bar}, nroles);
        if (ROLE == 0) {
            for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
            for (int k = 0; k < ND; ++k) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
        }
        pthread_barrier_wait(bar);     // the new state is complete
    }
}
static int step_selfcol_waves(const SimParams* P, int nenv, float* state, const float* tau, float* out, int nroles);
extern "C" int hs_step_selfcol2(const SimParams* P, int nenv, float* state, const float* tau, float* out) { return step_selfcol_waves(P, nenv, state, tau, out, 2); }
extern "C" int hs_step_selfcol3(const SimParams* P, int nenv, float* state, const float* tau, float* out) { return step_selfcol_waves(P, nenv, state, tau, out, 3); }
static int step_selfcol_waves(const SimParams* P, int nenv, float* state, const float* tau, float* out, int nroles) {
    using M = ModelHumanoid;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS, NPG = Sim<M>::NPG;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND + 3 * NPG, os = 6 * NSENS + ND + 3 * NSPH + 9 * NPG;
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        static float rows[Sim<M>::ROW_SLOTS];
        float pf[3 * NPG];
        for (int k = 0; k < Sim<M>::ROW_SLOTS; ++k) rows[k] = 0.f;
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, nroles);
        std::thread t0(sc2_thread<0>, P, s, tau + (size_t)e * ND, o, rows, pf, &bar, nroles), t1(sc2_thread<1>, P, s, tau + (size_t)e * ND, o, rows, pf, &bar, nroles);
        if (nroles == 3) { std::thread t2(sc2_thread<2>, P, s, tau + (size_t)e * ND, o, rows, pf, &bar, nroles); t2.join(); }
        t0.join(); t1.join();
        pthread_barrier_destroy(&bar);
        for (int g = 0; g < NPG; ++g) for (int k = 0; k < 3; ++k) o[6 * NSENS + ND + 3 * NSPH + 9 * g + k] = pf[3 * g + k];
    }
    return 0;
}
#endif

#ifdef HOSTSIM_HUMANOID
// the limb-per-wave sub-step on the compact store (core/engine_mwc.hpp): four threads per env = the four role waves, one shared row
// store, a pthread barrier where the GPU waves meet at s_barrier.  State / out layout of hs_step_selfcol; selfcol = 0: the actor
// ignores itself (no lamp / pair outputs touched).
template <int R>
static void mwc_thread(const SimParams* P, float* s, const float* tau, float* o, float* rows, float* pf, int* dropped, pthread_barrier_t* bar, int selfcol) {
    using M = ModelHumanoid;
    using S = SimMWC<M>;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const float h = P->dt / (float)P->substeps;
    for (int ss = 0; ss < P->substeps; ++ss) {
        S sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        pthread_barrier_wait(bar);
        const SelfCol sc{Strided{s + 13 + 3 * ND + 3 * NSPH, 1}, Strided{pf, 1}, dropped, 1};
        if constexpr (S::HAS_PAIR_ROLE && R == S::PAIR_ROLE) {
            sim.substep_pair(*P, h, RowStore<1>{rows}, selfcol ? &sc : nullptr, HostBarrier{bar});
        } else {
            sim.template substep_role_c<R>(*P, tau, h, RowStore<1>{rows}, Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 2 * ND + 3 * NSPH, 1}, Strided{o, 1},
                                           Strided{o + 6 * NSENS, 1}, -1.f, selfcol ? &sc : nullptr, HostBarrier{bar});
        }
        for (int k = 0; k < ND; ++k)
            if (S::MW::role_of_gi(M::OFF + k) == R || (S::MW::trunk_gi(M::OFF + k) && R == M::TRUNK_ROLE)) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
        if (R == M::TRUNK_ROLE) for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        pthread_barrier_wait(bar);
    }
}
// the same with all sub-steps of the step inside ONE call per role (SimMWC::substeps_fused: what substep_mwc_fused_kernel runs per wave): the
// roles keep their state between sub-steps, integrate the trunk redundantly and hand the pair role the new pose through the row store
template <int R>
static void mwc_fused_thread(const SimParams* P, float* s, const float* tau, float* o, float* rows, float* pf, int* dropped, pthread_barrier_t* bar, int selfcol) {
    using M = ModelHumanoid;
    using S = SimMWC<M>;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const float h = P->dt / (float)P->substeps;
    S sim;
    for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
    for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
    pthread_barrier_wait(bar);
    const SelfCol sc{Strided{s + 13 + 3 * ND + 3 * NSPH, 1}, Strided{pf, 1}, dropped, 1};
    sim.template substeps_fused<R>(*P, tau, h, RowStore<1>{rows}, Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 2 * ND + 3 * NSPH, 1}, Strided{o, 1},
                                   Strided{o + 6 * NSENS, 1}, -1.f, selfcol ? &sc : nullptr, HostBarrier{bar}, P->substeps);
    pthread_barrier_wait(bar);
    for (int k = 0; k < ND; ++k)
        if (S::MW::role_of_gi(M::OFF + k) == R || (S::MW::trunk_gi(M::OFF + k) && R == M::TRUNK_ROLE)) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
    if (R == M::TRUNK_ROLE) for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
}
extern "C" int hs_step_mwc_fused(const SimParams* P, int nenv, float* state, const float* tau, float* out, int selfcol, int* dropped_out) {
    using M = ModelHumanoid;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS, NPG = Sim<M>::NPG;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND + 3 * NPG, os = 6 * NSENS + ND + 3 * NSPH + 9 * NPG;
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        static float rows[SimMWC<M>::MWC_SLOTS];
        float pf[3 * NPG];
        int dropped[2] = {0, 0};
        for (int k = 0; k < SimMWC<M>::MWC_SLOTS; ++k) rows[k] = 0.f;
        for (int k = 0; k < 3 * NPG; ++k) pf[k] = 0.f;
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, 4);
        const float* t = tau + (size_t)e * ND;
        std::thread t0(mwc_fused_thread<0>, P, s, t, o, rows, pf, dropped, &bar, selfcol), t1(mwc_fused_thread<1>, P, s, t, o, rows, pf, dropped, &bar, selfcol),
            t2(mwc_fused_thread<2>, P, s, t, o, rows, pf, dropped, &bar, selfcol), t3(mwc_fused_thread<3>, P, s, t, o, rows, pf, dropped, &bar, selfcol);
        t0.join(); t1.join(); t2.join(); t3.join();
        pthread_barrier_destroy(&bar);
        for (int g = 0; g < NPG; ++g) for (int k = 0; k < 3; ++k) o[6 * NSENS + ND + 3 * NSPH + 9 * g + k] = pf[3 * g + k];
        if (dropped_out) { dropped_out[2 * e] = dropped[0]; dropped_out[2 * e + 1] = dropped[1]; }
    }
    return 0;
}
extern "C" int hs_step_mwc(const SimParams* P, int nenv, float* state, const float* tau, float* out, int selfcol, int* dropped_out) {
    using M = ModelHumanoid;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS, NPG = Sim<M>::NPG;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND + 3 * NPG, os = 6 * NSENS + ND + 3 * NSPH + 9 * NPG;
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        static float rows[SimMWC<M>::MWC_SLOTS];
        float pf[3 * NPG];
        int dropped[2] = {0, 0};
        for (int k = 0; k < SimMWC<M>::MWC_SLOTS; ++k) rows[k] = 0.f;
        for (int k = 0; k < 3 * NPG; ++k) pf[k] = 0.f;
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, 4);
        const float* t = tau + (size_t)e * ND;
        std::thread t0(mwc_thread<0>, P, s, t, o, rows, pf, dropped, &bar, selfcol), t1(mwc_thread<1>, P, s, t, o, rows, pf, dropped, &bar, selfcol),
            t2(mwc_thread<2>, P, s, t, o, rows, pf, dropped, &bar, selfcol), t3(mwc_thread<3>, P, s, t, o, rows, pf, dropped, &bar, selfcol);
        t0.join(); t1.join(); t2.join(); t3.join();
        pthread_barrier_destroy(&bar);
        for (int g = 0; g < NPG; ++g) for (int k = 0; k < 3; ++k) o[6 * NSENS + ND + 3 * NSPH + 9 * g + k] = pf[3 * g + k];
        if (dropped_out) { dropped_out[2 * e] = dropped[0]; dropped_out[2 * e + 1] = dropped[1]; }
    }
    return 0;
}
#endif

extern "C" int hs_step(const char* model, const SimParams* P, int nenv, float* state, const float* tau, float* out) {
#ifdef HOSTSIM_CARTPOLE
    if (!strcmp(model, "cartpole")) { run<ModelCartpole>(P, nenv, state, tau, out); return 0; }
#endif
#ifdef HOSTSIM_ANT
    if (!strcmp(model, "ant")) { run<ModelAnt>(P, nenv, state, tau, out); return 0; }
#endif
#ifdef HOSTSIM_HUMANOID
    if (!strcmp(model, "humanoid")) { run<ModelHumanoid>(P, nenv, state, tau, out); return 0; }
#endif
    return -1;
}

#ifdef HOSTSIM_QUADCOPTER
// PD position drives + local-frame forces on the sensor bodies (Quadcopter): target[nenv][ND], fsens[nenv][NSENS][3]
extern "C" int hs_step_drive(const char* model, const SimParams* P, int nenv, float* state, float* out, float kp, float kd,
                             const float* target, const float* fsens) {
    if (strcmp(model, "quadcopter")) return -1;
    using M = ModelQuadcopter;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND, os = 6 * NSENS + ND + 3 * NSPH;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        const Drive drv{kp, kd, target + (size_t)e * ND, fsens ? fsens + (size_t)e * 3 * NSENS : nullptr};
        float tau[ND] = {0}, rows[Sim<M>::ROW_SLOTS > 0 ? Sim<M>::ROW_SLOTS : 1];
        const float h = P->dt / (float)P->substeps;
        for (int it = 0; it < P->substeps; ++it)
            sim.substep(*P, tau, h, RowStore<1>{rows}, Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 2 * ND + 3 * NSPH, 1}, Strided{o, 1},
                        Strided{o + 6 * NSENS, 1}, PlaneGround{}, -1.f, Strided{nullptr, 1}, &drv);
        for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        for (int k = 0; k < ND; ++k) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
    }
    return 0;
}

#endif

#ifdef HOSTSIM_ANYMAL
// height-field variant (AnymalTerrain): per-env friction mu[nenv], net contact forces netf[nenv][3*NB]
extern "C" int hs_step_terrain(const char* model, const SimParams* P, int nenv, float* state, const float* tau, float* out,
                               const short* hs, int rows, int cols, float hscale, float vscale, float border, const float* mu,
                               float* netf) {
    if (strcmp(model, "anymal")) return -1;
    using M = ModelAnymal;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND, os = 6 * NSENS + ND + 3 * NSPH;
    const HeightfieldGround g{hs, rows, cols, hscale, vscale, border, g_slope_thr, g_walls};
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        sim.step_terrain(*P, tau + (size_t)e * ND, s + 13 + 2 * ND, s + 13 + 2 * ND + 3 * NSPH, o, o + 6 * NSENS, g, mu[e],
                         netf + (size_t)e * 3 * M::NB);
        for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        for (int k = 0; k < ND; ++k) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
    }
    return 0;
}

#endif

#ifdef HOSTSIM_HAND
// Shadow hand + cube: per env  q[24] | qd[24] | laml[24] | target[24] | obj[13]  ->  updated in place; out: sensor[30] | dof_force[24] | ncontact
// scale: null or [nenv][8] per-env `actor_params` factors (core/hand_engine.hpp HS_*); limit_shift: null or [nenv][48]
// stiffness of the asset's hand-to-hand contact pairs for the next hs_step_hand / hs_step_hand_mw calls (default: the engine's 2e4 N/m for the Shadow Hand; 0 = off); the pair sides
// pushed in the last sub-step of the last call's env 0 come back through hs_hand_pair_sides
static float g_pair_k = 2.0e4f;
static int g_pair_sides = 0;
extern "C" void hs_set_hand_pair_stiffness(float k) { g_pair_k = k; }
extern "C" int hs_hand_pair_sides() { return g_pair_sides; }
extern "C" int hs_step_hand(const SimParams* P, int nenv, float* state, float* out, const float* root13, float half, float mass, float inertia,
                            float mu, const float* scale, const float* limit_shift) {
    using M = ModelShadowHand;
    constexpr int ND = M::ND, NS = M::NSENS;
    const int ss = 4 * ND + 13, os = 6 * NS + ND + 1;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        HandSim<M> sim;
        ObjectParams OP{half, mass, inertia, mu};
        if (scale) {
            OP.randomise(scale[e * HS_COLUMNS + HS_OBJECT_MASS], scale[e * HS_COLUMNS + HS_OBJECT_SCALE]);
            sim.actor_scale = Strided{const_cast<float*>(scale) + e * HS_COLUMNS, 1};
        }
        static const float no_shift[2 * ND] = {0};
        sim.limit_shift = Strided{const_cast<float*>(limit_shift ? limit_shift + e * 2 * ND : no_shift), 1};
        for (int k = 0; k < 13; ++k) sim.root[k] = root13[k];
        sim.pair_k = g_pair_k;
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[k]; sim.qd[k] = s[ND + k]; }
        float* ob = s + 4 * ND;
        for (int k = 0; k < 3; ++k) { sim.obj.pos[k] = ob[k]; sim.obj.vel[k] = ob[7 + k]; sim.obj.angvel[k] = ob[10 + k]; }
        for (int k = 0; k < 4; ++k) sim.obj.quat[k] = ob[3 + k];
        static thread_local float rows[HandSim<M>::ROW_SLOTS];
        const float h = P->dt / (float)P->substeps;
        int nc = 0;
        for (int it = 0; it < P->substeps; ++it)
            sim.substep_hand(*P, OP, s + 3 * ND, h, RowStore<1>{rows}, Strided{s + 2 * ND, 1}, Strided{o, 1}, Strided{o + 6 * NS, 1}, &nc);
        o[6 * NS + ND] = (float)(nc & 0xFFFF);
        if (e == 0) g_pair_sides = sim.pair_active;
        for (int k = 0; k < ND; ++k) { s[k] = sim.q[k]; s[ND + k] = sim.qd[k]; }
        for (int k = 0; k < 3; ++k) { ob[k] = sim.obj.pos[k]; ob[7 + k] = sim.obj.vel[k]; ob[10 + k] = sim.obj.angvel[k]; }
        for (int k = 0; k < 4; ++k) ob[3 + k] = sim.obj.quat[k];
    }
    return 0;
}
// the finger-per-wave form (core/hand_engine_mw.hpp): the NROLE role waves of an env run as threads that meet at a pthread barrier where
// the GPU waves meet at s_barrier; one shared row store (stride 1).  State / out layout of hs_step_hand; shape 0 box, 1 capsule, 2 ellipsoid
// (dims / inertia3: the object's dimensions and principal inertias for shapes 1, 2)
struct HandMwJob {
    const SimParams* P; float* s; float* o; const float* root13; ObjectParams OP; const float* scale; const float* lshift; float* rows;
    pthread_barrier_t* bar; int* nc; int* sides;
};
template <int R, int SHAPE>
static void hand_mw_thread(HandMwJob j) {
    using M = ModelShadowHand;
    using MW = SimMW<M>;
    constexpr int ND = M::ND, NS = M::NSENS;
    HandSimMW<M> sim;
    if (j.scale) sim.actor_scale = Strided{const_cast<float*>(j.scale), 1};
    sim.limit_shift = Strided{const_cast<float*>(j.lshift), 1};
    sim.pair_k = g_pair_k;
    for (int k = 0; k < 13; ++k) sim.root[k] = j.root13[k];
    float* ob = j.s + 4 * ND;
    const float h = j.P->dt / (float)j.P->substeps;
    for (int it = 0; it < j.P->substeps; ++it) {
        for (int k = 0; k < ND; ++k) { sim.q[k] = j.s[k]; sim.qd[k] = j.s[ND + k]; }
        for (int k = 0; k < 3; ++k) { sim.obj.pos[k] = ob[k]; sim.obj.vel[k] = ob[7 + k]; sim.obj.angvel[k] = ob[10 + k]; }
        for (int k = 0; k < 4; ++k) sim.obj.quat[k] = ob[3 + k];
        pthread_barrier_wait(j.bar);     // everybody has read the state of the previous sub-step
        int nc = 0;
        sim.template substep_hand_role<R, 1, SHAPE>(*j.P, j.OP, j.s + 3 * ND, h, RowStore<1>{j.rows}, Strided{j.s + 2 * ND, 1}, Strided{j.o, 1},
                                                    Strided{j.o + 6 * NS, 1}, &nc, HostBarrier{j.bar});
        sfor<ND>([&](auto K) { if constexpr (MW::template owns_gi<R>(K)) { j.s[K] = sim.q[K]; j.s[ND + K] = sim.qd[K]; } });
        j.sides[R] = sim.pair_active;
        if constexpr (R == M::TRUNK_ROLE) {
            for (int k = 0; k < 3; ++k) { ob[k] = sim.obj.pos[k]; ob[7 + k] = sim.obj.vel[k]; ob[10 + k] = sim.obj.angvel[k]; }
            for (int k = 0; k < 4; ++k) ob[3 + k] = sim.obj.quat[k];
            *j.nc = nc;
        }
        pthread_barrier_wait(j.bar);     // the new state is complete
    }
}
template <int SHAPE>
static void hand_mw_env(const HandMwJob& j) {
    static_assert(ModelShadowHand::NROLE == 4, "four roles");
    std::thread t0(hand_mw_thread<0, SHAPE>, j), t1(hand_mw_thread<1, SHAPE>, j), t2(hand_mw_thread<2, SHAPE>, j), t3(hand_mw_thread<3, SHAPE>, j);
    t0.join(); t1.join(); t2.join(); t3.join();
}
extern "C" int hs_step_hand_mw(const SimParams* P, int nenv, float* state, float* out, const float* root13, float half, float mass, float inertia,
                               float mu, const float* scale, const float* limit_shift, int shape, const float* dims, const float* inertia3) {
    using M = ModelShadowHand;
    constexpr int ND = M::ND, NS = M::NSENS;
    const int ss = 4 * ND + 13, os = 6 * NS + ND + 1;
    static const float no_shift[2 * ND] = {0};
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        ObjectParams OP{half, mass, inertia, mu};
        if (shape != 0) for (int k = 0; k < 3; ++k) { OP.dims[k] = dims[k]; OP.inertia3[k] = inertia3[k]; }
        if (scale) OP.randomise(scale[e * HS_COLUMNS + HS_OBJECT_MASS], scale[e * HS_COLUMNS + HS_OBJECT_SCALE]);
        std::vector<float> rows(HandSimMW<M>::MW_SLOTS, 0.f);
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, 4);
        int nc = 0, sides[4] = {0, 0, 0, 0};
        HandMwJob j{P, s, o, root13, OP, scale ? scale + e * HS_COLUMNS : nullptr, limit_shift ? limit_shift + e * 2 * ND : no_shift, rows.data(), &bar, &nc, sides};
        if (shape == 0) hand_mw_env<OBJ_BOX>(j); else if (shape == 1) hand_mw_env<OBJ_CAPSULE>(j); else hand_mw_env<OBJ_ELLIPSOID>(j);
        pthread_barrier_destroy(&bar);
        if (e == 0) g_pair_sides = sides[0] + sides[1] + sides[2] + sides[3];
        o[6 * NS + ND] = (float)(nc & 0xFFFF);
    }
    return 0;
}
extern "C" int hs_hand_fingertips(int nenv, const float* state, const float* root13, float* out /* [nenv][5][13] */) {
    using M = ModelShadowHand;
    constexpr int ND = M::ND;
    for (int e = 0; e < nenv; ++e) {
        HandSim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = root13[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = state[(size_t)e * (4 * ND + 13) + k]; sim.qd[k] = state[(size_t)e * (4 * ND + 13) + ND + k]; }
        sim.fingertip_states((float (*)[13])(out + (size_t)e * M::NSENS * 13));
    }
    return 0;
}
#endif

#ifdef HOSTSIM_BBOT
// BallBalance (core/bbot_engine.hpp).  state per env: root 13, q 6, qd 6, laml 6, lamp 9, ball 13; target[nenv][6]; out per env: sensor 18, ncontact
extern "C" int hs_step_bbot(const SimParams* P, const BbotPhys* bp, int nenv, float* state, const float* target, float* out) {
    using M = ModelBalanceBot;
    constexpr int ND = M::ND, SS = 13 + 3 * ND + 9 + 13, OS = 19;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * SS;
        float* o = out + (size_t)e * OS;
        BbotSim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        const float* b = s + 13 + 3 * ND + 9;
        for (int k = 0; k < 3; ++k) { sim.ball.pos[k] = b[k]; sim.ball.vel[k] = b[7 + k]; sim.ball.angvel[k] = b[10 + k]; }
        for (int k = 0; k < 4; ++k) sim.ball.quat[k] = b[3 + k];
        const float h = P->dt / (float)P->substeps;
        int nc = 0;
        for (int it = 0; it < P->substeps; ++it)
            sim.substep(*P, *bp, h, target + (size_t)e * ND, Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 3 * ND, 1}, Strided{o, 1}, &nc);
        o[18] = (float)nc;
        for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        for (int k = 0; k < ND; ++k) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
        float* bo = s + 13 + 3 * ND + 9;
        for (int k = 0; k < 3; ++k) { bo[k] = sim.ball.pos[k]; bo[7 + k] = sim.ball.vel[k]; bo[10 + k] = sim.ball.angvel[k]; }
        for (int k = 0; k < 4; ++k) bo[3 + k] = sim.ball.quat[k];
    }
    return 0;
}
#endif

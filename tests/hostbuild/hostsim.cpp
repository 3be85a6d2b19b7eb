// Host (g++) build of the specialised engine core -- TEST-ONLY.  Lets the CPU test-suite compare the exact
// code the HIP kernels run (csrc/core/engine.hpp + generated model tables) against the independent oracle
// without a GPU.  Never loaded by the product path.
#include <cstring>
#include "../../isaacgymenvs_amd/csrc/core/engine.hpp"
#include "../../isaacgymenvs_amd/csrc/gen/model_cartpole.h"
#include "../../isaacgymenvs_amd/csrc/gen/model_ant.h"
#include "../../isaacgymenvs_amd/csrc/gen/model_anymal.h"
#ifndef HOSTSIM_NO_HUMANOID
#include "../../isaacgymenvs_amd/csrc/gen/model_humanoid.h"
#endif

using namespace mi;

// state/out layouts identical to oracle/physics.c (AoS per env)
template <class M>
static void run(const SimParams* P, int nenv, float* state, const float* tau, float* out) {
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND, os = 6 * NSENS + ND + 3 * NSPH;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nenv; ++e) {
        float* s = state + (size_t)e * ss;
        float* o = out + (size_t)e * os;
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = s[k];
        for (int k = 0; k < ND; ++k) { sim.q[k] = s[13 + k]; sim.qd[k] = s[13 + ND + k]; }
        // warm-start impulses and outputs are updated in place (same layout as oracle/physics.c)
        sim.step(*P, tau + (size_t)e * ND, s + 13 + 2 * ND, s + 13 + 2 * ND + 3 * NSPH, o, o + 6 * NSENS);
        for (int k = 0; k < 13; ++k) s[k] = sim.root[k];
        for (int k = 0; k < ND; ++k) { s[13 + k] = sim.q[k]; s[13 + ND + k] = sim.qd[k]; }
    }
}

extern "C" int hs_step(const char* model, const SimParams* P, int nenv, float* state, const float* tau, float* out) {
    if (!strcmp(model, "cartpole")) run<ModelCartpole>(P, nenv, state, tau, out);
    else if (!strcmp(model, "ant")) run<ModelAnt>(P, nenv, state, tau, out);
#ifndef HOSTSIM_NO_HUMANOID
    else if (!strcmp(model, "humanoid")) run<ModelHumanoid>(P, nenv, state, tau, out);
#endif
    else return -1;
    return 0;
}

// height-field variant (AnymalTerrain): per-env friction mu[nenv], net contact forces netf[nenv][3*NB]
extern "C" int hs_step_terrain(const char* model, const SimParams* P, int nenv, float* state, const float* tau, float* out,
                               const short* hs, int rows, int cols, float hscale, float vscale, float border, const float* mu,
                               float* netf) {
    if (strcmp(model, "anymal")) return -1;
    using M = ModelAnymal;
    constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
    const int ss = 13 + 2 * ND + 3 * NSPH + ND, os = 6 * NSENS + ND + 3 * NSPH;
    const HeightfieldGround g{hs, rows, cols, hscale, vscale, border};
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

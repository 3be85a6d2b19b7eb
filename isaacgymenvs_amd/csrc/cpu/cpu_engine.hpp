// cpu_engine.hpp -- what the translation units of the CPU backend share: the engine object and the per-thread statistics accumulators.
// (cpu/mi_engine_cpu.cpp: C ABI + Cartpole / Ant / Humanoid / Quadcopter / Ingenuity / BallBalance; cpu/cpu_anymal.cpp: AnymalTerrain, Anymal;
//  cpu/cpu_hand.cpp, compiled once per hand and object shape: ShadowHand, AllegroHand.)
#pragma once
#include <omp.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../core/engine.hpp"
#include "../arena_layout.hpp"
#include "../task_views.hpp"
#include "../tasks/shadow_hand.hpp"

using namespace mi;

struct MiEngine {
    int task, N;
    SimParams P;
    LocoParams loco;
    CartpoleParams cart;
    QuadcopterParams quad;
    IngenuityParams ing;
    BallBalanceParams bbot;
    AnymalParams anymal;
    AnymalFlatParams anymal_flat;
    HandParams hand;
    float artic[(sizeof(MiArticulationParams) + 3) / 4];    // ArticulationParams (csrc/tasks/articulation.hpp), opaque outside cpu_articulation.cpp
    QuadView qv;
    IngenuityView iv;
    BbotView bv;
    HandView hv;
    AnymalTerrainDesc terrain;          // points into the two vectors below (mi_engine_set_terrain copies the caller's arrays)
    std::vector<short> terrain_hs;
    std::vector<float> terrain_origins;
    int max_init_level;
    View v;
    float clip_obs;
    int control_freq_inv, num_threads;
    std::vector<MiTensorDesc> descs;
    unsigned long long steps;
    float* lamp_arena;
    float* dof_api_arena;
    float* actor_scale_arena;   // the `actor_params` tensors: read by the sub-step once option "actor_tensors" is on (as in mi_engine.hip)
    float* limit_shift_arena;
};

// job statistics (View::stats) of one OpenMP thread; added up in thread order after the loop over envs
struct StatAcc { double fin_ret = 0, fin_len = 0, fin = 0, r = 0, cnt = 0; };
static inline void episode_stats_env(const View& v, int en, float rew, long long reset, long long progress, StatAcc& a) {
    float ret = v.ep_ret[en] + rew;
    a.r += rew; a.cnt += 1;
    if (reset != 0) { a.fin_ret += ret; a.fin_len += (double)(progress + 1); a.fin += 1; ret = 0.f; }
    v.ep_ret[en] = ret;
}
static inline void flush_stats(const View& v, const std::vector<StatAcc>& accs) {
    for (const StatAcc& a : accs) { v.stats[0] += (float)a.fin_ret; v.stats[1] += (float)a.fin_len; v.stats[2] += (float)a.fin; v.stats[3] += (float)a.r; v.stats[4] += (float)a.cnt; }
}

// the tasks of the other translation units: rc 0, or -1 with *err set
int cpu_anymal_init(MiEngine* e, std::string* err);
int cpu_anymal_step(MiEngine* e, const float* actions, bool simulate_only);
int cpu_anymal_reset(MiEngine* e, const int64_t* ids, int n);
int cpu_anymal_body_states(MiEngine* e);
int cpu_anymal_kinematics(MiEngine* e, float* out_j, float* out_h);
int cpu_articulation_reset(MiEngine* e, const int64_t* ids, int n);      // ids == nullptr: every env
int cpu_articulation_simulate(MiEngine* e);
int cpu_articulation_body_states(MiEngine* e);
int cpu_articulation_kinematics(MiEngine* e, float* out_j, float* out_h);
int cpu_hand_init(MiEngine* e);
int cpu_hand_step(MiEngine* e, const float* actions, bool simulate_only);
int cpu_hand_reset(MiEngine* e, const int64_t* ids, int n);
int cpu_hand_body_states(MiEngine* e);
int cpu_hand_kinematics(MiEngine* e, float* out_j, float* out_h);

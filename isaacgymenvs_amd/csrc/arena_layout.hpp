// arena_layout.hpp -- task table and arena layout shared by the HIP library (mi_engine.hip) and the CPU backend
// (cpu/mi_engine_cpu.cpp): one definition of which tensor sits where, so that the Python side sees the same views on both.
// No HIP dependency.
#pragma once
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mi_engine.h"
#include "arena.hpp"
#include "tasks/locomotion.hpp"
#include "tasks/anymal.hpp"
#include "tasks/quadcopter.hpp"
#include "tasks/ingenuity.hpp"
#include "tasks/ball_balance.hpp"
#include "gen/model_ant.h"
#include "gen/model_cartpole.h"
#include "gen/model_humanoid.h"
#include "gen/model_anymal.h"
#include "gen/model_shadow_hand.h"
#include "gen/model_allegro_hand.h"
#include "gen/model_quadcopter.h"
#include "gen/model_ingenuity.h"
#include "gen/model_balance_bot.h"

using namespace mi;

enum TaskId { T_CARTPOLE = 0, T_ANT = 1, T_HUMANOID = 2, T_ANYMAL = 3, T_SHADOWHAND = 4, T_ANYMAL_FLAT = 5, T_QUADCOPTER = 6, T_INGENUITY = 7, T_BALLBALANCE = 8, T_ALLEGROHAND = 9,
              T_ARTICULATION = 10 };
constexpr int kNumTasks = 11;
// The Articulation task's robot is compiled at run time (assets/runtime.py): its constants live in ONE translation unit of each library
// (kernels_articulation.hip / cpu/cpu_articulation.cpp), which is all a run-time variant has to recompile; everybody else learns the robot's sizes
// from this function instead of including gen/model_articulation.h.
struct ArticulationMeta { int nd, nb, nsens, nsph, fixed; };
ArticulationMeta mi_articulation_meta();
static inline bool is_hand_task(int t) { return t == T_SHADOWHAND || t == T_ALLEGROHAND; }
struct TaskMeta { const char* name; int nobs, nact, nd, nb, nsens, nsph, fixed; size_t pbytes; };
static const TaskMeta kTasks[] = {
    {"Cartpole", 4, 1, ModelCartpole::ND, ModelCartpole::NB, 0, ModelCartpole::NSPH, 1, sizeof(MiCartpoleParams)},
    {"Ant", Loco<ModelAnt::ND, 6 * ModelAnt::NSENS, false>::NOBS, ModelAnt::ND, ModelAnt::ND, ModelAnt::NB, ModelAnt::NSENS, ModelAnt::NSPH, 0, sizeof(MiLocoParams)},
    {"Humanoid", Loco<ModelHumanoid::ND, 6 * ModelHumanoid::NSENS, true>::NOBS, ModelHumanoid::ND, ModelHumanoid::ND, ModelHumanoid::NB, ModelHumanoid::NSENS, ModelHumanoid::NSPH, 0, sizeof(MiLocoParams)},
    {"AnymalTerrain", kAnymalObs, kAnymalDof, ModelAnymal::ND, ModelAnymal::NB, 0, ModelAnymal::NSPH, 0, sizeof(MiAnymalParams)},
    {"ShadowHand", 211, 20, ModelShadowHand::ND, ModelShadowHand::NB, ModelShadowHand::NSENS, 0, 1, sizeof(MiHandParams)},
    {"Anymal", kAnymalFlatObs, kAnymalDof, ModelAnymal::ND, ModelAnymal::NB, 0, ModelAnymal::NSPH, 0, sizeof(MiAnymalFlatParams)},
    {"Quadcopter", kQuadObs, kQuadAct, ModelQuadcopter::ND, ModelQuadcopter::NB, ModelQuadcopter::NSENS, ModelQuadcopter::NSPH, 0, sizeof(MiQuadcopterParams)},
    {"Ingenuity", kIngObs, kIngAct, ModelIngenuity::ND, ModelIngenuity::NB, ModelIngenuity::NSENS, ModelIngenuity::NSPH, 0, sizeof(MiIngenuityParams)},
    {"BallBalance", kBbotObs, kBbotAct, ModelBalanceBot::ND, ModelBalanceBot::NB, ModelBalanceBot::NSENS, ModelBalanceBot::NSPH, 0, sizeof(MiBallBalanceParams)},
    // reference allegro_hand.py: 88-wide full_state (:485-507), 16 driven dofs, the task parameters of the ShadowHand
    {"AllegroHand", 88, 16, ModelAllegroHand::ND, ModelAllegroHand::NB, ModelAllegroHand::NSENS, 0, 1, sizeof(MiHandParams)},
    // no observations / actions of its own (one placeholder column each); sizes from the articulation's translation unit, at load time
    {"Articulation", 1, 1, mi_articulation_meta().nd, mi_articulation_meta().nb, mi_articulation_meta().nsens, mi_articulation_meta().nsph,
     mi_articulation_meta().fixed, sizeof(MiArticulationParams)},
};
static int find_task(const char* t) {
    for (int i = 0; i < kNumTasks; ++i) if (!strcmp(t, kTasks[i].name)) return i;
    return -1;
}

struct Layout {
    std::vector<MiTensorDesc> d;
    size_t off = 0;
    size_t add(const char* name, int dtype, std::vector<int64_t> shape, std::vector<int64_t> stride, size_t count) {
        static const size_t es[] = {4, 8, 1, 4};
        off = (off + 255) & ~size_t(255);
        MiTensorDesc t;
        memset(&t, 0, sizeof(t));
        snprintf(t.name, sizeof(t.name), "%s", name);
        t.dtype = dtype; t.ndim = (int)shape.size();
        for (size_t i = 0; i < shape.size(); ++i) { t.shape[i] = shape[i]; t.stride[i] = stride[i]; }
        t.byte_offset = (int64_t)off;
        d.push_back(t);
        size_t o = off;
        off += count * es[dtype];
        return o;
    }
};

// nobs: observation width; 0 = the task's (maximum) width.  ShadowHand's observationType variants are narrower.
static void build_layout(int task, int N, Layout& L, View* v, char* base, int nobs = 0) {
    const TaskMeta& m = kTasks[task];
    const int64_t n = N, nd = m.nd, ns = m.nsens, nsp = m.nsph, no = nobs > 0 ? nobs : m.nobs, na = m.nact;
    auto P = [&](size_t o) { return base ? base + o : (char*)nullptr; };
    size_t o;
    o = L.add("root_states", MI_F32, {n, 13}, {1, n}, 13 * n); if (v) v->root = (float*)P(o);
    o = L.add("dof_state", MI_F32, {n, nd, 2}, {1, n, nd * n}, 2 * nd * n); if (v) v->dof = (float*)P(o);
    o = L.add("dof_actuation_force", MI_F32, {n, nd}, {1, n}, nd * n); if (v) v->tau = (float*)P(o);
    o = L.add("contact_impulse", MI_F32, {n, nsp > 0 ? nsp : 1, 3}, {1, 3 * n, n}, (nsp > 0 ? 3 * nsp : 3) * n); if (v) v->lamc = (float*)P(o);
    o = L.add("limit_impulse", MI_F32, {n, nd}, {1, n}, nd * n); if (v) v->laml = (float*)P(o);
    o = L.add("force_sensor", MI_F32, {n, ns > 0 ? ns : 1, 6}, {1, 6 * n, n}, (ns > 0 ? 6 * ns : 6) * n); if (v) v->sensor = (float*)P(o);
    o = L.add("dof_force", MI_F32, {n, nd}, {1, n}, nd * n); if (v) v->dof_force = (float*)P(o);
    o = L.add("potentials", MI_F32, {n}, {1}, n); if (v) v->potentials = (float*)P(o);
    o = L.add("prev_potentials", MI_F32, {n}, {1}, n); if (v) v->prev_potentials = (float*)P(o);
    o = L.add("up_vec", MI_F32, {n, 3}, {1, n}, 3 * n); if (v) v->up_vec = (float*)P(o);
    o = L.add("heading_vec", MI_F32, {n, 3}, {1, n}, 3 * n); if (v) v->heading_vec = (float*)P(o);
    o = L.add("actions", MI_F32, {n, na}, {1, n}, na * n); if (v) v->actions = (float*)P(o);
    o = L.add("initial_root_states", MI_F32, {n, 13}, {1, n}, 13 * n); if (v) v->init_root = (float*)P(o);
    o = L.add("obs_buf", MI_F32, {n, no}, {no, 1}, no * n); if (v) v->obs = (float*)P(o);
    o = L.add("obs_out", MI_F32, {2, n, no}, {n * no, no, 1}, 2 * no * n); if (v) v->obs_out = (float*)P(o);
    o = L.add("rew_buf", MI_F32, {n}, {1}, n); if (v) v->rew = (float*)P(o);
    o = L.add("reset_buf", MI_I64, {n}, {1}, n); if (v) v->reset = (long long*)P(o);
    o = L.add("progress_buf", MI_I64, {n}, {1}, n); if (v) v->progress = (long long*)P(o);
    o = L.add("randomize_buf", MI_I64, {n}, {1}, n); if (v) v->randomize = (long long*)P(o);
    o = L.add("timeout_buf", MI_U8, {n}, {1}, n); if (v) v->timeout = (unsigned char*)P(o);
    o = L.add("episode_count", MI_I32, {n}, {1}, n); if (v) v->episode = (int*)P(o);
    o = L.add("episode_return", MI_F32, {n}, {1}, n); if (v) v->ep_ret = (float*)P(o);
    o = L.add("episode_stats", MI_F32, {8}, {1}, 8); if (v) v->stats = (float*)P(o);
    // gym.acquire_rigid_body_state_tensor (shadow_hand.py:150-175): [pos3, quat xyzw, linvel3, angvel3] of every body of the articulation,
    // refreshed on demand by mi_engine_refresh_rigid_body_states (never touched by step / simulate)
    o = L.add("rigid_body_state", MI_F32, {n, (int64_t)m.nb, 13}, {1, 13 * n, n}, 13 * (int64_t)m.nb * n); if (v) v->body_state = (float*)P(o);
    if (task == T_ANT || task == T_HUMANOID) {
        // per-env friction of the robot's shapes for `actor_params.<actor>.rigid_shape_properties.friction` domain randomisation
        // (vec_task.py:752-828); negative = the model's own value
        o = L.add("friction", MI_F32, {n}, {1}, n); if (v) v->friction = (float*)P(o);
        // per-env scales of the actor's link masses (`actor_params.<actor>.rigid_body_properties.mass`: [0, nb)) and joint damping / stiffness /
        // armature (`.dof_properties.*`: [nb, nb + nd), [nb + nd, nb + 2 nd), [nb + 2 nd, nb + 3 nd)); 1 = the model's own values
        // (round 3: one factor per BODY for mass + inertia, then per DOF for damping, stiffness, armature -- the reference's granularity)
        const int64_t nas = (int64_t)m.nb + 3 * nd;
        o = L.add("actor_scale", MI_F32, {n, nas}, {1, n}, nas * n); if (v) { v->actor_scale = (float*)P(o); v->nas = (int)nas; }
        // `.dof_properties.lower / upper`: one shift per joint limit and env (0 = the model's limits)
        o = L.add("dof_limit_shift", MI_F32, {n, 2 * nd}, {1, n}, 2 * nd * n); if (v) v->limit_shift = (float*)P(o);
    }
    if (task == T_HUMANOID) {
        // the Humanoid actor collides with itself (collision filter 0, humanoid.py:194): warm-start impulses and contact forces of the
        // limb-pair groups (normal + 2 tangents; force = world force on the first body of the group's contact, last sub-step)
        const int64_t npg = ModelHumanoid::NPG;
        o = L.add("self_contact_impulse", MI_F32, {n, npg, 3}, {1, 3 * n, n}, 3 * npg * n); if (v) v->lamp = (float*)P(o);
        o = L.add("self_contact_force", MI_F32, {n, npg, 3}, {1, 3 * n, n}, 3 * npg * n); if (v) v->pairf = (float*)P(o);
        o = L.add("contact_dropped", MI_I32, {n, 2}, {1, n}, 2 * n); if (v) v->dropped = (int*)P(o);   // refused for want of a slot: ground, self
    }
    if (task == T_ANYMAL) {   // anymal_terrain.py:117-168
        const int64_t nb = m.nb;
        o = L.add("net_contact_force", MI_F32, {n, nb, 3}, {1, 3 * n, n}, 3 * nb * n); if (v) v->netf = (float*)P(o);
        o = L.add("commands", MI_F32, {n, 4}, {1, n}, 4 * n); if (v) v->commands = (float*)P(o);
        o = L.add("last_actions", MI_F32, {n, nd}, {1, n}, nd * n); if (v) v->last_actions = (float*)P(o);
        o = L.add("last_dof_vel", MI_F32, {n, nd}, {1, n}, nd * n); if (v) v->last_dof_vel = (float*)P(o);
        o = L.add("feet_air_time", MI_F32, {n, 4}, {1, n}, 4 * n); if (v) v->feet_air_time = (float*)P(o);
        o = L.add("episode_sums", MI_F32, {n, kAnymalSums}, {1, n}, kAnymalSums * n); if (v) v->episode_sums = (float*)P(o);
        o = L.add("env_origins", MI_F32, {n, 3}, {1, n}, 3 * n); if (v) v->env_origins = (float*)P(o);
        o = L.add("friction", MI_F32, {n}, {1}, n); if (v) v->friction = (float*)P(o);
        o = L.add("terrain_levels", MI_I32, {n}, {1}, n); if (v) v->terrain_levels = (int*)P(o);
        o = L.add("terrain_types", MI_I32, {n}, {1}, n); if (v) v->terrain_types = (int*)P(o);
        o = L.add("episode_step_stats", MI_F32, {16}, {1}, 16); if (v) v->ep_stats = (float*)P(o);
        o = L.add("episode_means", MI_F32, {16}, {1}, 16); if (v) v->ep_means = (float*)P(o);
        o = L.add("episode_cum_stats", MI_F32, {32}, {1}, 32); if (v) v->ep_cum = (float*)P(o);
        // the task's dof-state tensor as of its last gym.refresh_dof_state_tensor (anymal_terrain.py:451): one sim step behind `dof_state` after a
        // step (the base class's simulate() refreshes nothing, vec_task.py:379-382); what its PD law, observations and reward read -- View::dof_api
        o = L.add("dof_state_refreshed", MI_F32, {n, nd, 2}, {1, n, nd * n}, 2 * nd * n); if (v) v->dof_api = (float*)P(o);
    }
    if (task == T_ARTICULATION) {
        const int64_t nb = m.nb;
        o = L.add("net_contact_force", MI_F32, {n, nb, 3}, {1, 3 * n, n}, 3 * nb * n); if (v) v->netf = (float*)P(o);
        o = L.add("dof_position_targets", MI_F32, {n, nd}, {1, n}, nd * n); if (v) v->targets = (float*)P(o);
        // the scene beside the actor (core/scene_engine.hpp, MiScene): root states of its free boxes -- gym's actor root state rows of those actors
        // (franka_cube_stack.py:377-386) -- and its contact counters
        const int64_t nfb = MI_SCENE_MAX_FREE;
        o = L.add("scene_state", MI_F32, {n, nfb, 13}, {1, 13 * n, n}, 13 * nfb * n); if (v) v->scene = (float*)P(o);
        o = L.add("scene_contacts", MI_I32, {n, 2}, {1, n}, 2 * n); if (v) v->scene_nc = (int*)P(o);
        // warm start of the scene's contacts: (feature id as int bits, impulses x 3) of every contact slot of the last sub-step (SceneSim::KSLOT = 48)
        o = L.add("scene_warm", MI_F32, {n, MI_SCENE_WARM_SLOTS, 4}, {1, 4 * n, n}, 4 * MI_SCENE_WARM_SLOTS * n); if (v) v->scene_warm = (float*)P(o);
    }
    if (task == T_ANYMAL_FLAT) {   // anymal.py:100-125
        const int64_t nb = m.nb;
        o = L.add("net_contact_force", MI_F32, {n, nb, 3}, {1, 3 * n, n}, 3 * nb * n); if (v) v->netf = (float*)P(o);
        o = L.add("commands", MI_F32, {n, 3}, {1, n}, 3 * n); if (v) v->commands = (float*)P(o);
        // `actor_params` of Anymal.yaml (:121-165): per-env shape friction (negative = the model's own), link mass factors per body, and
        // the factors of the dof properties `stiffness` / `damping`, which for this task ARE the position drives' gains (anymal.py:203-206);
        // columns as for Ant / Humanoid: [nb] mass, [nd] damping, [nd] stiffness, [nd] armature.  The URDF has no joint limits: no limit shifts.
        o = L.add("friction", MI_F32, {n}, {1}, n); if (v) v->friction = (float*)P(o);
        const int64_t nas = nb + 3 * nd;
        o = L.add("actor_scale", MI_F32, {n, nas}, {1, n}, nas * n); if (v) { v->actor_scale = (float*)P(o); v->nas = (int)nas; }
    }
    L.off = (L.off + 255) & ~size_t(255);
}

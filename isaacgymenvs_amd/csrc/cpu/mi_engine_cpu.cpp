// mi_engine_cpu.cpp -- CPU product backend: the engine lifecycle part of include/mi_engine.h on a HOST arena.
//
// What it is for: the reference's `sim_device=cpu pipeline=cpu` configuration (BASELINE.json config 1: Cartpole num_envs=64;
// reference device resolution isaacgymenvs/tasks/base/vec_task.py:78-88, PhysX worker threads `num_threads`, cfg/config.yaml:30).
// What it is built from: the SAME sources the HIP kernels compile -- csrc/core/engine.hpp (per-env physics sub-step, host build of
// the specialised code), csrc/tasks/locomotion.hpp (observations / reward / reset), csrc/arena_layout.hpp (arena layout) -- with
// OpenMP over envs where the kernels have one env per lane.  It does NOT use oracle/ (test infrastructure).  Same SoA arena layout,
// same counter-based reset RNG: the Python side sees identical tensors and, up to fp32 summation order in the episode statistics,
// identical numbers on both devices.
//
// Tasks: every task of the table (csrc/arena_layout.hpp).  This file: the C ABI + Cartpole, Ant, Humanoid (self-collision included), Quadcopter,
// Ingenuity, BallBalance; cpu_anymal.cpp: AnymalTerrain (BASELINE config 4), Anymal; cpu_hand.cpp + cpu_hand_physics.cpp: ShadowHand (config 5),
// AllegroHand.
// Build: g++ -fopenmp, one object per translation unit in parallel, linked into libmi_engine_cpu.so (isaacgymenvs_amd/native.py::build_cpu);
// the library exports the lifecycle subset of include/mi_engine.h.
#include "cpu_engine.hpp"
#include "cpu_hand.hpp"
#include "../core/bbot_engine.hpp"

static_assert(sizeof(MiSimParams) == sizeof(SimParams), "MiSimParams layout");
static_assert(sizeof(MiLocoParams) == sizeof(LocoParams), "MiLocoParams layout");
static_assert(sizeof(MiCartpoleParams) == sizeof(CartpoleParams), "MiCartpoleParams layout");
static_assert(sizeof(MiQuadcopterParams) == sizeof(QuadcopterParams), "MiQuadcopterParams layout");
static_assert(sizeof(MiIngenuityParams) == sizeof(IngenuityParams), "MiIngenuityParams layout");
static_assert(sizeof(MiBallBalanceParams) == sizeof(BallBalanceParams), "MiBallBalanceParams layout");
static_assert(sizeof(MiAnymalParams) == sizeof(AnymalParams), "MiAnymalParams layout");
static_assert(sizeof(MiAnymalFlatParams) == sizeof(AnymalFlatParams), "MiAnymalFlatParams layout");
static_assert(sizeof(MiHandRewardParams) == sizeof(HandRewardParams), "MiHandRewardParams layout");
static_assert(sizeof(MiHandParams) == sizeof(HandParams), "MiHandParams layout");

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }
extern "C" const char* mi_last_error(void) { return g_err.c_str(); }
extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }

static bool cpu_task(int t) { return t >= 0 && t < kNumTasks; }      // every task of the table has a host build
static void build_task_extras(int t, int n, Layout& L, MiEngine* e, char* base) {
    if (is_hand_task(t)) build_hand_layout(t, n, L, e ? &e->hv : nullptr, base);
    if (t == T_QUADCOPTER) build_quad_layout(n, L, e ? &e->qv : nullptr, base);
    if (t == T_INGENUITY) build_ingenuity_layout(n, L, e ? &e->iv : nullptr, base);
    if (t == T_BALLBALANCE) build_bbot_layout(n, L, e ? &e->bv : nullptr, base);
}

extern "C" int mi_task_info(const char* task, MiTaskInfo* out) {
    const int t = task ? find_task(task) : -1;
    if (t < 0 || !out) return fail(std::string("unknown task: ") + (task ? task : "(null)"));
    const TaskMeta& m = kTasks[t];
    out->num_obs = m.nobs; out->num_actions = m.nact; out->num_dofs = m.nd; out->num_bodies = m.nb; out->num_sensors = m.nsens;
    out->num_contact_spheres = m.nsph; out->fixed_base = m.fixed; out->task_params_bytes = (int)m.pbytes;
    return 0;
}
extern "C" size_t mi_engine_arena_bytes(const char* task, int num_envs) {
    const int t = task ? find_task(task) : -1;
    if (t < 0 || num_envs <= 0 || !cpu_task(t)) { fail("mi_engine_arena_bytes: task not available on the CPU backend"); return 0; }
    Layout L;
    build_layout(t, num_envs, L, nullptr, nullptr);      // (a narrower ShadowHand observationType only uses a prefix of obs_buf / obs_out)
    build_task_extras(t, num_envs, L, nullptr, nullptr);
    return L.off;
}
extern "C" int mi_engine_create(const char* task, const MiSimParams* sim, const void* task_params, size_t task_params_bytes, int num_envs,
                                int env_id_offset, uint64_t seed, void* arena, size_t arena_bytes, MiEngine** out) {
    if (!task || !sim || !task_params || !arena || !out) return fail("mi_engine_create: null argument");
    const int t = find_task(task);
    if (t < 0) return fail(std::string("unknown task: ") + task);
    if (task_params_bytes != kTasks[t].pbytes) return fail("mi_engine_create: task_params size mismatch");
    if (num_envs <= 0) return fail("mi_engine_create: num_envs <= 0");
    if (sim->substeps < 1 || sim->dt <= 0.f) return fail("mi_engine_create: invalid sim params");
    MiEngine* e = new (std::nothrow) MiEngine();
    if (!e) return fail("out of host memory");
    e->task = t; e->N = num_envs; e->steps = 0; e->control_freq_inv = 1; e->clip_obs = INFINITY; e->num_threads = 4;   // cfg/config.yaml:30
    memcpy(&e->P, sim, sizeof(SimParams));
    if (t == T_CARTPOLE) memcpy(&e->cart, task_params, sizeof(CartpoleParams));
    else if (t == T_QUADCOPTER) memcpy(&e->quad, task_params, sizeof(QuadcopterParams));
    else if (t == T_INGENUITY) memcpy(&e->ing, task_params, sizeof(IngenuityParams));
    else if (t == T_BALLBALANCE) memcpy(&e->bbot, task_params, sizeof(BallBalanceParams));
    else if (t == T_ANYMAL) memcpy(&e->anymal, task_params, sizeof(AnymalParams));
    else if (t == T_ANYMAL_FLAT) memcpy(&e->anymal_flat, task_params, sizeof(AnymalFlatParams));
    else if (is_hand_task(t)) memcpy(&e->hand, task_params, sizeof(HandParams));
    else if (t == T_ARTICULATION) memcpy(e->artic, task_params, sizeof(MiArticulationParams));
    else memcpy(&e->loco, task_params, sizeof(LocoParams));
    memset(&e->terrain, 0, sizeof(e->terrain));
    e->terrain.walls = 1;
    e->max_init_level = 0;
    int nobs = 0;
    if (is_hand_task(t)) {          // the checks of mi_engine.hip
        const HandParams& hp = e->hand;
        const int nfull = kTasks[t].nobs;     // ShadowHand 211, AllegroHand 88
        const bool ok = (hp.obs_type == 0 && hp.num_obs == nfull) || (hp.obs_type >= 1 && hp.obs_type <= 3 && hp.num_obs >= 1 && hp.num_obs <= 160);
        if (!ok) { delete e; return fail("mi_engine_create: hand task obs_type / num_obs invalid"); }
        if (hp.object_shape < 0 || hp.object_shape > 2) { delete e; return fail("mi_engine_create: hand task object_shape must be 0 (block), 1 (pen) or 2 (egg)"); }
        if (hp.object_shape != 0)
            for (int k = 0; k < 3; ++k)
                if ((!(hp.object_dims[k] > 0.f) && !(hp.object_shape == 1 && k == 2)) || !(hp.object_inertia[k] > 0.f)) {
                    delete e;
                    return fail("mi_engine_create: hand task egg / pen need positive object_dims / object_inertia");
                }
        if (!(hp.cube_mass > 0.f)) { delete e; return fail("mi_engine_create: hand task object mass must be positive"); }
        for (int k = 0; hp.obs_type != 0 && k < hp.num_obs; ++k)
            if (hp.obs_map[k] < 0 || hp.obs_map[k] >= nfull) { delete e; return fail("mi_engine_create: hand task obs_map entry out of range"); }
        nobs = hp.num_obs;
    }
    if (t == T_INGENUITY && e->ing.target_period < 1) { delete e; return fail("mi_engine_create: Ingenuity target_period must be positive"); }
    if (t == T_BALLBALANCE && !(e->bbot.ball_mass > 0.f && e->bbot.ball_inertia > 0.f && e->bbot.ball_radius > 0.f && e->bbot.pin_stiffness + e->bbot.pin_damping > 0.f)) {
        delete e;
        return fail("mi_engine_create: BallBalance needs positive ball mass / inertia / radius and a non-zero attractor");
    }
    memset(&e->v, 0, sizeof(View));
    memset(&e->qv, 0, sizeof(e->qv)); memset(&e->iv, 0, sizeof(e->iv)); memset(&e->bv, 0, sizeof(e->bv)); memset(&e->hv, 0, sizeof(e->hv));
    e->hv.drive_clamp = 1;
    // the asset's hand-to-hand contact pairs (Shadow Hand, shared.xml:31-51) are on by default; the Allegro hand's URDF lists none
    e->hv.pair_k = (t == T_SHADOWHAND) ? 2.0e4f : 0.f;
    e->hv.tips_in_post = 1;
    e->hv.pre_parts = 4;
    Layout L;
    build_layout(t, num_envs, L, &e->v, (char*)arena, nobs);
    build_task_extras(t, num_envs, L, e, (char*)arena);
    if (arena_bytes < L.off) { delete e; return fail("mi_engine_create: arena too small"); }
    e->descs = L.d;
    e->lamp_arena = e->v.lamp;
    e->dof_api_arena = e->v.dof_api;
    e->actor_scale_arena = e->v.actor_scale; e->limit_shift_arena = e->v.limit_shift;
    e->v.actor_scale = nullptr; e->v.limit_shift = nullptr;
    e->v.N = num_envs; e->v.env_offset = env_id_offset; e->v.seed = (uint32_t)(seed ^ (seed >> 32));
    e->v.clip_obs = INFINITY;
    *out = e;
    return 0;
}
extern "C" void mi_engine_destroy(MiEngine* e) { delete e; }
extern "C" int mi_engine_num_tensors(const MiEngine* e) { return e ? (int)e->descs.size() : fail("null engine"); }
extern "C" int mi_engine_tensor_desc(const MiEngine* e, int i, MiTensorDesc* out) {
    if (!e || !out || i < 0 || i >= (int)e->descs.size()) return fail("mi_engine_tensor_desc: bad index");
    *out = e->descs[i];
    return 0;
}
extern "C" int mi_engine_set_option(MiEngine* e, const char* key, double value) {
    if (!e) return fail("null engine");
    if (!strcmp(key, "clip_obs")) { e->clip_obs = (float)value; e->v.clip_obs = (float)value; return 0; }
    if (!strcmp(key, "gravity_x")) { e->P.g[0] = (float)value; return 0; }
    if (!strcmp(key, "gravity_y")) { e->P.g[1] = (float)value; return 0; }
    if (!strcmp(key, "gravity_z")) { e->P.g[2] = (float)value; return 0; }
    if (!strcmp(key, "control_freq_inv")) { if (value < 1) return fail("control_freq_inv < 1"); e->control_freq_inv = (int)value; return 0; }
    if (!strcmp(key, "self_collision")) {
        if (value != 0 && !e->lamp_arena) return fail("self_collision: this task's actor has no self-collision tables");
        e->v.lamp = value != 0 ? e->lamp_arena : nullptr;
        return 0;
    }
    if (!strcmp(key, "multi_wave") || !strcmp(key, "fused_sub") || !strcmp(key, "fused_post")) return 0;     // GPU launch shapes: nothing to do here
    if (!strcmp(key, "dof_state_lag")) {        // AnymalTerrain: PD law / observations / reward read the dof state of the task's last refresh (1, default, the
        // reference's behaviour: anymal_terrain.py:441-455 + vec_task.py:379-382) or the physics state (0).  Switching it on re-synchronises the tensor.
        if (e->task != T_ANYMAL) return fail("dof_state_lag: an AnymalTerrain option");
        if (value != 0 && e->v.dof_api == nullptr) memcpy(e->dof_api_arena, e->v.dof, (size_t)2 * kTasks[e->task].nd * e->N * sizeof(float));
        e->v.dof_api = value != 0 ? e->dof_api_arena : nullptr; return 0;
    }
    if (!strcmp(key, "hand_body_mass")) {       // ShadowHand: the sub-step reads the per-body link-mass factors of `hand_body_mass_scale` (0, default: it does not)
        if (e->task != T_SHADOWHAND) return fail("hand_body_mass: a ShadowHand option (the Allegro hand's kernels take one mass factor per env)");
        e->hv.body_mass = value != 0 ? e->hv.body_mass_arena : nullptr; return 0;
    }
    if (!strcmp(key, "hand_pair_stiffness")) {  // hands: N/m of the compliant hand-to-hand contact pairs; 0 = the pairs off
        if (!is_hand_task(e->task)) return fail("hand_pair_stiffness: a hand-task option");
        if (!(value >= 0)) return fail("hand_pair_stiffness: >= 0");
        e->hv.pair_k = (float)value; return 0;
    }
    if (!strcmp(key, "drive_force_limit")) {
        if (!is_hand_task(e->task)) return fail("drive_force_limit: a hand-task option");
        e->hv.drive_clamp = value != 0 ? 1 : 0; return 0;
    }
    if (!strcmp(key, "terrain_slope_threshold")) {   // terrain.slopeTreshold of the mesh generator (anymal_terrain.py:576); 0 = off
        if (e->task != T_ANYMAL) return fail("terrain_slope_threshold: only AnymalTerrain has a terrain");
        e->terrain.slope_threshold = (float)value;
        return 0;
    }
    if (!strcmp(key, "terrain_walls")) {
        if (e->task != T_ANYMAL) return fail("terrain_walls: only AnymalTerrain has a terrain");
        e->terrain.walls = value != 0 ? 1 : 0;
        return 0;
    }
    if (!strcmp(key, "actor_tensors")) {
        if (is_hand_task(e->task)) return 0;         // the hands always read their own factor tensors
        if (value != 0 && e->actor_scale_arena == nullptr) return fail("actor_tensors: this task carries no actor_scale / dof_limit_shift tensors");
        e->v.actor_scale = value != 0 ? e->actor_scale_arena : nullptr;
        e->v.limit_shift = value != 0 ? e->limit_shift_arena : nullptr;
        return 0;
    }
    if (!strcmp(key, "steps")) { if (value < 0) return fail("steps < 0"); e->steps = (unsigned long long)value; return 0; }
    // sim.physx.num_threads of the reference's CPU pipeline (vec_task.py:541, cfg/config.yaml:30): OpenMP threads over envs
    if (!strcmp(key, "num_threads")) { if (value < 1) return fail("num_threads < 1"); e->num_threads = (int)value; return 0; }
    return fail(std::string("unknown option: ") + key);
}
static_assert(sizeof(MiNoiseParams) == sizeof(NoiseParams), "MiNoiseParams layout");
extern "C" int mi_engine_set_noise(MiEngine* e, int which, const MiNoiseParams* p) {
    if (!e || !p) return fail("mi_engine_set_noise: null argument");
    if (which != 0 && which != 1) return fail("mi_engine_set_noise: which must be 0 (observations) or 1 (actions)");
    if (p->dist < 0 || p->dist > 2 || p->op < 0 || p->op > 1) return fail("mi_engine_set_noise: dist in {0,1,2}, op in {0,1}");
    if (e->task != T_CARTPOLE && e->task != T_ANT && e->task != T_HUMANOID && !is_hand_task(e->task) && p->dist != 0)
        return fail("mi_engine_set_noise: in-kernel noise exists for Cartpole, Ant, Humanoid and ShadowHand");
    memcpy(which == 0 ? &e->v.obs_noise : &e->v.act_noise, p, sizeof(NoiseParams));
    return 0;
}
extern "C" int mi_engine_get_option(const MiEngine* e, const char* key, double* out) {
    if (!e || !key || !out) return fail("mi_engine_get_option: null argument");
    if (!strcmp(key, "clip_obs")) { *out = e->clip_obs; return 0; }
    if (!strcmp(key, "gravity_x")) { *out = e->P.g[0]; return 0; }
    if (!strcmp(key, "gravity_y")) { *out = e->P.g[1]; return 0; }
    if (!strcmp(key, "gravity_z")) { *out = e->P.g[2]; return 0; }
    if (!strcmp(key, "control_freq_inv")) { *out = e->control_freq_inv; return 0; }
    if (!strcmp(key, "self_collision")) { *out = e->v.lamp != nullptr ? 1.0 : 0.0; return 0; }
    if (!strcmp(key, "multi_wave") || !strcmp(key, "fused_sub") || !strcmp(key, "fused_post")) { *out = 0; return 0; }
    if (!strcmp(key, "drive_force_limit")) { *out = is_hand_task(e->task) ? e->hv.drive_clamp : 0; return 0; }
    if (!strcmp(key, "dof_state_lag")) { *out = (e->task == T_ANYMAL && e->v.dof_api != nullptr) ? 1 : 0; return 0; }
    if (!strcmp(key, "hand_body_mass")) { *out = (is_hand_task(e->task) && e->hv.body_mass != nullptr) ? 1 : 0; return 0; }
    if (!strcmp(key, "hand_pair_stiffness")) { *out = is_hand_task(e->task) ? e->hv.pair_k : 0; return 0; }
    if (!strcmp(key, "terrain_slope_threshold")) { *out = e->terrain.slope_threshold; return 0; }
    if (!strcmp(key, "terrain_walls")) { *out = e->terrain.walls; return 0; }
    if (!strcmp(key, "actor_tensors")) { *out = (is_hand_task(e->task) || e->v.actor_scale != nullptr) ? 1.0 : 0.0; return 0; }
    if (!strcmp(key, "steps")) { *out = (double)e->steps; return 0; }
    if (!strcmp(key, "num_threads")) { *out = e->num_threads; return 0; }
    return fail(std::string("unknown option: ") + key);
}
extern "C" int mi_engine_last_ring(const MiEngine* e) { return e ? (int)((e->steps + 1) & 1) : -1; }
// the terrain arrays are copied: the engine does not depend on the caller keeping its buffers alive
extern "C" int mi_engine_set_terrain(MiEngine* e, const int16_t* height_samples, int rows, int cols, float horizontal_scale, float vertical_scale,
                                     float border_size, const float* env_origins, int num_levels, int num_terrains, float env_length, int max_init_level) {
    if (!e) return fail("null engine");
    if (e->task != T_ANYMAL) return fail("mi_engine_set_terrain: only AnymalTerrain uses a terrain");
    if (!height_samples || !env_origins || rows < 2 || cols < 2 || num_levels < 1 || num_terrains < 1 || horizontal_scale <= 0.f)
        return fail("mi_engine_set_terrain: bad argument");
    e->terrain_hs.assign(height_samples, height_samples + (size_t)rows * cols);
    e->terrain_origins.assign(env_origins, env_origins + (size_t)num_levels * num_terrains * 3);
    e->terrain.hs = e->terrain_hs.data(); e->terrain.rows = rows; e->terrain.cols = cols;
    e->terrain.hscale = horizontal_scale; e->terrain.vscale = vertical_scale; e->terrain.border = border_size;
    e->terrain.origins = e->terrain_origins.data(); e->terrain.levels = num_levels; e->terrain.types = num_terrains;
    e->terrain.env_length = env_length;
    e->max_init_level = max_init_level < 0 ? 0 : (max_init_level >= num_levels ? num_levels - 1 : max_init_level);
    return 0;
}
static int hand_call(MiEngine* e, int what, const float* actions, bool simulate_only, const int64_t* ids, int n, float* out_j, float* out_h) {
    const bool sh = e->task == T_SHADOWHAND;
    switch (what) {
        case 0: return sh ? cpu_hand0_init(e) : cpu_hand1_init(e);
        case 1: return sh ? cpu_hand0_step(e, actions, simulate_only) : cpu_hand1_step(e, actions, simulate_only);
        case 2: return sh ? cpu_hand0_reset(e, ids, n) : cpu_hand1_reset(e, ids, n);
        case 3: return sh ? cpu_hand0_body_states(e) : cpu_hand1_body_states(e);
        default: return sh ? cpu_hand0_kinematics(e, out_j, out_h) : cpu_hand1_kinematics(e, out_j, out_h);
    }
}

// ------------------------------------------------------------------------------------------------ initial state (= init_state_kernel)
extern "C" int mi_engine_init_state(MiEngine* e, void*) {
    if (!e) return fail("null engine");
    const TaskMeta& m = kTasks[e->task];
    const View& v = e->v;
    const int N = v.N, nd = m.nd;
    if (is_hand_task(e->task)) {
        e->steps = 0;
        const int rc = hand_call(e, 0, nullptr, false, nullptr, 0, nullptr, nullptr);
        return rc == -2 ? fail("mi_engine_init_state: the hand model of this library and its task table have different sizes (a variant built without its engine unit)") : rc;
    }
    if (e->task == T_ANYMAL && e->terrain.hs == nullptr) return fail("mi_engine_init_state: AnymalTerrain needs mi_engine_set_terrain first");
    const bool loco = e->task == T_ANT || e->task == T_HUMANOID;
    const float root_z = e->task == T_CARTPOLE ? 2.0f : e->task == T_QUADCOPTER ? e->quad.init_height : e->task == T_INGENUITY ? e->ing.init_height
                       : e->task == T_BALLBALANCE ? e->bbot.tray_height : e->task == T_ANYMAL ? e->anymal.base_init_state[2]
                       : e->task == T_ANYMAL_FLAT ? e->anymal_flat.base_init_state[2]
                       : e->task == T_ARTICULATION ? reinterpret_cast<const MiArticulationParams*>(e->artic)->init_root[2] : e->loco.start_height;      // cartpole.py:93 / ant.py:164 / the tasks' default poses
    const float pot0 = loco ? -1000.f / e->loco.dt : 0.f;                                   // ant.py:113
    const float root[13] = {0, 0, root_z, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    for (int en = 0; en < N; ++en) {
        for (int k = 0; k < 13; ++k) { v.root[k * N + en] = root[k]; v.init_root[k * N + en] = root[k]; }
        for (int k = 0; k < nd; ++k) {
            v.dof[k * N + en] = loco ? e->loco.initial_dof_pos[k] : 0.f;
            v.dof[(nd + k) * N + en] = 0.f;
            v.tau[k * N + en] = 0.f; v.laml[k * N + en] = 0.f; v.dof_force[k * N + en] = 0.f;
        }
        for (int k = 0; k < 3 * m.nsph; ++k) v.lamc[k * N + en] = 0.f;
        if (v.lamp) for (int k = 0; k < 3 * ModelHumanoid::NPG; ++k) { v.lamp[k * N + en] = 0.f; v.pairf[k * N + en] = 0.f; }
        if (v.dropped) { v.dropped[en] = 0; v.dropped[N + en] = 0; }
        for (int k = 0; k < 6 * m.nsens; ++k) v.sensor[k * N + en] = 0.f;
        for (int k = 0; k < m.nact; ++k) v.actions[k * N + en] = 0.f;
        for (int k = 0; k < m.nobs; ++k) { v.obs[(size_t)en * m.nobs + k] = 0.f; v.obs_out[(size_t)en * m.nobs + k] = 0.f; v.obs_out[((size_t)N + en) * m.nobs + k] = 0.f; }
        v.potentials[en] = pot0; v.prev_potentials[en] = pot0;
        if (v.friction) v.friction[en] = -1.f;
        if (e->actor_scale_arena) for (int k = 0; k < v.nas; ++k) e->actor_scale_arena[k * N + en] = 1.f;
        if (e->limit_shift_arena) for (int k = 0; k < 2 * nd; ++k) e->limit_shift_arena[k * N + en] = 0.f;
        for (int k = 0; k < 3; ++k) { v.up_vec[k * N + en] = (k == 2) ? 1.f : 0.f; v.heading_vec[k * N + en] = (k == 0) ? 1.f : 0.f; }
        v.rew[en] = 0.f;
        v.reset[en] = 1;      // vec_task.py:316-317: every env is reset inside the first step()
        v.progress[en] = 0; v.randomize[en] = 0; v.timeout[en] = 0; v.episode[en] = 0;
        v.ep_ret[en] = 0.f;
    }
    for (int k = 0; k < 8; ++k) v.stats[k] = 0.f;
    for (int en = 0; en < N; ++en) {      // the tasks' own tensors (quad_init_kernel / ing_init_kernel / bbot_init_kernel)
        if (e->task == T_QUADCOPTER) {
            for (int d = 0; d < kQuadDof; ++d) e->qv.targets[d * N + en] = 0.f;
            for (int k = 0; k < kQuadRotors; ++k) e->qv.thrusts[k * N + en] = 0.f;
            for (int k = 0; k < 3 * ModelQuadcopter::NB; ++k) e->qv.forces[k * N + en] = 0.f;
        } else if (e->task == T_INGENUITY) {
            for (int k = 0; k < 13; ++k) e->iv.marker[k * N + en] = root[k];
            for (int k = 0; k < 3; ++k) e->iv.target[k * N + en] = (k == 2) ? 1.f : 0.f;
            for (int k = 0; k < 3 * kIngRotors; ++k) e->iv.thrusts[k * N + en] = 0.f;
            for (int k = 0; k < 3 * kIngBodies; ++k) e->iv.forces[k * N + en] = 0.f;
        } else if (e->task == T_BALLBALANCE) {
            for (int k = 0; k < 13; ++k) e->bv.ball[k * N + en] = (k < 3) ? e->bbot.ball_init_pos[k] : (k == 6 ? 1.f : 0.f);
            for (int d = 0; d < kBbotDof; ++d) e->bv.targets[d * N + en] = 0.f;
            for (int k = 0; k < 9; ++k) e->bv.lamp[k * N + en] = 0.f;
            e->bv.ncontact[en] = 0;
        }
    }
    e->steps = 0;
    if (e->task == T_ANYMAL || e->task == T_ANYMAL_FLAT) {
        std::string err;
        if (cpu_anymal_init(e, &err) != 0) return fail(err);
    }
    if (e->task == T_ARTICULATION) cpu_articulation_reset(e, nullptr, 0);
    return 0;
}

// ------------------------------------------------------------------------------------------------ one env on the host
template <class M>
static inline void load_env(Sim<M>& s, const View& v, int en) {
    const int N = v.N;
    for (int k = 0; k < 13; ++k) s.root[k] = v.root[k * N + en];
    for (int k = 0; k < M::ND; ++k) { s.q[k] = v.dof[k * N + en]; s.qd[k] = v.dof[(M::ND + k) * N + en]; }
}
template <class M>
static inline void store_env(const Sim<M>& s, const View& v, int en) {
    const int N = v.N;
    for (int k = 0; k < 13; ++k) v.root[k * N + en] = s.root[k];
    for (int k = 0; k < M::ND; ++k) { v.dof[k * N + en] = s.q[k]; v.dof[(M::ND + k) * N + en] = s.qd[k]; }
}
// gym.simulate(): `substeps` sub-steps with the efforts in tau (what substep_kernel does per lane)
template <class M>
static void simulate_env(const View& v, const SimParams& P, int en, const float* tau) {
    if constexpr (M::ACTOR_SCALES != 0 && !is_scaled<M>::value) {
        // option actor_tensors: the instantiation that reads the `actor_params` factors, as on the device (step_kernels.hpp launch_substeps)
        if (v.actor_scale != nullptr || v.limit_shift != nullptr) return simulate_env<Scaled<M>>(v, P, en, tau);
    }
    const int N = v.N;
    Sim<M> sim;
    load_env(sim, v, en);
    if constexpr (Sim<M>::SCALED) {
        if (v.actor_scale) sim.actor_scale = Strided{v.actor_scale + en, N};
        if (v.limit_shift) sim.limit_shift = Strided{v.limit_shift + en, N};
    }
    const float h = P.dt / (float)P.substeps;
    float rows[Sim<M>::ROW_SLOTS > 0 ? Sim<M>::ROW_SLOTS : 1];
    const SelfCol sc{Strided{v.lamp ? v.lamp + en : nullptr, N}, Strided{v.pairf ? v.pairf + en : nullptr, N}, v.dropped ? v.dropped + en : nullptr, N};
    const float mu_env = v.friction ? v.friction[en] : -1.f;
    for (int ss = 0; ss < P.substeps; ++ss)
        sim.substep(P, tau, h, RowStore<1>{rows}, Strided{v.lamc + en, N}, Strided{v.laml + en, N}, Strided{v.sensor + en, N},
                    Strided{v.dof_force + en, N}, PlaneGround{}, mu_env, Strided{nullptr, N}, nullptr, false,
                    (Sim<M>::NPG > 0 && v.lamp) ? &sc : nullptr);
    store_env(sim, v, en);
}
// post_physics_step of Ant / Humanoid for one env (what loco_post_kernel does per lane; reference ant.py:287-297)
template <class M, bool HUM>
static void loco_post_env(const View& v, const LocoParams& tp, int en, StatAcc& acc) {
    using T = Loco<M::ND, 6 * M::NSENS, HUM>;
    constexpr int ND = M::ND, NOBS = T::NOBS;
    const int N = v.N;
    float root[13], q[ND], qd[ND], dof_force[ND], sensor[6 * M::NSENS > 0 ? 6 * M::NSENS : 1], act[ND];
    for (int k = 0; k < 13; ++k) root[k] = v.root[k * N + en];
    for (int k = 0; k < ND; ++k) {
        q[k] = v.dof[k * N + en]; qd[k] = v.dof[(ND + k) * N + en];
        dof_force[k] = v.dof_force[k * N + en]; act[k] = v.actions[k * N + en];
    }
    for (int k = 0; k < 6 * M::NSENS; ++k) sensor[k] = v.sensor[k * N + en];
    long long progress = v.progress[en] + 1;
    float potentials = v.potentials[en], prev_potentials;
    int ep = v.episode[en];
    if (v.reset[en] != 0) {
        float init_root[13];
        for (int k = 0; k < 13; ++k) init_root[k] = v.init_root[k * N + en];
        T::reset(tp, v.seed, (uint32_t)(v.env_offset + en), (uint32_t)ep, init_root, root, q, qd, &potentials, &prev_potentials);
        ep += 1;
        progress = 0;
        for (int k = 0; k < 13; ++k) v.root[k * N + en] = root[k];
        for (int k = 0; k < ND; ++k) { v.dof[k * N + en] = q[k]; v.dof[(ND + k) * N + en] = qd[k]; v.laml[k * N + en] = 0.f; }
        for (int k = 0; k < 3 * M::NSPH; ++k) v.lamc[k * N + en] = 0.f;
        if (M::NPG > 0 && v.lamp) for (int k = 0; k < 3 * M::NPG; ++k) v.lamp[k * N + en] = 0.f;
    }
    float obs[NOBS], up_vec[3], heading_vec[3];
    T::observations(tp, root, tp.targets, potentials, tp.inv_start_rot, q, qd, dof_force, tp.dof_lower, tp.dof_upper, sensor, act,
                    tp.basis_vec0, tp.basis_vec1, obs, &potentials, &prev_potentials, up_vec, heading_vec);
    float rew;
    long long reset;
    T::reward(tp, obs, 0LL, progress, act, potentials, prev_potentials, &rew, &reset);
    if (v.obs_noise.dist != 0)
        for (int k = 0; k < NOBS; ++k) obs[k] = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + en), v.step, 0u, (uint32_t)k, obs[k]);
    episode_stats_env(v, en, rew, reset, progress, acc);
    v.randomize[en] += 1;
    v.episode[en] = ep;
    v.potentials[en] = potentials; v.prev_potentials[en] = prev_potentials;
    for (int k = 0; k < 3; ++k) { v.up_vec[k * N + en] = up_vec[k]; v.heading_vec[k * N + en] = heading_vec[k]; }
    float* ob = v.obs + (size_t)en * NOBS;
    float* oc = v.obs_out + ((size_t)v.ring * N + en) * NOBS;
    for (int k = 0; k < NOBS; ++k) { ob[k] = obs[k]; oc[k] = fminf(fmaxf(obs[k], -v.clip_obs), v.clip_obs); }
    v.rew[en] = rew; v.reset[en] = reset; v.progress[en] = progress;
    v.timeout[en] = (unsigned char)(((float)progress >= tp.max_episode_length - 1.f) && (reset != 0));   // vec_task.py:394
}
static void cartpole_post_env(const View& v, const CartpoleParams& tp, int en, StatAcc& acc) {
    const int N = v.N;
    float q[2] = {v.dof[en], v.dof[N + en]}, qd[2] = {v.dof[2 * N + en], v.dof[3 * N + en]};
    long long progress = v.progress[en] + 1;                   // cartpole.py:165-174
    int ep = v.episode[en];
    if (v.reset[en] != 0) {
        cartpole_reset(v.seed, (uint32_t)(v.env_offset + en), (uint32_t)ep, q, qd);
        ep += 1;
        progress = 0;
        for (int k = 0; k < 2; ++k) { v.dof[k * N + en] = q[k]; v.dof[(2 + k) * N + en] = qd[k]; v.laml[k * N + en] = 0.f; }
    }
    float obs[4] = {q[0], qd[0], q[1], qd[1]};                 // cartpole.py:131-142
    float rew;
    long long reset;
    cartpole_reward(tp, obs[2], obs[3], obs[1], obs[0], 0LL, progress, &rew, &reset);
    if (v.obs_noise.dist != 0)
        for (int k = 0; k < 4; ++k) obs[k] = apply_noise(v.obs_noise, v.seed, (uint32_t)(v.env_offset + en), v.step, 0u, (uint32_t)k, obs[k]);
    episode_stats_env(v, en, rew, reset, progress, acc);
    v.randomize[en] += 1;
    v.episode[en] = ep;
    float* ob = v.obs + (size_t)en * 4;
    float* oc = v.obs_out + ((size_t)v.ring * N + en) * 4;
    for (int k = 0; k < 4; ++k) { ob[k] = obs[k]; oc[k] = fminf(fmaxf(obs[k], -v.clip_obs), v.clip_obs); }
    v.rew[en] = rew; v.reset[en] = reset; v.progress[en] = progress;
    v.timeout[en] = (unsigned char)(((float)progress >= tp.max_episode_length - 1.f) && (reset != 0));
}

template <class M>
static void load_tau(const View& v, int en, float* tau) { for (int k = 0; k < M::ND; ++k) tau[k] = v.tau[k * v.N + en]; }

// VecTask.step for one env: clamp -> efforts (pre_physics_step) -> control_freq_inv x simulate -> post_physics_step
template <class M, bool HUM>
static void step_loco(MiEngine* e, const float* actions) {
    const View& v = e->v;
    const int N = v.N;
    const LocoParams& tp = e->loco;
    std::vector<StatAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        float tau[M::NDA];
        for (int k = 0; k < M::ND; ++k) {
            float a = actions[(size_t)en * M::ND + k];
            if (v.act_noise.dist != 0) a = apply_noise(v.act_noise, v.seed, (uint32_t)(v.env_offset + en), v.step, 1u, (uint32_t)k, a);
            a = fminf(fmaxf(a, -tp.clip_actions), tp.clip_actions);                                             // vec_task.py:374
            v.actions[k * N + en] = a;
            tau[k] = a * tp.gear[k] * tp.power_scale;                                                            // ant.py:281-285
            v.tau[k * N + en] = tau[k];
        }
        for (int c = 0; c < e->control_freq_inv; ++c) simulate_env<M>(v, e->P, en, tau);
        loco_post_env<M, HUM>(v, tp, en, accs[omp_get_thread_num()]);
    }
    for (const StatAcc& a : accs) { v.stats[0] += (float)a.fin_ret; v.stats[1] += (float)a.fin_len; v.stats[2] += (float)a.fin; v.stats[3] += (float)a.r; v.stats[4] += (float)a.cnt; }
}
static void step_cartpole(MiEngine* e, const float* actions) {
    using M = ModelCartpole;
    const View& v = e->v;
    const int N = v.N;
    const CartpoleParams& tp = e->cart;
    std::vector<StatAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        float tau[M::NDA];
        float a = actions[en];
        if (v.act_noise.dist != 0) a = apply_noise(v.act_noise, v.seed, (uint32_t)(v.env_offset + en), v.step, 1u, 0u, a);
        a = fminf(fmaxf(a, -tp.clip_actions), tp.clip_actions);
        v.actions[en] = a;
        tau[0] = a * tp.max_push_effort; tau[1] = 0.f;            // cartpole.py:159-163: effort on DoF 0 only
        v.tau[en] = tau[0]; v.tau[N + en] = 0.f;
        for (int c = 0; c < e->control_freq_inv; ++c) simulate_env<M>(v, e->P, en, tau);
        cartpole_post_env(v, tp, en, accs[omp_get_thread_num()]);
    }
    for (const StatAcc& a : accs) { v.stats[0] += (float)a.fin_ret; v.stats[1] += (float)a.fin_len; v.stats[2] += (float)a.fin; v.stats[3] += (float)a.r; v.stats[4] += (float)a.cnt; }
}

// ---- the small tasks: the per-env bodies of kernels_quadcopter.hip / kernels_ingenuity.hip / kernels_ball_balance.hip, one env at a time
static inline void write_obs(const View& v, int en, const float* obs, int nobs) {
    float* ob = v.obs + (size_t)en * nobs;
    float* oc = v.obs_out + ((size_t)v.ring * v.N + en) * nobs;
    for (int k = 0; k < nobs; ++k) { ob[k] = obs[k]; oc[k] = fminf(fmaxf(obs[k], -v.clip_obs), v.clip_obs); }
}
static inline void clamp_angular_speed(float* root, float lim) {      // asset_options.max_angular_velocity
    const float w2 = root[10] * root[10] + root[11] * root[11] + root[12] * root[12];
    if (w2 > lim * lim) { const float sc = lim / sqrtf(w2); root[10] *= sc; root[11] *= sc; root[12] *= sc; }
}
template <class M, int NR>
static void drive_substeps(const View& v, const SimParams& P, int en, int n_sub, float kp, float kd, const float* target, const float* fs, float wmax) {
    const int N = v.N;
    Sim<M> sim;
    load_env(sim, v, en);
    float tau[M::NDA] = {0}, rows[Sim<M>::ROW_SLOTS > 0 ? Sim<M>::ROW_SLOTS : 1];
    const Drive drv{kp, kd, target, fs};
    const float h = P.dt / (float)P.substeps;
    for (int ss = 0; ss < n_sub; ++ss) {
        sim.substep(P, tau, h, RowStore<1>{rows}, Strided{v.lamc + en, N}, Strided{v.laml + en, N}, Strided{v.sensor + en, N},
                    Strided{v.dof_force + en, N}, PlaneGround{}, -1.f, Strided{nullptr, N}, &drv);
        clamp_angular_speed(sim.root, wmax);
    }
    store_env(sim, v, en);
}
static void quad_reset_env(MiEngine* e, int en) {
    const View& v = e->v;
    const int N = v.N;
    float root[13], q[kQuadDof], qd[kQuadDof];
    const int ep = v.episode[en];
    quadcopter_reset(e->quad, v.seed, (uint32_t)(v.env_offset + en), (uint32_t)ep, root, q, qd);
    for (int k = 0; k < 13; ++k) v.root[k * N + en] = root[k];
    for (int d = 0; d < kQuadDof; ++d) { v.dof[d * N + en] = q[d]; v.dof[(kQuadDof + d) * N + en] = 0.f; v.laml[d * N + en] = 0.f; }
    v.episode[en] = ep + 1; v.reset[en] = 0; v.progress[en] = 0;
}
static void step_quadcopter(MiEngine* e, const float* actions, bool simulate_only) {
    using QM = ModelQuadcopter;
    const View& v = e->v;
    const QuadView& qv = e->qv;
    const QuadcopterParams& p = e->quad;
    const int N = v.N;
    std::vector<StatAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        if (!simulate_only) {          // pre_physics_step (quadcopter.py:276-292)
            const bool rs = v.reset[en] != 0;
            if (rs) quad_reset_env(e, en);
            for (int d = 0; d < kQuadDof; ++d) {
                const float a = fminf(fmaxf(actions[(size_t)en * kQuadAct + d], -p.clip_actions), p.clip_actions);
                v.actions[d * N + en] = a;
                float t = qv.targets[d * N + en] + p.dt * p.dof_action_speed_scale * a;
                t = fmaxf(fminf(t, p.dof_upper[d]), p.dof_lower[d]);
                if (rs) t = v.dof[d * N + en];
                qv.targets[d * N + en] = t;
            }
            for (int k = 0; k < kQuadRotors; ++k) {
                const float a = fminf(fmaxf(actions[(size_t)en * kQuadAct + kQuadDof + k], -p.clip_actions), p.clip_actions);
                v.actions[(kQuadDof + k) * N + en] = a;
                float th = qv.thrusts[k * N + en] + p.dt * p.thrust_action_speed_scale * a;
                th = fmaxf(fminf(th, p.max_thrust), 0.f);
                qv.thrusts[k * N + en] = rs ? 0.f : th;
                qv.forces[(3 * QM::sens_body[k] + 2) * N + en] = rs ? 0.f : th;
            }
        }
        float target[kQuadDof], fs[kQuadRotors][3];
        for (int d = 0; d < kQuadDof; ++d) target[d] = qv.targets[d * N + en];
        for (int k = 0; k < kQuadRotors; ++k) { fs[k][0] = fs[k][1] = 0.f; fs[k][2] = qv.forces[(3 * QM::sens_body[k] + 2) * N + en]; }
        drive_substeps<QM, kQuadRotors>(v, e->P, en, (simulate_only ? 1 : e->control_freq_inv) * e->P.substeps, p.drive_stiffness, p.drive_damping, target, &fs[0][0], p.max_angular_velocity);
        if (simulate_only) continue;
        float root[13], q[kQuadDof], obs[kQuadObs], rew;     // post_physics_step (:294-302)
        for (int k = 0; k < 13; ++k) root[k] = v.root[k * N + en];
        for (int d = 0; d < kQuadDof; ++d) q[d] = v.dof[d * N + en];
        const long long progress = v.progress[en] + 1;
        long long reset;
        quadcopter_observations(root, q, obs);
        quadcopter_reward(root, progress, p.max_episode_length, &rew, &reset);
        episode_stats_env(v, en, rew, reset, progress, accs[omp_get_thread_num()]);
        v.randomize[en] += 1;
        write_obs(v, en, obs, kQuadObs);
        v.rew[en] = rew; v.reset[en] = reset; v.progress[en] = progress;
        v.timeout[en] = (unsigned char)(((float)progress >= p.max_episode_length - 1.f) && (reset != 0));
    }
    for (const StatAcc& a : accs) { v.stats[0] += (float)a.fin_ret; v.stats[1] += (float)a.fin_len; v.stats[2] += (float)a.fin; v.stats[3] += (float)a.r; v.stats[4] += (float)a.cnt; }
}
static void ing_set_target(MiEngine* e, int en, const float* t) {
    const int N = e->v.N;
    for (int k = 0; k < 3; ++k) { e->iv.target[k * N + en] = t[k]; e->iv.marker[k * N + en] = t[k] + (k == 2 ? 0.4f : 0.f); }
}
static void ing_reset_env(MiEngine* e, int en) {
    const View& v = e->v;
    const int N = v.N;
    const uint32_t genv = (uint32_t)(v.env_offset + en);
    const int ep = v.episode[en];
    float root[13], target[3];
    ingenuity_target(v.seed, genv, (uint32_t)ep, 3u, target);
    ing_set_target(e, en, target);
    ingenuity_reset_root(e->ing, v.seed, genv, (uint32_t)ep, root);
    for (int k = 0; k < 13; ++k) v.root[k * N + en] = root[k];
    v.dof[(kIngDof + 1) * N + en] = -e->ing.rotor_speed;
    v.dof[(kIngDof + 3) * N + en] = e->ing.rotor_speed;
    for (int d = 0; d < kIngDof; ++d) v.laml[d * N + en] = 0.f;
    v.episode[en] = ep + 1; v.reset[en] = 0; v.progress[en] = 0;
}
static void step_ingenuity(MiEngine* e, const float* actions, bool simulate_only) {
    using IM = ModelIngenuity;
    const View& v = e->v;
    const IngenuityView& iv = e->iv;
    const IngenuityParams& p = e->ing;
    const int N = v.N;
    std::vector<StatAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        if (!simulate_only) {          // pre_physics_step (ingenuity.py:321-354)
            const uint32_t genv = (uint32_t)(v.env_offset + en);
            const long long progress = v.progress[en];
            float target[3];
            if (progress % p.target_period == 0) {
                ingenuity_target(v.seed, genv, (uint32_t)v.episode[en], 8u + 3u * (uint32_t)(progress / p.target_period), target);
                ing_set_target(e, en, target);
            }
            const bool rs = v.reset[en] != 0;
            if (rs) ing_reset_env(e, en);
            float a[kIngAct];
            for (int k = 0; k < kIngAct; ++k) { a[k] = fminf(fmaxf(actions[(size_t)en * kIngAct + k], -p.clip_actions), p.clip_actions); v.actions[k * N + en] = a[k]; }
            for (int r = 0; r < kIngRotors; ++r) {
                const float vertical = fminf(fmaxf(a[3 * r + 2] * p.thrust_action_speed_scale, -p.thrust_upper_limit), p.thrust_upper_limit);
                float th[3];
                th[2] = p.dt * vertical;
                for (int k = 0; k < 2; ++k) th[k] = th[2] * fminf(fmaxf(a[3 * r + k], -p.thrust_lateral_component), p.thrust_lateral_component);
                for (int k = 0; k < 3; ++k) {
                    const float x = rs ? 0.f : th[k];
                    iv.thrusts[(3 * r + k) * N + en] = x;
                    iv.forces[(3 * IM::sens_body[r] + k) * N + en] = x;
                }
            }
        }
        float target[kIngDof] = {0}, fs[kIngRotors][3];
        for (int r = 0; r < kIngRotors; ++r) for (int k = 0; k < 3; ++k) fs[r][k] = iv.forces[(3 * IM::sens_body[r] + k) * N + en];
        drive_substeps<IM, kIngRotors>(v, e->P, en, (simulate_only ? 1 : e->control_freq_inv) * e->P.substeps, 0.f, 0.f, target, &fs[0][0], p.max_angular_velocity);
        if (simulate_only) continue;
        float root[13], tg[3], obs[kIngObs], rew;          // post_physics_step (:356-365)
        for (int k = 0; k < 13; ++k) root[k] = v.root[k * N + en];
        for (int k = 0; k < 3; ++k) tg[k] = iv.target[k * N + en];
        const long long progress = v.progress[en] + 1;
        long long reset;
        ingenuity_observations(root, tg, obs);
        ingenuity_reward(root, tg, root + 3, root + 10, progress, p.max_episode_length, &rew, &reset);
        episode_stats_env(v, en, rew, reset, progress, accs[omp_get_thread_num()]);
        v.randomize[en] += 1;
        write_obs(v, en, obs, kIngObs);
        v.rew[en] = rew; v.reset[en] = reset; v.progress[en] = progress;
        v.timeout[en] = (unsigned char)(((float)progress >= p.max_episode_length - 1.f) && (reset != 0));
    }
    for (const StatAcc& a : accs) { v.stats[0] += (float)a.fin_ret; v.stats[1] += (float)a.fin_len; v.stats[2] += (float)a.fin; v.stats[3] += (float)a.r; v.stats[4] += (float)a.cnt; }
}
static void bbot_reset_env(MiEngine* e, int en) {
    const View& v = e->v;
    const int N = v.N;
    const int ep = v.episode[en];
    float ball[13];
    bbot_reset_ball(e->bbot, v.seed, (uint32_t)(v.env_offset + en), (uint32_t)ep, ball);
    for (int k = 0; k < 13; ++k) { v.root[k * N + en] = v.init_root[k * N + en]; e->bv.ball[k * N + en] = ball[k]; }
    for (int d = 0; d < kBbotDof; ++d) { v.dof[d * N + en] = 0.f; v.dof[(kBbotDof + d) * N + en] = 0.f; v.laml[d * N + en] = 0.f; }
    for (int k = 0; k < 9; ++k) e->bv.lamp[k * N + en] = 0.f;
    v.episode[en] = ep + 1; v.reset[en] = 0; v.progress[en] = 0;
}
static void step_ball_balance(MiEngine* e, const float* actions, bool simulate_only) {
    using BM = ModelBalanceBot;
    const View& v = e->v;
    const BbotView& bv = e->bv;
    const BallBalanceParams& p = e->bbot;
    const BbotPhys& ph = *reinterpret_cast<const BbotPhys*>(&p.pin_stiffness);
    static_assert(sizeof(BbotPhys) == sizeof(BallBalanceParams) - offsetof(BallBalanceParams, pin_stiffness), "BbotPhys is the tail of BallBalanceParams");
    const int N = v.N;
    std::vector<StatAcc> accs(e->num_threads);
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        if (!simulate_only) {          // pre_physics_step (ball_balance.py:395-413)
            const bool rs = v.reset[en] != 0;
            if (rs) bbot_reset_env(e, en);
            float a[kBbotAct];
            for (int k = 0; k < kBbotAct; ++k) { a[k] = fminf(fmaxf(actions[(size_t)en * kBbotAct + k], -p.clip_actions), p.clip_actions); v.actions[k * N + en] = a[k]; }
            for (int d = 0; d < kBbotDof; ++d) {
                float t = bv.targets[d * N + en];
                if (d & 1) t += p.dt * p.action_speed_scale * a[d >> 1];
                t = fmaxf(fminf(t, p.dof_upper[d]), p.dof_lower[d]);
                bv.targets[d * N + en] = rs ? 0.f : t;
            }
        }
        BbotSim<BM> sim;
        load_env(sim, v, en);
        float target[kBbotDof];
        for (int d = 0; d < kBbotDof; ++d) target[d] = bv.targets[d * N + en];
        for (int k = 0; k < 3; ++k) { sim.ball.pos[k] = bv.ball[k * N + en]; sim.ball.vel[k] = bv.ball[(7 + k) * N + en]; sim.ball.angvel[k] = bv.ball[(10 + k) * N + en]; }
        for (int k = 0; k < 4; ++k) sim.ball.quat[k] = bv.ball[(3 + k) * N + en];
        int nc = 0;
        const int n_sub = (simulate_only ? 1 : e->control_freq_inv) * e->P.substeps;
        for (int ss = 0; ss < n_sub; ++ss)
            sim.substep(e->P, ph, e->P.dt / (float)e->P.substeps, target, Strided{v.laml + en, N}, Strided{bv.lamp + en, N}, Strided{v.sensor + en, N}, &nc);
        bv.ncontact[en] = nc;
        store_env(sim, v, en);
        for (int k = 0; k < 3; ++k) { bv.ball[k * N + en] = sim.ball.pos[k]; bv.ball[(7 + k) * N + en] = sim.ball.vel[k]; bv.ball[(10 + k) * N + en] = sim.ball.angvel[k]; }
        for (int k = 0; k < 4; ++k) bv.ball[(3 + k) * N + en] = sim.ball.quat[k];
        if (simulate_only) continue;
        float q[kBbotDof], qd[kBbotDof], ball[13], sens[18], obs[kBbotObs], rew;     // post_physics_step (:415-424)
        for (int d = 0; d < kBbotDof; ++d) { q[d] = v.dof[d * N + en]; qd[d] = v.dof[(kBbotDof + d) * N + en]; }
        for (int k = 0; k < 13; ++k) ball[k] = bv.ball[k * N + en];
        for (int k = 0; k < 18; ++k) sens[k] = v.sensor[k * N + en];
        const long long progress = v.progress[en] + 1;
        long long reset;
        bbot_observations(q, qd, ball, sens, obs);
        bbot_reward(ball, ball + 7, p.ball_radius, v.reset[en], progress, p.max_episode_length, &rew, &reset);
        episode_stats_env(v, en, rew, reset, progress, accs[omp_get_thread_num()]);
        v.randomize[en] += 1;
        write_obs(v, en, obs, kBbotObs);
        v.rew[en] = rew; v.reset[en] = reset; v.progress[en] = progress;
        v.timeout[en] = (unsigned char)(((float)progress >= p.max_episode_length - 1.f) && (reset != 0));
    }
    for (const StatAcc& a : accs) { v.stats[0] += (float)a.fin_ret; v.stats[1] += (float)a.fin_len; v.stats[2] += (float)a.fin; v.stats[3] += (float)a.r; v.stats[4] += (float)a.cnt; }
}

extern "C" int mi_engine_step(MiEngine* e, const float* actions, void*) {
    if (!e || !actions) return fail("mi_engine_step: null argument");
    e->v.ring = (int)(e->steps & 1);
    e->v.step = (unsigned)e->steps;
    switch (e->task) {
        case T_CARTPOLE: step_cartpole(e, actions); break;
        case T_ANT: step_loco<ModelAnt, false>(e, actions); break;
        case T_HUMANOID: step_loco<ModelHumanoid, true>(e, actions); break;
        case T_QUADCOPTER: step_quadcopter(e, actions, false); break;
        case T_INGENUITY: step_ingenuity(e, actions, false); break;
        case T_BALLBALANCE: step_ball_balance(e, actions, false); break;
        case T_ANYMAL:
            if (e->terrain.hs == nullptr) return fail("mi_engine_step: AnymalTerrain needs mi_engine_set_terrain first");
            cpu_anymal_step(e, actions, false);
            break;
        case T_ANYMAL_FLAT: cpu_anymal_step(e, actions, false); break;
        case T_SHADOWHAND: case T_ALLEGROHAND: hand_call(e, 1, actions, false, nullptr, 0, nullptr, nullptr); break;
        case T_ARTICULATION: return fail("mi_engine_step: the Articulation task has no task kernels -- drive it with mi_engine_simulate (gym.simulate) and keep "
                                         "the observation / reward code on the caller's side");
        default: return fail("mi_engine_step: task not on the CPU backend");
    }
    e->steps++;
    return 0;
}
template <class M>
static void simulate_all(MiEngine* e) {
    const View& v = e->v;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < v.N; ++en) {
        float tau[M::NDA];
        load_tau<M>(v, en, tau);
        simulate_env<M>(v, e->P, en, tau);
    }
}
extern "C" int mi_engine_simulate(MiEngine* e, void*) {
    if (!e) return fail("null engine");
    switch (e->task) {
        case T_CARTPOLE: simulate_all<ModelCartpole>(e); break;
        case T_ANT: simulate_all<ModelAnt>(e); break;
        case T_HUMANOID: simulate_all<ModelHumanoid>(e); break;
        case T_QUADCOPTER: step_quadcopter(e, nullptr, true); break;
        case T_INGENUITY: step_ingenuity(e, nullptr, true); break;
        case T_BALLBALANCE: step_ball_balance(e, nullptr, true); break;
        case T_ANYMAL:
            if (e->terrain.hs == nullptr) return fail("mi_engine_simulate: AnymalTerrain needs mi_engine_set_terrain first");
            cpu_anymal_step(e, nullptr, true);
            break;
        case T_ANYMAL_FLAT: cpu_anymal_step(e, nullptr, true); break;
        case T_SHADOWHAND: case T_ALLEGROHAND: hand_call(e, 1, nullptr, true, nullptr, 0, nullptr, nullptr); break;
        case T_ARTICULATION: cpu_articulation_simulate(e); break;
        default: return fail("mi_engine_simulate: task not on the CPU backend");
    }
    return 0;
}
// gym.refresh_rigid_body_state_tensor: Sim<M>::body_state for every (env, body)
template <class M>
static void body_states_all(MiEngine* e) {
    const View& v = e->v;
    const int N = v.N;
#pragma omp parallel for schedule(static) num_threads(e->num_threads)
    for (int en = 0; en < N; ++en) {
        Sim<M> sim;
        for (int k = 0; k < 13; ++k) sim.root[k] = v.root[k * N + en];
        for (int k = 0; k < M::ND; ++k) { sim.q[k] = v.dof[k * N + en]; sim.qd[k] = v.dof[(M::ND + k) * N + en]; }
        sfor<M::NB>([&](auto B_) {
            constexpr int b = decltype(B_)::value;
            float o[13];
            sim.template body_state<b>(o);
            for (int k = 0; k < 13; ++k) v.body_state[(b * 13 + k) * N + en] = o[k];
        });
    }
}
extern "C" int mi_engine_refresh_rigid_body_states(MiEngine* e, void*) {
    if (!e) return fail("null engine");
    switch (e->task) {
        case T_CARTPOLE: body_states_all<ModelCartpole>(e); break;
        case T_ANT: body_states_all<ModelAnt>(e); break;
        case T_HUMANOID: body_states_all<ModelHumanoid>(e); break;
        case T_QUADCOPTER: body_states_all<ModelQuadcopter>(e); break;
        case T_INGENUITY: body_states_all<ModelIngenuity>(e); break;
        case T_BALLBALANCE: body_states_all<ModelBalanceBot>(e); break;
        case T_ANYMAL: case T_ANYMAL_FLAT: cpu_anymal_body_states(e); break;
        case T_SHADOWHAND: case T_ALLEGROHAND: hand_call(e, 3, nullptr, false, nullptr, 0, nullptr, nullptr); break;
        case T_ARTICULATION: cpu_articulation_body_states(e); break;
        default: return fail("mi_engine_refresh_rigid_body_states: task not on the CPU backend");
    }
    return 0;
}
// gym.refresh_jacobian_tensors / refresh_mass_matrix_tensors into caller tensors (csrc/kernels_body_states.hip on the device)
template <class M>
static void kinematics_views_all(MiEngine* e, float* out_j, float* out_h) {
    const View& v = e->v;
    const int N = v.N;
    constexpr int NV = M::NV;
#pragma omp parallel for schedule(static)
    for (int en = 0; en < N; ++en) {
        Sim<M> sim;
        load_env(sim, v, en);
        if (out_j) {
            sfor<M::NB>([&](auto B_) {
                constexpr int b = decltype(B_)::value;
                sim.template body_jacobian<b>(out_j + ((size_t)en * M::NB + b) * 6 * NV);
            });
        }
        if (out_h) sim.mass_matrix(e->P, out_h + (size_t)en * NV * NV);
    }
}
static int kinematics_views(MiEngine* e, float* out_j, float* out_h, const char* who) {
    if (!e || (!out_j && !out_h)) return fail(std::string(who) + ": null argument");
    switch (e->task) {
        case T_CARTPOLE: kinematics_views_all<ModelCartpole>(e, out_j, out_h); break;
        case T_ANT: kinematics_views_all<ModelAnt>(e, out_j, out_h); break;
        case T_HUMANOID: kinematics_views_all<ModelHumanoid>(e, out_j, out_h); break;
        case T_QUADCOPTER: kinematics_views_all<ModelQuadcopter>(e, out_j, out_h); break;
        case T_INGENUITY: kinematics_views_all<ModelIngenuity>(e, out_j, out_h); break;
        case T_BALLBALANCE: kinematics_views_all<ModelBalanceBot>(e, out_j, out_h); break;
        case T_ANYMAL: case T_ANYMAL_FLAT: cpu_anymal_kinematics(e, out_j, out_h); break;
        case T_SHADOWHAND: case T_ALLEGROHAND: hand_call(e, 4, nullptr, false, nullptr, 0, out_j, out_h); break;
        case T_ARTICULATION: cpu_articulation_kinematics(e, out_j, out_h); break;
        default: return fail(std::string(who) + ": task not on the CPU backend");
    }
    return 0;
}
extern "C" int mi_engine_compute_jacobians(MiEngine* e, float* out, void*) { return kinematics_views(e, out, nullptr, "mi_engine_compute_jacobians"); }
extern "C" int mi_engine_compute_mass_matrices(MiEngine* e, float* out, void*) { return kinematics_views(e, nullptr, out, "mi_engine_compute_mass_matrices"); }

template <class M, bool HUM>
static void reset_loco(MiEngine* e, const int64_t* ids, int n) {
    using T = Loco<M::ND, 6 * M::NSENS, HUM>;
    const View& v = e->v;
    const int N = v.N;
    for (int i = 0; i < n; ++i) {
        const int en = (int)ids[i];
        if (en < 0 || en >= N) continue;
        float init_root[13], root[13], q[M::ND], qd[M::ND], pot, prev;
        for (int k = 0; k < 13; ++k) init_root[k] = v.init_root[k * N + en];
        const int ep = v.episode[en];
        T::reset(e->loco, v.seed, (uint32_t)(v.env_offset + en), (uint32_t)ep, init_root, root, q, qd, &pot, &prev);
        v.episode[en] = ep + 1;
        for (int k = 0; k < 13; ++k) v.root[k * N + en] = root[k];
        for (int k = 0; k < M::ND; ++k) { v.dof[k * N + en] = q[k]; v.dof[(M::ND + k) * N + en] = qd[k]; v.laml[k * N + en] = 0.f; }
        for (int k = 0; k < 3 * M::NSPH; ++k) v.lamc[k * N + en] = 0.f;
        if (M::NPG > 0 && v.lamp) for (int k = 0; k < 3 * M::NPG; ++k) v.lamp[k * N + en] = 0.f;
        v.potentials[en] = pot; v.prev_potentials[en] = prev;
        v.progress[en] = 0; v.reset[en] = 0;
    }
}
extern "C" int mi_engine_reset_idx(MiEngine* e, const int64_t* env_ids, int n, void*) {
    if (!e) return fail("null engine");
    if (n <= 0) return 0;
    if (!env_ids) return fail("mi_engine_reset_idx: null env_ids");
    const View& v = e->v;
    const int N = v.N;
    switch (e->task) {
        case T_CARTPOLE:
            for (int i = 0; i < n; ++i) {
                const int en = (int)env_ids[i];
                if (en < 0 || en >= N) continue;
                float q[2], qd[2];
                const int ep = v.episode[en];
                cartpole_reset(v.seed, (uint32_t)(v.env_offset + en), (uint32_t)ep, q, qd);
                v.episode[en] = ep + 1;
                for (int k = 0; k < 2; ++k) { v.dof[k * N + en] = q[k]; v.dof[(2 + k) * N + en] = qd[k]; v.laml[k * N + en] = 0.f; }
                v.progress[en] = 0; v.reset[en] = 0;
            }
            break;
        case T_ANT: reset_loco<ModelAnt, false>(e, env_ids, n); break;
        case T_HUMANOID: reset_loco<ModelHumanoid, true>(e, env_ids, n); break;
        case T_QUADCOPTER: for (int i = 0; i < n; ++i) if (env_ids[i] >= 0 && env_ids[i] < N) quad_reset_env(e, (int)env_ids[i]); break;
        case T_INGENUITY: for (int i = 0; i < n; ++i) if (env_ids[i] >= 0 && env_ids[i] < N) ing_reset_env(e, (int)env_ids[i]); break;
        case T_BALLBALANCE:
            for (int i = 0; i < n; ++i) if (env_ids[i] >= 0 && env_ids[i] < N) {
                bbot_reset_env(e, (int)env_ids[i]);
                for (int d = 0; d < kBbotDof; ++d) e->bv.targets[d * N + (int)env_ids[i]] = 0.f;
            }
            break;
        case T_ANYMAL: case T_ANYMAL_FLAT: cpu_anymal_reset(e, env_ids, n); break;
        case T_SHADOWHAND: case T_ALLEGROHAND: hand_call(e, 2, nullptr, false, env_ids, n, nullptr, nullptr); break;
        case T_ARTICULATION: cpu_articulation_reset(e, env_ids, n); break;
        default: return fail("mi_engine_reset_idx: task not on the CPU backend");
    }
    return 0;
}

// mi_engine.hip -- the C ABI of include/mi_engine.h: arena layout, engine lifecycle, launches.
// The fused step kernels live in step_kernels.hpp and are instantiated per robot in kernels_<task>.hip; this file
// also holds the small stand-alone kernels (init, jit-fn replacements).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mi_engine.h"
#include "step_kernels.hpp"
#include "arena_layout.hpp"
#include "task_views.hpp"
#include "gen/model_ant.h"
#include "gen/model_cartpole.h"
#include "gen/model_humanoid.h"
#include "gen/model_anymal.h"
#include "gen/model_shadow_hand.h"
#include "gen/model_allegro_hand.h"
#include "tasks/anymal.hpp"
#include "gen/model_quadcopter.h"
#include "tasks/quadcopter.hpp"
#include "tasks/shadow_hand.hpp"
#include "tasks/articulation.hpp"

using namespace mi;

static_assert(sizeof(MiSimParams) == sizeof(SimParams), "MiSimParams layout");
static_assert(sizeof(MiLocoParams) == sizeof(LocoParams), "MiLocoParams layout");
static_assert(sizeof(MiCartpoleParams) == sizeof(CartpoleParams), "MiCartpoleParams layout");
static_assert(MI_MAX_DOF == mi::kMaxDof, "MI_MAX_DOF");
static_assert(sizeof(MiAnymalParams) == sizeof(AnymalParams), "MiAnymalParams layout");
static_assert(sizeof(MiAnymalFlatParams) == sizeof(AnymalFlatParams), "MiAnymalFlatParams layout");
static_assert(sizeof(MiQuadcopterParams) == sizeof(QuadcopterParams), "MiQuadcopterParams layout");
static_assert(sizeof(MiIngenuityParams) == sizeof(IngenuityParams), "MiIngenuityParams layout");
static_assert(sizeof(MiBallBalanceParams) == sizeof(BallBalanceParams), "MiBallBalanceParams layout");
static_assert(sizeof(MiHandRewardParams) == sizeof(HandRewardParams), "MiHandRewardParams layout");
static_assert(sizeof(MiHandParams) == sizeof(HandParams), "MiHandParams layout");
static_assert(sizeof(MiArticulationParams) == sizeof(ArticulationParams), "MiArticulationParams layout");

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }
namespace mi { int abi_fail(const char* msg) { return fail(msg); } }   // for the entry points that live in other translation units
extern "C" const char* mi_last_error(void) { return g_err.c_str(); }
extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }

#define HIP_OK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

// ------------------------------------------------------------------------------------------------ init
__global__ void init_state_kernel(View v, int nd, int nsph3, int nsens6, int nobs, int nact, float root_z,
                                  const float* __restrict__ init_dof /* [nd] or null */, float pot0) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    const float root[13] = {0, 0, root_z, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 13; ++k) { v.root[k * N + e] = root[k]; v.init_root[k * N + e] = root[k]; }
    for (int k = 0; k < nd; ++k) {
        v.dof[k * N + e] = init_dof ? init_dof[k] : 0.f;
        v.dof[(nd + k) * N + e] = 0.f;
        v.tau[k * N + e] = 0.f; v.laml[k * N + e] = 0.f; v.dof_force[k * N + e] = 0.f;
    }
    for (int k = 0; k < nsph3; ++k) v.lamc[k * N + e] = 0.f;
    if (v.lamp) for (int k = 0; k < 3 * ModelHumanoid::NPG; ++k) { v.lamp[k * N + e] = 0.f; v.pairf[k * N + e] = 0.f; }
    if (v.dropped) { v.dropped[e] = 0; v.dropped[N + e] = 0; }
    for (int k = 0; k < nsens6; ++k) v.sensor[k * N + e] = 0.f;
    for (int k = 0; k < nact; ++k) v.actions[k * N + e] = 0.f;
    for (int k = 0; k < nobs; ++k) { v.obs[(size_t)e * nobs + k] = 0.f; v.obs_out[(size_t)e * nobs + k] = 0.f; v.obs_out[((size_t)N + e) * nobs + k] = 0.f; }
    v.potentials[e] = pot0; v.prev_potentials[e] = pot0;
    if (v.friction) v.friction[e] = -1.f;       // model friction until somebody writes the tensor (AnymalTerrain's init overwrites it)
    if (v.actor_scale) for (int k = 0; k < v.nas; ++k) v.actor_scale[k * N + e] = 1.f;
    if (v.limit_shift) for (int k = 0; k < 2 * nd; ++k) v.limit_shift[k * N + e] = 0.f;
    for (int k = 0; k < 3; ++k) { v.up_vec[k * N + e] = (k == 2) ? 1.f : 0.f; v.heading_vec[k * N + e] = (k == 0) ? 1.f : 0.f; }
    v.rew[e] = 0.f;
    v.reset[e] = 1;  // vec_task.py:316-317: every env is reset inside the first step()
    v.progress[e] = 0; v.randomize[e] = 0; v.timeout[e] = 0; v.episode[e] = 0;
    v.ep_ret[e] = 0.f;
    if (e == 0) for (int k = 0; k < 8; ++k) v.stats[k] = 0.f;
}

// ------------------------------------------------------------------------------------------------ stand-alone jit-fn replacements
template <int ND, int NSV, bool HUM>
__global__ void loco_obs_kernel(int n, LocoParams p, const float* root_states, const float* targets, float* potentials,
                                float* prev_potentials, const float* inv_start_rot, const float* dof_pos,
                                const float* dof_vel, const float* dof_force, const float* lower, const float* upper,
                                const float* sensors, const float* actions, const float* basis0, const float* basis1,
                                float* obs_buf, float* up_vec, float* heading_vec) {
    using T = Loco<ND, NSV, HUM>;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float root[13], q[ND], qd[ND], df[ND], se[NSV > 0 ? NSV : 1], ac[ND], lo[ND], up[ND], obs[T::NOBS], uv[3], hv[3], pot, prev;
    for (int k = 0; k < 13; ++k) root[k] = root_states[(size_t)e * 13 + k];
    for (int k = 0; k < ND; ++k) {
        q[k] = dof_pos[(size_t)e * ND + k]; qd[k] = dof_vel[(size_t)e * ND + k];
        df[k] = (HUM && dof_force) ? dof_force[(size_t)e * ND + k] : 0.f;
        ac[k] = actions[(size_t)e * ND + k]; lo[k] = lower[k]; up[k] = upper[k];
    }
    for (int k = 0; k < NSV; ++k) se[k] = sensors[(size_t)e * NSV + k];
    T::observations(p, root, targets + (size_t)e * 3, potentials[e], inv_start_rot + (size_t)e * 4, q, qd, df, lo, up, se, ac,
                    basis0 + (size_t)e * 3, basis1 + (size_t)e * 3, obs, &pot, &prev, uv, hv);
    for (int k = 0; k < T::NOBS; ++k) obs_buf[(size_t)e * T::NOBS + k] = obs[k];
    potentials[e] = pot; prev_potentials[e] = prev;
    for (int k = 0; k < 3; ++k) { up_vec[(size_t)e * 3 + k] = uv[k]; heading_vec[(size_t)e * 3 + k] = hv[k]; }
}
template <int ND, int NSV, bool HUM>
__global__ void loco_reward_kernel(int n, LocoParams p, const float* obs_buf, const long long* reset_in,
                                   const long long* progress, const float* actions, const float* potentials,
                                   const float* prev_potentials, float* rew, long long* reset_out) {
    using T = Loco<ND, NSV, HUM>;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float obs[T::NOBS], ac[ND];
    for (int k = 0; k < T::NOBS; ++k) obs[k] = obs_buf[(size_t)e * T::NOBS + k];
    for (int k = 0; k < ND; ++k) ac[k] = actions[(size_t)e * ND + k];
    float r; long long rs;
    T::reward(p, obs, reset_in[e], progress[e], ac, potentials[e], prev_potentials[e], &r, &rs);
    rew[e] = r; reset_out[e] = rs;
}
__global__ void cartpole_reward_kernel(int n, CartpoleParams p, const float* pole_angle, const float* pole_vel,
                                       const float* cart_vel, const float* cart_pos, const long long* reset_in,
                                       const long long* progress, float* rew, long long* reset_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float r; long long rs;
    cartpole_reward(p, pole_angle[e], pole_vel[e], cart_vel[e], cart_pos[e], reset_in[e], progress[e], &r, &rs);
    rew[e] = r; reset_out[e] = rs;
}

// ------------------------------------------------------------------------------------------------ ShadowHand jit-fn replacements
__global__ void hand_reward_kernel(int n, HandRewardParams p, const float* object_pos, const float* object_rot, const float* target_pos,
                                   const float* target_rot, const float* actions, int nact, float* rew, long long* reset_buf,
                                   long long* reset_goal_buf, long long* progress_buf, float* successes, float* ws /* [2] */) {
    const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e0 < n;
    const int e = valid ? e0 : n - 1;
    float r, succ;
    long long rs, gr, prog;
    hand_reward(p, object_pos + (size_t)e * 3, object_rot + (size_t)e * 4, target_pos + (size_t)e * 3, target_rot + (size_t)e * 4,
                actions + (size_t)e * nact, nact, reset_buf[e], reset_goal_buf[e], progress_buf[e], successes[e], &r, &rs, &gr, &prog, &succ);
    float nres = valid ? (float)rs : 0.f, fin = valid ? succ * (float)rs : 0.f;
    for (int o = 32; o > 0; o >>= 1) { nres += __shfl_xor(nres, o, 64); fin += __shfl_xor(fin, o, 64); }
    if ((threadIdx.x & 63) == 0 && nres > 0.f) { atomicAdd(ws, nres); atomicAdd(ws + 1, fin); }
    if (!valid) return;
    rew[e] = r; reset_buf[e] = rs; reset_goal_buf[e] = gr; progress_buf[e] = prog; successes[e] = succ;
}
// consecutive_successes moving average over the whole batch (shadow_hand.py:792-797)
__global__ void hand_reward_finalize_kernel(HandRewardParams p, const float* ws, float* consecutive_successes) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float num_resets = ws[0], finished = ws[1], cs = consecutive_successes[0];
        consecutive_successes[0] = (num_resets > 0.f) ? p.av_factor * finished / num_resets + (1.0f - p.av_factor) * cs : cs;
    }
}
// compute_full_state (shadow_hand.py:528-584): 3*nd | obj pose 7, linvel 3, angvel*s 3 | goal pose 7, quat diff 4 |
// fingertip states 13*nf | fingertip force-torques*s 6*nf | actions
__global__ void hand_full_state_kernel(int n, int nd, int nf, int nact, float vel_obs_scale, float ft_scale, const float* dof_pos,
                                       const float* dof_vel, const float* dof_force, const float* lower, const float* upper,
                                       const float* object_state /* [n,13] */, const float* goal_pose /* [n,7] */,
                                       const float* fingertip_state /* [n,nf,13] */, const float* sensors /* [n,6nf] */,
                                       const float* actions, float* obs, int obs_stride) {
    MI_NO_CONTRACT
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float* o = obs + (size_t)e * obs_stride;
    for (int d = 0; d < nd; ++d) {
        o[d] = (2.0f * dof_pos[(size_t)e * nd + d] - upper[d] - lower[d]) / (upper[d] - lower[d]);   // unscale, torch_jit_utils.py:239
        o[nd + d] = vel_obs_scale * dof_vel[(size_t)e * nd + d];
        o[2 * nd + d] = ft_scale * dof_force[(size_t)e * nd + d];
    }
    const float* os = object_state + (size_t)e * 13;
    int k = 3 * nd;
    for (int i = 0; i < 7; ++i) o[k + i] = os[i];
    for (int i = 0; i < 3; ++i) o[k + 7 + i] = os[7 + i];
    for (int i = 0; i < 3; ++i) o[k + 10 + i] = vel_obs_scale * os[10 + i];
    k += 13;
    const float* gp = goal_pose + (size_t)e * 7;
    for (int i = 0; i < 7; ++i) o[k + i] = gp[i];
    float conj[4], qd[4];
    quat_conjugate(gp + 3, conj);
    quat_mul(os + 3, conj, qd);
    for (int i = 0; i < 4; ++i) o[k + 7 + i] = qd[i];
    k += 11;
    for (int i = 0; i < 13 * nf; ++i) o[k + i] = fingertip_state[(size_t)e * 13 * nf + i];
    k += 13 * nf;
    for (int i = 0; i < 6 * nf; ++i) o[k + i] = ft_scale * sensors[(size_t)e * 6 * nf + i];
    k += 6 * nf;
    for (int i = 0; i < nact; ++i) o[k + i] = actions[(size_t)e * nact + i];
}
__global__ void randomize_rotation_kernel(int n, const float* rand0, const float* rand1, const float* x_unit, const float* y_unit, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    randomize_rotation(rand0[e], rand1[e], x_unit + (size_t)e * 3, y_unit + (size_t)e * 3, out + (size_t)e * 4);
}

// ------------------------------------------------------------------------------------------------ host side
namespace mi {  // defined in kernels_anymal.hip
hipError_t launch_step_anymal(const View& v, const SimParams& P, const AnymalParams& tp, const AnymalTerrainDesc& T,
                              const float* actions, int cfi, unsigned step_counter, hipStream_t s);
hipError_t launch_simulate_anymal(const View& v, const SimParams& P, const AnymalTerrainDesc& T, hipStream_t s);
hipError_t launch_init_anymal(const View& v, const AnymalParams& tp, const AnymalTerrainDesc& T, int max_init_level, hipStream_t s);
hipError_t launch_reset_anymal(const View& v, const AnymalParams& tp, const AnymalTerrainDesc& T, const long long* ids, int n, hipStream_t s);
}
namespace mi {  // defined in kernels_shadow_hand.hip
hipError_t launch_step_anymal_flat(const View& v, const SimParams& P, const AnymalFlatParams& tp, const float* actions, int cfi, hipStream_t s);
hipError_t launch_simulate_anymal_flat(const View& v, const SimParams& P, const AnymalFlatParams& tp, hipStream_t s);
hipError_t launch_init_anymal_flat(const View& v, const AnymalFlatParams& tp, hipStream_t s);
hipError_t launch_reset_anymal_flat(const View& v, const AnymalFlatParams& tp, const long long* ids, int n, hipStream_t s);
hipError_t launch_anymal_obs(int n, const AnymalFlatParams& p, const float* root_states, const float* commands, const float* dof_pos,
                             const float* dof_vel, const float* actions, float* obs, hipStream_t s);
hipError_t launch_anymal_reward(int n, const AnymalFlatParams& p, const float* root_states, const float* commands, const float* torques,
                                const float* contact_forces, int num_bodies, const long long* episode_lengths, float* rew,
                                long long* reset, hipStream_t s);
hipError_t launch_step_quadcopter(const View& v, const QuadView& qv, const SimParams& P, const QuadcopterParams& p, const float* actions,
                                  int cfi, hipStream_t s);
hipError_t launch_simulate_quadcopter(const View& v, const QuadView& qv, const SimParams& P, const QuadcopterParams& p, hipStream_t s);
hipError_t launch_init_quadcopter(const View& v, const QuadView& qv, const QuadcopterParams& p, hipStream_t s);
hipError_t launch_reset_quadcopter(const View& v, const QuadView& qv, const QuadcopterParams& p, const long long* ids, int n, hipStream_t s);
hipError_t launch_quadcopter_reward(int n, const float* root_positions, const float* root_quats, const float* root_linvels,
                                    const float* root_angvels, const long long* progress_buf, float max_episode_length, float* rew,
                                    long long* reset, hipStream_t s);
hipError_t launch_step_ingenuity(const View& v, const IngenuityView& iv, const SimParams& P, const IngenuityParams& p, const float* actions,
                                 int cfi, hipStream_t s);
hipError_t launch_simulate_ingenuity(const View& v, const IngenuityView& iv, const SimParams& P, const IngenuityParams& p, hipStream_t s);
hipError_t launch_init_ingenuity(const View& v, const IngenuityView& iv, const IngenuityParams& p, hipStream_t s);
hipError_t launch_reset_ingenuity(const View& v, const IngenuityView& iv, const IngenuityParams& p, const long long* ids, int n, hipStream_t s);
hipError_t launch_step_ball_balance(const View& v, const BbotView& bv, const SimParams& P, const BallBalanceParams& p, const float* actions,
                                    int cfi, hipStream_t s);
hipError_t launch_simulate_ball_balance(const View& v, const BbotView& bv, const SimParams& P, const BallBalanceParams& p, hipStream_t s);
hipError_t launch_init_ball_balance(const View& v, const BbotView& bv, const BallBalanceParams& p, hipStream_t s);
hipError_t launch_reset_ball_balance(const View& v, const BbotView& bv, const BallBalanceParams& p, const long long* ids, int n, hipStream_t s);
// kernels_articulation.hip (the one translation unit that holds the run-time-compiled robot)
hipError_t launch_simulate_articulation(const View& v, const SimParams& P, const ArticulationParams& p, hipStream_t s);
hipError_t launch_reset_articulation(const View& v, const ArticulationParams& p, const long long* ids, int n, hipStream_t s);
hipError_t launch_body_states_articulation(const View& v, hipStream_t s);
hipError_t launch_kinematics_articulation(const View& v, const SimParams& P, float* out_j, float* out_h, hipStream_t s);
hipError_t launch_step_shadow_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const float* actions, int cfi,
                                   unsigned step_counter, hipStream_t s);
hipError_t launch_simulate_shadow_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, hipStream_t s);
hipError_t launch_init_shadow_hand(const View& v, const HandView& hv, const HandParams& p, hipStream_t s);
hipError_t launch_reset_shadow_hand(const View& v, const HandView& hv, const HandParams& p, const long long* ids, int n, hipStream_t s);
// kernels_allegro_hand.hip
hipError_t launch_step_allegro_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, const float* actions, int cfi,
                                    unsigned step_counter, hipStream_t s);
hipError_t launch_simulate_allegro_hand(const View& v, const HandView& hv, const SimParams& P, const HandParams& p, hipStream_t s);
hipError_t launch_init_allegro_hand(const View& v, const HandView& hv, const HandParams& p, hipStream_t s);
hipError_t launch_reset_allegro_hand(const View& v, const HandView& hv, const HandParams& p, const long long* ids, int n, hipStream_t s);
}
struct MiEngine {
    int task, N;
    SimParams P;
    LocoParams loco;
    CartpoleParams cart;
    AnymalParams anymal;
    AnymalFlatParams anymal_flat;
    QuadcopterParams quad;
    QuadView qv;
    IngenuityParams ing;
    IngenuityView iv;
    BallBalanceParams bbot;
    BbotView bv;
    AnymalTerrainDesc terrain;
    HandParams hand;
    HandView hv;
    ArticulationParams artic;
    int max_init_level;
    int device;            // HIP device the caller's arena lives on: every entry point must be called with it current
    View v;
    float clip_obs;
    int control_freq_inv;
    std::vector<MiTensorDesc> descs;
    unsigned long long steps;
    // the `actor_params` tensors (actor_scale, dof_limit_shift) of Ant / Humanoid: the kernels only read them once the option
    // "actor_tensors" is on -- with the option off (default) the sub-step runs on the model's constants and skips the loads
    float* actor_scale_arena;
    float* limit_shift_arena;
    float* lamp_arena;     // the self-contact impulse tensor (Humanoid), kept while the option self_collision is 0
    float* dof_api_arena;  // the refreshed dof-state tensor (AnymalTerrain), kept while the option dof_state_lag is 0
};

extern "C" int mi_task_info(const char* task, MiTaskInfo* out) {
    int t = find_task(task);
    if (t < 0) return fail(std::string("unknown task: ") + task);
    const TaskMeta& m = kTasks[t];
    out->num_obs = m.nobs; out->num_actions = m.nact; out->num_dofs = m.nd; out->num_bodies = m.nb;
    out->num_sensors = m.nsens; out->num_contact_spheres = m.nsph; out->fixed_base = m.fixed;
    out->task_params_bytes = (int)m.pbytes;
    return 0;
}

extern "C" size_t mi_engine_arena_bytes(const char* task, int num_envs) {
    int t = find_task(task);
    if (t < 0 || num_envs <= 0) { fail("mi_engine_arena_bytes: bad task or num_envs"); return 0; }
    Layout L;
    build_layout(t, num_envs, L, nullptr, nullptr);
    if (is_hand_task(t)) build_hand_layout(t, num_envs, L, nullptr, nullptr);
    if (t == T_QUADCOPTER) build_quad_layout(num_envs, L, nullptr, nullptr);
    if (t == T_INGENUITY) build_ingenuity_layout(num_envs, L, nullptr, nullptr);
    if (t == T_BALLBALANCE) build_bbot_layout(num_envs, L, nullptr, nullptr);
    return L.off;
}

extern "C" int mi_engine_create(const char* task, const MiSimParams* sim, const void* task_params, size_t task_params_bytes,
                                int num_envs, int env_id_offset, uint64_t seed, void* arena, size_t arena_bytes,
                                MiEngine** out) {
    int t = find_task(task);
    if (t < 0) return fail(std::string("unknown task: ") + task);
    if (!sim || !task_params || !arena || !out) return fail("mi_engine_create: null argument");
    if (num_envs <= 0) return fail("mi_engine_create: num_envs must be positive");
    if (task_params_bytes != kTasks[t].pbytes) return fail("mi_engine_create: task_params size mismatch (ABI)");
    if (sim->substeps < 1 || sim->dt <= 0.f) return fail("mi_engine_create: invalid sim params");
    MiEngine* e = new (std::nothrow) MiEngine();
    if (!e) return fail("out of host memory");
    e->task = t; e->N = num_envs; e->steps = 0; e->control_freq_inv = 1; e->clip_obs = INFINITY;
    memcpy(&e->P, sim, sizeof(SimParams));
    memset(&e->terrain, 0, sizeof(e->terrain));
    e->terrain.walls = 1;
    e->max_init_level = 0;
    if (t == T_CARTPOLE) memcpy(&e->cart, task_params, sizeof(CartpoleParams));
    else if (t == T_ANYMAL) memcpy(&e->anymal, task_params, sizeof(AnymalParams));
    else if (t == T_ANYMAL_FLAT) memcpy(&e->anymal_flat, task_params, sizeof(AnymalFlatParams));
    else if (t == T_QUADCOPTER) memcpy(&e->quad, task_params, sizeof(QuadcopterParams));
    else if (t == T_INGENUITY) {
        memcpy(&e->ing, task_params, sizeof(IngenuityParams));
        if (e->ing.target_period < 1) { delete e; return fail("mi_engine_create: Ingenuity target_period must be positive"); }
    }
    else if (t == T_BALLBALANCE) {
        memcpy(&e->bbot, task_params, sizeof(BallBalanceParams));
        const BallBalanceParams& b = e->bbot;
        if (!(b.ball_mass > 0.f) || !(b.ball_inertia > 0.f) || !(b.ball_radius > 0.f) || !(b.pin_stiffness >= 0.f) || !(b.pin_damping >= 0.f) ||
            !(b.pin_stiffness + b.pin_damping > 0.f)) {
            delete e;
            return fail("mi_engine_create: BallBalance needs positive ball mass / inertia / radius and a non-zero attractor");
        }
    }
    else if (is_hand_task(t)) memcpy(&e->hand, task_params, sizeof(HandParams));
    else if (t == T_ARTICULATION) memcpy(&e->artic, task_params, sizeof(ArticulationParams));
    else memcpy(&e->loco, task_params, sizeof(LocoParams));
    Layout L;
    memset(&e->v, 0, sizeof(View));
    e->v.fused_post = 0;
    e->v.fused_sub = 0;
    int nobs = 0;
    if (is_hand_task(t)) {
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
    build_layout(t, num_envs, L, &e->v, (char*)arena, nobs);
    e->lamp_arena = e->v.lamp;       // self-collision is on by default where the reference's actor collides with itself
    e->dof_api_arena = e->v.dof_api; // (AnymalTerrain: the task's lagging dof-state tensor is on by default, as in the reference)
    e->actor_scale_arena = e->v.actor_scale; e->limit_shift_arena = e->v.limit_shift;
    e->v.actor_scale = nullptr; e->v.limit_shift = nullptr;
    memset(&e->hv, 0, sizeof(e->hv));
    e->hv.drive_clamp = 1;
    // the asset's hand-to-hand contact pairs (Shadow Hand, shared.xml:31-51) are on by default; the Allegro hand's URDF lists none
    e->hv.pair_k = (t == T_SHADOWHAND) ? 2.0e4f : 0.f;
    e->hv.tips_in_post = 1;
    e->hv.pre_parts = 4;
    if (is_hand_task(t)) build_hand_layout(t, num_envs, L, &e->hv, (char*)arena);
    memset(&e->qv, 0, sizeof(e->qv));
    if (t == T_QUADCOPTER) build_quad_layout(num_envs, L, &e->qv, (char*)arena);
    memset(&e->iv, 0, sizeof(e->iv));
    if (t == T_INGENUITY) build_ingenuity_layout(num_envs, L, &e->iv, (char*)arena);
    memset(&e->bv, 0, sizeof(e->bv));
    if (t == T_BALLBALANCE) build_bbot_layout(num_envs, L, &e->bv, (char*)arena);
    if (arena_bytes < L.off) { delete e; return fail("mi_engine_create: arena too small"); }
    e->descs = L.d;
    {
        // a host arena is accepted for layout inspection (mi_engine_tensor_desc); every launching entry point refuses it
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, arena) == hipSuccess && attr.type == hipMemoryTypeDevice) e->device = attr.device;
        else { (void)hipGetLastError(); e->device = -1; }
    }
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
// optional knobs (env.clipObservations vec_task.py:115, env.controlFrequencyInv :111)
extern "C" int mi_engine_set_option(MiEngine* e, const char* key, double value) {
    if (!e) return fail("null engine");
    if (!strcmp(key, "clip_obs")) { e->clip_obs = (float)value; e->v.clip_obs = (float)value; return 0; }
    // sim_params.gravity randomisation (reference vec_task.py:720-732 -> gym.set_sim_params)
    if (!strcmp(key, "gravity_x")) { e->P.g[0] = (float)value; return 0; }
    if (!strcmp(key, "gravity_y")) { e->P.g[1] = (float)value; return 0; }
    if (!strcmp(key, "gravity_z")) { e->P.g[2] = (float)value; return 0; }
    if (!strcmp(key, "control_freq_inv")) { if (value < 1) return fail("control_freq_inv < 1"); e->control_freq_inv = (int)value; return 0; }
    // self-collision of the actor (reference: the collision filter passed to gym.create_actor, humanoid.py:194 uses 0 = collide);
    // only tasks whose arena has the self_contact_impulse tensor accept 1
    if (!strcmp(key, "self_collision")) {
        if (value != 0 && !e->lamp_arena) return fail("self_collision: this task's actor has no self-collision tables");
        e->v.lamp = value != 0 ? e->lamp_arena : nullptr;
        return 0;
    }
    // multi-wave sub-step (core/engine_mw.hpp, core/engine_mwc.hpp): envs per workgroup, 0 = one wave per workgroup.  Ignored by tasks
    // whose model has no multi-wave form (Cartpole, Quadcopter).  ShadowHand: any non-zero value selects the finger-per-wave form (32 envs per
    // workgroup, core/hand_engine_mw.hpp).  2: the Humanoid's round-2 form (main wave + self-collision helper).
    if (!strcmp(key, "multi_wave")) {
        const bool hand64 = value == 64 && is_hand_task(e->task);        // full 64-env waves, one workgroup per CU (hand_mw_kernels.hpp)
        // 8: the Ant's one-launch kernel (sub-steps + post step) on 8-env workgroups, two per CU (mw_kernels.hpp); every other launch of such an engine
        // takes the 32-env shape
        const bool ant8 = MI_MW_HAS8 && value == 8 && e->task == T_ANT;       // (an A/B build only: -DMI_MW_HAS8=1)
        if (value != 0 && value != 32 && value != 2 && !(MI_MW_HAS16 && value == 16) && !hand64 && !ant8) return fail("multi_wave: 0, 16 or 32 (envs per workgroup); 2: main + helper wave; 8: Ant only; 64: ShadowHand / AllegroHand only");
        e->v.mw = (int)value;
        return 0;
    }
    // limb-per-wave locomotion (Ant): 1 = post_physics_step runs on one wave of every sub-step workgroup at the end of the step's last
    // sub-step launch (mw_kernels.hpp), 0 (default) = in loco_post_kernel as for the one-wave form
    if (!strcmp(key, "pre_parts")) {           // hands: lanes per env of the pre kernel, 4 (default) or 1
        if (!is_hand_task(e->task)) return fail("pre_parts: a hand-task option");
        if (value != 1 && value != 4) return fail("pre_parts: 1 or 4");
        e->hv.pre_parts = (int)value; return 0;
    }
    if (!strcmp(key, "tips_in_post")) {        // hands: fingertip states by the post kernel's fingertip groups (1, default) or by hand_tips_kernel (0)
        if (!is_hand_task(e->task)) return fail("tips_in_post: a hand-task option");
        e->hv.tips_in_post = value != 0 ? 1 : 0; return 0;
    }
    if (!strcmp(key, "dof_state_lag")) {        // AnymalTerrain: PD law / observations / reward read the dof state of the task's last refresh (1, default, the
        // reference's behaviour: anymal_terrain.py:441-455 + vec_task.py:379-382) or the physics state (0).  Switching it on re-synchronises the tensor.
        if (e->task != T_ANYMAL) return fail("dof_state_lag: an AnymalTerrain option");
        if (value != 0 && e->v.dof_api == nullptr)      // back on: the tensor starts from the physics state (a synchronous copy: this is a set-up call)
            HIP_OK(hipMemcpy(e->dof_api_arena, e->v.dof, (size_t)2 * kTasks[e->task].nd * e->N * sizeof(float), hipMemcpyDeviceToDevice));
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
    if (!strcmp(key, "drive_force_limit")) {   // hands: the position drives deliver at most their force range (shared.xml:250-269, allegro_hand.py:264); default 1
        if (!is_hand_task(e->task)) return fail("drive_force_limit: a hand-task option");
        e->hv.drive_clamp = value != 0 ? 1 : 0; return 0;
    }
    if (!strcmp(key, "fused_post")) { e->v.fused_post = value != 0 ? 1 : 0; return 0; }
    if (!strcmp(key, "fused_sub")) { e->v.fused_sub = value != 0 ? 1 : 0; return 0; }
    // control-step counter (observation ring parity, AnymalTerrain push schedule, noise counters): part of a state checkpoint
    if (!strcmp(key, "steps")) { if (value < 0) return fail("steps < 0"); e->steps = (unsigned long long)value; return 0; }
    if (!strcmp(key, "actor_tensors")) {   // 1: the sub-step reads actor_scale / dof_limit_shift (Ant, Humanoid); ShadowHand always reads its own
        if (is_hand_task(e->task)) return 0;
        if (value != 0 && e->actor_scale_arena == nullptr) return fail("actor_tensors: this task carries no actor_scale / dof_limit_shift tensors");
        e->v.actor_scale = value != 0 ? e->actor_scale_arena : nullptr;
        e->v.limit_shift = value != 0 ? e->limit_shift_arena : nullptr;
        return 0;
    }
    if (!strcmp(key, "terrain_slope_threshold")) {   // terrain.slopeTreshold of the mesh generator (anymal_terrain.py:576); 0 = off
        if (e->task != T_ANYMAL) return fail("terrain_slope_threshold: only AnymalTerrain has a terrain");
        e->terrain.slope_threshold = (float)value;
        return 0;
    }
    if (!strcmp(key, "terrain_walls")) {   // the vertical faces of the slope-corrected triangle mesh (anymal_terrain.py:198-211, :576) collide from the side
        if (e->task != T_ANYMAL) return fail("terrain_walls: only AnymalTerrain has a terrain");
        e->terrain.walls = value != 0 ? 1 : 0;
        return 0;
    }
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
    if (!strcmp(key, "multi_wave")) { *out = e->v.mw; return 0; }
    if (!strcmp(key, "drive_force_limit")) { *out = is_hand_task(e->task) ? e->hv.drive_clamp : 0; return 0; }
    if (!strcmp(key, "dof_state_lag")) { *out = (e->task == T_ANYMAL && e->v.dof_api != nullptr) ? 1 : 0; return 0; }
    if (!strcmp(key, "hand_body_mass")) { *out = (is_hand_task(e->task) && e->hv.body_mass != nullptr) ? 1 : 0; return 0; }
    if (!strcmp(key, "hand_pair_stiffness")) { *out = is_hand_task(e->task) ? e->hv.pair_k : 0; return 0; }
    if (!strcmp(key, "tips_in_post")) { *out = is_hand_task(e->task) ? e->hv.tips_in_post : 0; return 0; }
    if (!strcmp(key, "pre_parts")) { *out = is_hand_task(e->task) ? e->hv.pre_parts : 0; return 0; }
    if (!strcmp(key, "fused_post")) { *out = e->v.fused_post; return 0; }
    if (!strcmp(key, "fused_sub")) { *out = e->v.fused_sub; return 0; }
    if (!strcmp(key, "steps")) { *out = (double)e->steps; return 0; }
    if (!strcmp(key, "terrain_slope_threshold")) { *out = e->terrain.slope_threshold; return 0; }
    if (!strcmp(key, "terrain_walls")) { *out = e->terrain.walls; return 0; }
    if (!strcmp(key, "actor_tensors")) { *out = (is_hand_task(e->task) || e->v.actor_scale != nullptr) ? 1.0 : 0.0; return 0; }
    return fail(std::string("unknown option: ") + key);
}

// kernels launch on the current device; the arena belongs to one device
static int check_device(const MiEngine* e, const char* where) {
    int dev = -1;
    if (e->device < 0) return fail(std::string(where) + ": the engine's arena is not device memory");
    if (hipGetDevice(&dev) != hipSuccess) return fail(std::string(where) + ": hipGetDevice failed");
    if (dev != e->device)
        return fail(std::string(where) + ": the engine's arena is on device " + std::to_string(e->device) + " but device " +
                    std::to_string(dev) + " is current (hipSetDevice / torch.cuda.set_device first)");
    return 0;
}

// the view the initial-state kernel writes through: the `actor_params` tensors are initialised whether the kernels read them or not
static View init_view(const MiEngine* e) {
    View v = e->v;
    v.actor_scale = e->actor_scale_arena; v.limit_shift = e->limit_shift_arena;
    return v;
}
extern "C" int mi_engine_init_state(MiEngine* e, void* stream) {
    if (!e) return fail("null engine");
    if (int rc = check_device(e, "mi_engine_init_state")) return rc;
    hipStream_t s = (hipStream_t)stream;
    const TaskMeta& m = kTasks[e->task];
    float* d_init = nullptr;
    float root_z = 0.f, pot0 = 0.f;
    if (is_hand_task(e->task)) {
        if (e->task == T_SHADOWHAND) HIP_OK(launch_init_shadow_hand(e->v, e->hv, e->hand, s));
        else HIP_OK(launch_init_allegro_hand(e->v, e->hv, e->hand, s));
        e->steps = 0;
        return 0;
    }
    if (e->task == T_QUADCOPTER) {
        const int blocks = (e->N + 255) / 256;
        hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 6 * m.nsens, m.nobs, m.nact,
                           e->quad.init_height, (const float*)nullptr, 0.f);
        HIP_OK(hipGetLastError());
        HIP_OK(launch_init_quadcopter(e->v, e->qv, e->quad, s));
        e->steps = 0;
        return 0;
    }
    if (e->task == T_INGENUITY) {
        const int blocks = (e->N + 255) / 256;
        hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 6 * m.nsens, m.nobs, m.nact,
                           e->ing.init_height, (const float*)nullptr, 0.f);
        HIP_OK(hipGetLastError());
        HIP_OK(launch_init_ingenuity(e->v, e->iv, e->ing, s));
        e->steps = 0;
        return 0;
    }
    if (e->task == T_BALLBALANCE) {
        const int blocks = (e->N + 255) / 256;
        hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 6 * m.nsens, m.nobs, m.nact,
                           e->bbot.tray_height, (const float*)nullptr, 0.f);
        HIP_OK(hipGetLastError());
        HIP_OK(launch_init_ball_balance(e->v, e->bv, e->bbot, s));
        e->steps = 0;
        return 0;
    }
    if (e->task == T_ARTICULATION) {
        const int blocks = (e->N + 255) / 256;
        hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 6 * m.nsens, m.nobs, m.nact,
                           e->artic.init_root[2], (const float*)nullptr, 0.f);
        HIP_OK(hipGetLastError());
        HIP_OK(launch_reset_articulation(e->v, e->artic, nullptr, e->N, s));
        e->steps = 0;
        return 0;
    }
    if (e->task == T_ANYMAL_FLAT) {
        const int blocks = (e->N + 255) / 256;
        hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 0, m.nobs, m.nact,
                           e->anymal_flat.base_init_state[2], (const float*)nullptr, 0.f);
        HIP_OK(hipGetLastError());
        HIP_OK(launch_init_anymal_flat(e->v, e->anymal_flat, s));
        e->steps = 0;
        return 0;
    }
    if (e->task == T_ANYMAL) {
        if (e->terrain.hs == nullptr) return fail("mi_engine_init_state: AnymalTerrain needs mi_engine_set_terrain first");
        const int blocks = (e->N + 255) / 256;
        hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 0, m.nobs, m.nact,
                           e->anymal.base_init_state[2], (const float*)nullptr, 0.f);
        HIP_OK(hipGetLastError());
        HIP_OK(launch_init_anymal(e->v, e->anymal, e->terrain, e->max_init_level, s));
        // the constructor's reset_idx(arange(num_envs)) (anymal_terrain.py:170)
        long long* ids = nullptr;
        HIP_OK(hipMalloc(&ids, sizeof(long long) * e->N));
        std::vector<long long> h(e->N);
        for (int i = 0; i < e->N; ++i) h[i] = i;
        HIP_OK(hipMemcpyAsync(ids, h.data(), sizeof(long long) * e->N, hipMemcpyHostToDevice, s));
        HIP_OK(launch_reset_anymal(e->v, e->anymal, e->terrain, ids, e->N, s));
        HIP_OK(hipStreamSynchronize(s));
        HIP_OK(hipFree(ids));
        e->steps = 0;
        return 0;
    }
    if (e->task != T_CARTPOLE) {
        HIP_OK(hipMalloc(&d_init, sizeof(float) * m.nd));
        HIP_OK(hipMemcpyAsync(d_init, e->loco.initial_dof_pos, sizeof(float) * m.nd, hipMemcpyHostToDevice, s));
        root_z = e->loco.start_height;
        pot0 = -1000.f / e->loco.dt;  // ant.py:113
    } else {
        root_z = 2.0f;  // cartpole.py:93
    }
    const int blocks = (e->N + 255) / 256;
    hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init_view(e), m.nd, 3 * m.nsph, 6 * m.nsens, m.nobs, m.nact,
                       root_z, d_init, pot0);
    HIP_OK(hipGetLastError());
    if (d_init) { HIP_OK(hipStreamSynchronize(s)); HIP_OK(hipFree(d_init)); }
    e->steps = 0;
    return 0;
}

extern "C" int mi_engine_step(MiEngine* e, const float* actions, void* stream) {
    if (!e || !actions) return fail("mi_engine_step: null argument");
    if (int rc = check_device(e, "mi_engine_step")) return rc;
    hipStream_t s = (hipStream_t)stream;
    e->v.ring = (int)(e->steps & 1);
    e->v.step = (unsigned)e->steps;
    switch (e->task) {
        case T_CARTPOLE: HIP_OK(launch_step_cartpole(e->v, e->P, e->cart, actions, e->control_freq_inv, s)); break;
        case T_ANT: HIP_OK(launch_step_ant(e->v, e->P, e->loco, actions, e->control_freq_inv, s)); break;
        case T_HUMANOID: HIP_OK(launch_step_humanoid(e->v, e->P, e->loco, actions, e->control_freq_inv, s)); break;
        case T_SHADOWHAND:
            HIP_OK(launch_step_shadow_hand(e->v, e->hv, e->P, e->hand, actions, e->control_freq_inv, (unsigned)(e->steps + 1), s));
            break;
        case T_ALLEGROHAND:
            HIP_OK(launch_step_allegro_hand(e->v, e->hv, e->P, e->hand, actions, e->control_freq_inv, (unsigned)(e->steps + 1), s));
            break;
        case T_ANYMAL:
            if (e->terrain.hs == nullptr) return fail("mi_engine_step: AnymalTerrain needs mi_engine_set_terrain first");
            // common_step_counter is incremented before the push test (anymal_terrain.py:460-462)
            HIP_OK(launch_step_anymal(e->v, e->P, e->anymal, e->terrain, actions, e->control_freq_inv, (unsigned)(e->steps + 1), s));
            break;
        case T_ANYMAL_FLAT: HIP_OK(launch_step_anymal_flat(e->v, e->P, e->anymal_flat, actions, e->control_freq_inv, s)); break;
        case T_QUADCOPTER: HIP_OK(launch_step_quadcopter(e->v, e->qv, e->P, e->quad, actions, e->control_freq_inv, s)); break;
        case T_INGENUITY: HIP_OK(launch_step_ingenuity(e->v, e->iv, e->P, e->ing, actions, e->control_freq_inv, s)); break;
        case T_BALLBALANCE: HIP_OK(launch_step_ball_balance(e->v, e->bv, e->P, e->bbot, actions, e->control_freq_inv, s)); break;
        case T_ARTICULATION: return fail("mi_engine_step: the Articulation task has no task kernels -- drive it with mi_engine_simulate (gym.simulate) and keep "
                                         "the observation / reward code on the caller's side");
    }
    e->steps++;
    return 0;
}
extern "C" int mi_engine_last_ring(const MiEngine* e) { return e ? (int)((e->steps + 1) & 1) : -1; }

extern "C" int mi_engine_simulate(MiEngine* e, void* stream) {
    if (!e) return fail("null engine");
    if (int rc = check_device(e, "mi_engine_simulate")) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (e->task) {
        case T_CARTPOLE: HIP_OK(launch_simulate_cartpole(e->v, e->P, s)); break;
        case T_ANT: HIP_OK(launch_simulate_ant(e->v, e->P, s)); break;
        case T_HUMANOID: HIP_OK(launch_simulate_humanoid(e->v, e->P, s)); break;
        case T_SHADOWHAND: HIP_OK(launch_simulate_shadow_hand(e->v, e->hv, e->P, e->hand, s)); break;
        case T_ALLEGROHAND: HIP_OK(launch_simulate_allegro_hand(e->v, e->hv, e->P, e->hand, s)); break;
        case T_ANYMAL:
            if (e->terrain.hs == nullptr) return fail("mi_engine_simulate: AnymalTerrain needs mi_engine_set_terrain first");
            HIP_OK(launch_simulate_anymal(e->v, e->P, e->terrain, s));
            break;
        case T_ANYMAL_FLAT: HIP_OK(launch_simulate_anymal_flat(e->v, e->P, e->anymal_flat, s)); break;
        case T_QUADCOPTER: HIP_OK(launch_simulate_quadcopter(e->v, e->qv, e->P, e->quad, s)); break;
        case T_INGENUITY: HIP_OK(launch_simulate_ingenuity(e->v, e->iv, e->P, e->ing, s)); break;
        case T_BALLBALANCE: HIP_OK(launch_simulate_ball_balance(e->v, e->bv, e->P, e->bbot, s)); break;
        case T_ARTICULATION: HIP_OK(launch_simulate_articulation(e->v, e->P, e->artic, s)); break;
    }
    return 0;
}

namespace mi { hipError_t launch_body_states(int task, const View& v, hipStream_t s); }
extern "C" int mi_engine_refresh_rigid_body_states(MiEngine* e, void* stream) {
    if (!e) return fail("null engine");
    if (int rc = check_device(e, "mi_engine_refresh_rigid_body_states")) return rc;
    static_assert(T_CARTPOLE == 0 && T_ANT == 1 && T_HUMANOID == 2 && T_ANYMAL == 3 && T_SHADOWHAND == 4 && T_ANYMAL_FLAT == 5 && T_QUADCOPTER == 6 &&
                  T_INGENUITY == 7 && T_BALLBALANCE == 8 && T_ALLEGROHAND == 9, "kernels_body_states.hip switches on these ids");
    if (e->task == T_ARTICULATION) { HIP_OK(launch_body_states_articulation(e->v, (hipStream_t)stream)); return 0; }
    HIP_OK(launch_body_states(e->task, e->v, (hipStream_t)stream));
    return 0;
}

namespace mi { hipError_t launch_kinematics_views(int task, const View& v, const SimParams& P, float* out_j, float* out_h, hipStream_t s); }
extern "C" int mi_engine_compute_jacobians(MiEngine* e, float* out, void* stream) {
    if (!e || !out) return fail("mi_engine_compute_jacobians: null argument");
    if (int rc = check_device(e, "mi_engine_compute_jacobians")) return rc;
    if (e->task == T_ARTICULATION) { HIP_OK(launch_kinematics_articulation(e->v, e->P, out, nullptr, (hipStream_t)stream)); return 0; }
    HIP_OK(launch_kinematics_views(e->task, e->v, e->P, out, nullptr, (hipStream_t)stream));
    return 0;
}
extern "C" int mi_engine_compute_mass_matrices(MiEngine* e, float* out, void* stream) {
    if (!e || !out) return fail("mi_engine_compute_mass_matrices: null argument");
    if (int rc = check_device(e, "mi_engine_compute_mass_matrices")) return rc;
    if (e->task == T_ARTICULATION) { HIP_OK(launch_kinematics_articulation(e->v, e->P, nullptr, out, (hipStream_t)stream)); return 0; }
    HIP_OK(launch_kinematics_views(e->task, e->v, e->P, nullptr, out, (hipStream_t)stream));
    return 0;
}

extern "C" int mi_engine_reset_idx(MiEngine* e, const int64_t* env_ids, int n, void* stream) {
    if (!e) return fail("null engine");
    if (n <= 0) return 0;
    if (!env_ids) return fail("mi_engine_reset_idx: null env_ids");
    if (int rc = check_device(e, "mi_engine_reset_idx")) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (e->task) {
        case T_CARTPOLE: HIP_OK(launch_reset_cartpole(e->v, (const long long*)env_ids, n, s)); break;
        case T_ANT: HIP_OK(launch_reset_ant(e->v, e->loco, (const long long*)env_ids, n, s)); break;
        case T_HUMANOID: HIP_OK(launch_reset_humanoid(e->v, e->loco, (const long long*)env_ids, n, s)); break;
        case T_ANYMAL: HIP_OK(launch_reset_anymal(e->v, e->anymal, e->terrain, (const long long*)env_ids, n, s)); break;
        case T_SHADOWHAND: HIP_OK(launch_reset_shadow_hand(e->v, e->hv, e->hand, (const long long*)env_ids, n, s)); break;
        case T_ALLEGROHAND: HIP_OK(launch_reset_allegro_hand(e->v, e->hv, e->hand, (const long long*)env_ids, n, s)); break;
        case T_ANYMAL_FLAT: HIP_OK(launch_reset_anymal_flat(e->v, e->anymal_flat, (const long long*)env_ids, n, s)); break;
        case T_QUADCOPTER: HIP_OK(launch_reset_quadcopter(e->v, e->qv, e->quad, (const long long*)env_ids, n, s)); break;
        case T_INGENUITY: HIP_OK(launch_reset_ingenuity(e->v, e->iv, e->ing, (const long long*)env_ids, n, s)); break;
        case T_BALLBALANCE: HIP_OK(launch_reset_ball_balance(e->v, e->bv, e->bbot, (const long long*)env_ids, n, s)); break;
        case T_ARTICULATION: HIP_OK(launch_reset_articulation(e->v, e->artic, (const long long*)env_ids, n, s)); break;
    }
    return 0;
}

extern "C" int mi_engine_set_terrain(MiEngine* e, const int16_t* height_samples, int rows, int cols, float horizontal_scale,
                                     float vertical_scale, float border_size, const float* env_origins, int num_levels,
                                     int num_terrains, float env_length, int max_init_level) {
    if (!e) return fail("null engine");
    if (e->task != T_ANYMAL) return fail("mi_engine_set_terrain: only AnymalTerrain uses a terrain");
    if (!height_samples || !env_origins || rows < 2 || cols < 2 || num_levels < 1 || num_terrains < 1 || horizontal_scale <= 0.f)
        return fail("mi_engine_set_terrain: bad argument");
    e->terrain.hs = (const short*)height_samples; e->terrain.rows = rows; e->terrain.cols = cols;
    e->terrain.hscale = horizontal_scale; e->terrain.vscale = vertical_scale; e->terrain.border = border_size;
    e->terrain.origins = env_origins; e->terrain.levels = num_levels; e->terrain.types = num_terrains;
    e->terrain.env_length = env_length;
    e->max_init_level = max_init_level < 0 ? 0 : (max_init_level >= num_levels ? num_levels - 1 : max_init_level);
    return 0;
}

extern "C" int mi_compute_locomotion_observations(const char* task, int n, const MiLocoParams* p, const float* root_states,
                                                  const float* targets, float* potentials, float* prev_potentials,
                                                  const float* inv_start_rot, const float* dof_pos, const float* dof_vel,
                                                  const float* dof_force, const float* lower, const float* upper,
                                                  const float* sensors, const float* actions, const float* basis0,
                                                  const float* basis1, float* obs_buf, float* up_vec, float* heading_vec,
                                                  void* stream) {
    int t = find_task(task);
    if (t != T_ANT && t != T_HUMANOID) return fail("mi_compute_locomotion_observations: task must be Ant or Humanoid");
    if (n <= 0) return 0;
    LocoParams lp;
    memcpy(&lp, p, sizeof(lp));
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (n + 63) / 64;
    if (t == T_ANT)
        hipLaunchKernelGGL((loco_obs_kernel<ModelAnt::ND, 6 * ModelAnt::NSENS, false>), dim3(blocks), dim3(64), 0, s, n, lp, root_states, targets,
                           potentials, prev_potentials, inv_start_rot, dof_pos, dof_vel, dof_force, lower, upper, sensors,
                           actions, basis0, basis1, obs_buf, up_vec, heading_vec);
    else
        hipLaunchKernelGGL((loco_obs_kernel<ModelHumanoid::ND, 6 * ModelHumanoid::NSENS, true>), dim3(blocks), dim3(64), 0, s, n, lp, root_states,
                           targets, potentials, prev_potentials, inv_start_rot, dof_pos, dof_vel, dof_force, lower, upper,
                           sensors, actions, basis0, basis1, obs_buf, up_vec, heading_vec);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int mi_compute_locomotion_reward(const char* task, int n, const MiLocoParams* p, const float* obs_buf,
                                            const int64_t* reset_in, const int64_t* progress, const float* actions,
                                            const float* potentials, const float* prev_potentials, float* rew,
                                            int64_t* reset_out, void* stream) {
    int t = find_task(task);
    if (t != T_ANT && t != T_HUMANOID) return fail("mi_compute_locomotion_reward: task must be Ant or Humanoid");
    if (n <= 0) return 0;
    LocoParams lp;
    memcpy(&lp, p, sizeof(lp));
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (n + 63) / 64;
    if (t == T_ANT)
        hipLaunchKernelGGL((loco_reward_kernel<ModelAnt::ND, 6 * ModelAnt::NSENS, false>), dim3(blocks), dim3(64), 0, s, n, lp, obs_buf,
                           (const long long*)reset_in, (const long long*)progress, actions, potentials, prev_potentials, rew, (long long*)reset_out);
    else
        hipLaunchKernelGGL((loco_reward_kernel<ModelHumanoid::ND, 6 * ModelHumanoid::NSENS, true>), dim3(blocks), dim3(64), 0, s, n, lp, obs_buf,
                           (const long long*)reset_in, (const long long*)progress, actions, potentials, prev_potentials, rew, (long long*)reset_out);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int mi_compute_cartpole_reward(int n, const MiCartpoleParams* p, const float* pole_angle, const float* pole_vel,
                                          const float* cart_vel, const float* cart_pos, const int64_t* reset_in,
                                          const int64_t* progress, float* rew, int64_t* reset_out, void* stream) {
    if (n <= 0) return 0;
    CartpoleParams cp;
    memcpy(&cp, p, sizeof(cp));
    hipLaunchKernelGGL(cartpole_reward_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, cp, pole_angle, pole_vel,
                       cart_vel, cart_pos, (const long long*)reset_in, (const long long*)progress, rew, (long long*)reset_out);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int mi_compute_quadcopter_reward(int n, const float* root_positions, const float* root_quats, const float* root_linvels,
                                            const float* root_angvels, const int64_t* reset_buf_in, const int64_t* progress_buf,
                                            float max_episode_length, float* rew_buf, int64_t* reset_buf_out, void* stream) {
    if (n <= 0) return 0;
    (void)reset_buf_in;   // the reference only uses its shape (torch.ones_like / zeros_like, quadcopter.py:376-377)
    if (!root_positions || !root_quats || !root_linvels || !root_angvels || !progress_buf || !rew_buf || !reset_buf_out)
        return fail("mi_compute_quadcopter_reward: null argument");
    HIP_OK(launch_quadcopter_reward(n, root_positions, root_quats, root_linvels, root_angvels, (const long long*)progress_buf,
                                    max_episode_length, rew_buf, (long long*)reset_buf_out, (hipStream_t)stream));
    return 0;
}
extern "C" int mi_compute_anymal_observations(int n, const MiAnymalFlatParams* p, const float* root_states, const float* commands,
                                              const float* dof_pos, const float* dof_vel, const float* actions, float* obs_buf,
                                              void* stream) {
    if (n <= 0) return 0;
    if (!p || !root_states || !commands || !dof_pos || !dof_vel || !actions || !obs_buf) return fail("mi_compute_anymal_observations: null argument");
    AnymalFlatParams ap;
    memcpy(&ap, p, sizeof(ap));
    HIP_OK(launch_anymal_obs(n, ap, root_states, commands, dof_pos, dof_vel, actions, obs_buf, (hipStream_t)stream));
    return 0;
}
extern "C" int mi_compute_anymal_reward(int n, const MiAnymalFlatParams* p, const float* root_states, const float* commands,
                                        const float* torques, const float* contact_forces, int num_bodies,
                                        const int64_t* episode_lengths, float* rew_buf, int64_t* reset_buf, void* stream) {
    if (n <= 0) return 0;
    if (!p || !root_states || !commands || !torques || !contact_forces || !episode_lengths || !rew_buf || !reset_buf)
        return fail("mi_compute_anymal_reward: null argument");
    if (num_bodies < ModelAnymal::NB) return fail("mi_compute_anymal_reward: contact_forces must cover the 13 ANYmal bodies");
    AnymalFlatParams ap;
    memcpy(&ap, p, sizeof(ap));
    HIP_OK(launch_anymal_reward(n, ap, root_states, commands, torques, contact_forces, num_bodies, (const long long*)episode_lengths,
                                rew_buf, (long long*)reset_buf, (hipStream_t)stream));
    return 0;
}

extern "C" int mi_compute_hand_reward(int n, const MiHandRewardParams* p, float* rew_buf, int64_t* reset_buf, int64_t* reset_goal_buf,
                                      int64_t* progress_buf, float* successes, float* consecutive_successes, const float* object_pos,
                                      const float* object_rot, const float* target_pos, const float* target_rot, const float* actions,
                                      int num_actions, float* workspace2, void* stream) {
    if (n <= 0) return 0;
    if (!p || !workspace2) return fail("mi_compute_hand_reward: null argument");
    HandRewardParams hp;
    memcpy(&hp, p, sizeof(hp));
    hipStream_t s = (hipStream_t)stream;
    HIP_OK(hipMemsetAsync(workspace2, 0, 2 * sizeof(float), s));
    hipLaunchKernelGGL(hand_reward_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, hp, object_pos, object_rot, target_pos, target_rot, actions,
                       num_actions, rew_buf, (long long*)reset_buf, (long long*)reset_goal_buf, (long long*)progress_buf, successes, workspace2);
    hipLaunchKernelGGL(hand_reward_finalize_kernel, dim3(1), dim3(64), 0, s, hp, workspace2, consecutive_successes);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int mi_compute_hand_full_state(int n, int num_dofs, int num_fingertips, int num_actions, float vel_obs_scale,
                                          float force_torque_obs_scale, const float* dof_pos, const float* dof_vel, const float* dof_force,
                                          const float* dof_lower, const float* dof_upper, const float* object_state, const float* goal_pose,
                                          const float* fingertip_state, const float* fingertip_force_torque, const float* actions,
                                          float* obs_buf, int obs_stride, void* stream) {
    if (n <= 0) return 0;
    if (obs_stride < 3 * num_dofs + 24 + 19 * num_fingertips + num_actions) return fail("mi_compute_hand_full_state: obs_stride too small");
    hipLaunchKernelGGL(hand_full_state_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, num_dofs, num_fingertips, num_actions,
                       vel_obs_scale, force_torque_obs_scale, dof_pos, dof_vel, dof_force, dof_lower, dof_upper, object_state, goal_pose,
                       fingertip_state, fingertip_force_torque, actions, obs_buf, obs_stride);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int mi_randomize_rotation(int n, const float* rand0, const float* rand1, const float* x_unit, const float* y_unit, float* out_quat,
                                     void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(randomize_rotation_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, rand0, rand1, x_unit, y_unit, out_quat);
    HIP_OK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ device probe (bench.py "box")
namespace {
__global__ __launch_bounds__(256) void probe_fma_kernel(float* out, int iters, float a, float b) {
    float x = (float)threadIdx.x * 1e-3f;
#pragma unroll 16
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, a, b);       // one dependent v_fma_f32 per iteration
    out[threadIdx.x] = x;
}
__global__ void probe_chase_init_kernel(unsigned* buf, unsigned n, unsigned stride) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[(size_t)i * 16] = (unsigned)(((unsigned long long)i * stride + 1u) % n);   // one hop per 64-byte line, a fixed odd stride: a full cycle
}
__global__ __launch_bounds__(64) void probe_chase_kernel(const unsigned* buf, unsigned* out, int hops, unsigned start) {
    if (threadIdx.x != 0) return;
    unsigned k = start;
    for (int i = 0; i < hops; ++i) k = __builtin_nontemporal_load(buf + (size_t)k * 16);
    out[0] = k;
}
// a chain of dependent quarter-rate operations (v_rcp_f32, v_sin_f32: what the tree pass and the fingertip chains are made of)
__global__ __launch_bounds__(256) void probe_trans_kernel(float* out, int iters) {
    float x = 0.3f + (float)threadIdx.x * 1e-3f;
#pragma unroll 8
    for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_sinf(__builtin_amdgcn_rcpf(x + 1.5f)) + 0.25f;
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
// four waves of a workgroup meeting at `iters` barriers, each exchanging one LDS word per barrier (the limb-per-wave kernels' pattern)
__global__ __launch_bounds__(256) void probe_barrier_kernel(float* out, int iters) {
    __shared__ float x[4][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    float acc = (float)l;
    for (int i = 0; i < iters; ++i) {
        x[w][l] = acc;
        __syncthreads();
        acc += x[(w + 1) & 3][l] * 1e-6f;
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// a chain of dependent LDS reads on one lane
__global__ __launch_bounds__(64) void probe_lds_kernel(unsigned* out, int hops) {
    __shared__ unsigned idx[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) idx[i] = (unsigned)((i * 389 + 1) & 1023);
    __syncthreads();
    unsigned k = threadIdx.x;
    for (int i = 0; i < hops; ++i) k = idx[k];
    out[threadIdx.x] = k;
}
}  // namespace
extern "C" int mi_device_probe(void* scratch, long long chase_bytes, int fma_iters, int hops, float* out2 /* [6] */, void* stream) {
    // scratch: the barrier / transcendental probes write 65536 floats into it, the pointer chase walks all of it
    if (!scratch || !out2 || chase_bytes < 65536 * (long long)sizeof(float) || fma_iters < 1 || hops < 1) return fail("mi_device_probe: bad argument (scratch of at least 256 KB)");
    hipStream_t s = (hipStream_t)stream;
    struct Events {       // destroyed on every return path (HIP_OK leaves early)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
    } ev;
    HIP_OK(hipEventCreate(&ev.a)); HIP_OK(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    float* fout = (float*)scratch;
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {       // (first launch: code upload, clocks ramping up)
        HIP_OK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(probe_fma_kernel, dim3(1), dim3(64), 0, s, fout, fma_iters, 0.999f, 1e-3f);
        HIP_OK(hipEventRecord(e1, s));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    out2[0] = best * 1e3f / ((float)fma_iters / 1000.f);
    // the same chain on one wave per SIMD of the whole chip (256 workgroups x 4 waves): what the clocks do under load
    best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_OK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(probe_fma_kernel, dim3(256), dim3(256), 0, s, fout, fma_iters, 0.999f, 1e-3f);
        HIP_OK(hipEventRecord(e1, s));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    out2[2] = best * 1e3f / ((float)fma_iters / 1000.f);
    {   // barrier + LDS exchange round trips of a four-wave workgroup, one workgroup per CU; dependent LDS reads
        const int bi = 20000;
        best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            HIP_OK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(probe_barrier_kernel, dim3(256), dim3(256), 0, s, fout, bi);
            HIP_OK(hipEventRecord(e1, s));
            HIP_OK(hipEventSynchronize(e1));
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        out2[3] = best * 1e6f / (2.f * bi);       // ns per barrier
        best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            HIP_OK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(probe_lds_kernel, dim3(1), dim3(64), 0, s, (unsigned*)scratch, 200000);
            HIP_OK(hipEventRecord(e1, s));
            HIP_OK(hipEventSynchronize(e1));
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        out2[4] = best * 1e6f / 200000.f;         // ns per dependent LDS read
        best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            HIP_OK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(probe_trans_kernel, dim3(256), dim3(256), 0, s, fout, 200000);
            HIP_OK(hipEventRecord(e1, s));
            HIP_OK(hipEventSynchronize(e1));
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        out2[5] = best * 1e6f / 200000.f;         // ns per dependent (v_add, v_rcp, v_sin, v_add) group, a wave on every SIMD
    }
    const unsigned n = (unsigned)(chase_bytes / 64);
    unsigned stride = (unsigned)(n * 0.6180339887) | 1u;                 // odd: coprime with a power-of-two line count
    while (n % 2 != 0 && stride > 1 && (n % stride) == 0) stride += 2;
    hipLaunchKernelGGL(probe_chase_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (unsigned*)scratch, n, stride);
    best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_OK(hipEventRecord(e0, s));
        // (every repetition walks another stretch of the cycle: lines no earlier repetition left in a cache)
        hipLaunchKernelGGL(probe_chase_kernel, dim3(1), dim3(64), 0, s, (const unsigned*)scratch, (unsigned*)scratch + 1, hops,
                           (unsigned)(((unsigned long long)(rep + 1) * 1000003ull) % n));
        HIP_OK(hipEventRecord(e1, s));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    out2[1] = best * 1e6f / (float)hops;
    return 0;
}

// ------------------------------------------------------------------------------------------------ LDS poison (tests)
// Fills the LDS of every CU with a bit pattern (one 160 KB workgroup per CU, a few rounds).  LDS keeps what the last kernel on the CU left
// in it; a kernel that reads a slot before writing it therefore depends on what ran before -- tests/test_gpu_fullsize.py poisons the LDS
// with NaNs before the benchmark-size comparisons so that such a read shows up as a NaN instead of as a rare mismatch.
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_words[i] = pattern;
    __syncthreads();
    if (threadIdx.x == 0 && sink != nullptr && lds_words[17] != pattern) sink[0] = 1u;      // (keeps the stores alive)
}
}  // namespace
extern "C" int mi_debug_poison_lds(unsigned pattern, void* stream) {
    static unsigned long long conf = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)poison_lds_kernel, 160 * 1024, &conf); e != hipSuccess) return fail(hipGetErrorString(e));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024, (hipStream_t)stream, pattern, (unsigned*)nullptr);
    HIP_OK(hipGetLastError());
    return 0;
}

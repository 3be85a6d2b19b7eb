// arena.hpp -- the view of the caller-owned SoA arena that every kernel (and the CPU backend) receives.  No HIP dependency.
#pragma once
#include <cstdint>
#include "core/rng.hpp"

namespace mi {

// ---------------------------------------------------------------------------------------------- arena view
struct View {
    int N;
    int env_offset;
    uint32_t seed;
    int ring;  // which obs_out slot this step writes
    int mw;    // multi-wave sub-step: envs per workgroup (16 or 32), 0 = one wave per workgroup (option "multi_wave")
    int nas;   // columns of actor_scale (NB + 3 ND), 0: the task has none
    int fused_post;   // limb-per-wave locomotion (Ant): post_physics_step inside the last sub-step launch instead of a kernel of its own (option "fused_post", default 0; the Python layer switches it on for small batches)
    int fused_sub;    // limb-per-wave sub-steps (Ant, ANYmal on the terrain): all sub-steps of a control step in ONE launch, the state staying in registers / LDS between them (option "fused_sub", default 0; mw_kernels.hpp)
    float clip_obs;
    unsigned step;      // control-step counter of this step() (white-noise stream of the in-kernel observation / action noise)
    NoiseParams obs_noise, act_noise;   // domain randomisation noise on observations / actions (dist 0: off), mi_engine_set_noise
    const float* alloc_fence;   // ALWAYS null: the pointer the never-taken register-allocation fence of the sub-step kernels tests (core/engine.hpp Sim::alloc_fence)
    float* limit_shift; // [2*ND][N] per-env shifts of the lower, then the upper joint limits (`actor_params` dof_properties.lower / upper), null: task has none
    float* actor_scale; // [NB + 3 ND][N] per-env scales: link masses per body, then joint damping, stiffness, armature per dof (`actor_params`), null: task has none
    float* root;        // [13][N]
    float* dof;         // [2][ND][N]  (pos block, vel block)
    float* dof_api = nullptr;   // [2][ND][N] AnymalTerrain: the joint state as of the task's last gym.refresh_dof_state_tensor -- the end of the decimation
                        // loop, anymal_terrain.py:451; the base class's simulate() calls after it refresh nothing (vec_task.py:379-382) -- which is what
                        // the task's PD law (first decimation iteration), observations and reward read.  nullptr (every other task; option
                        // dof_state_lag 0): they read `dof`
    float* tau;         // [ND][N]  dof_actuation_force
    float* lamc;        // [3*NSPH][N]
    float* laml;        // [ND][N]
    float* sensor;      // [6*NSENS][N]
    float* dof_force;   // [ND][N]
    float* potentials;  // [N]
    float* prev_potentials;
    float* up_vec;      // [3][N]
    float* heading_vec; // [3][N]
    float* actions;     // [NACT][N]
    float* init_root;   // [13][N]
    float* obs;         // [N][NOBS] row-major
    float* obs_out;     // [2][N][NOBS]
    float* rew;         // [N]
    long long* reset;   // [N]
    long long* progress;
    long long* randomize;
    unsigned char* timeout;
    int* episode;
    // ---- actors that collide with themselves (Humanoid, reference humanoid.py:194); null otherwise / when switched off
    float* lamp;        // [3*NPG][N] warm-start impulses of the self-contact groups
    float* pairf;       // [3*NPG][N] world force on side a of each group's contact, last sub-step
    int* dropped;       // [2][N] contacts refused since init because the env's slots were taken: ground (KMAX), self contacts (KPAIR); models with the compact store
    float* ep_ret;      // [N] running return of the current episode
    float* body_state;  // [13*NB][N] world state of every rigid body, filled by mi_engine_refresh_rigid_body_states only (gym rigid_body_state tensor)
    float* stats;       // [8] job statistics: sum finished returns, sum finished lengths, #finished, sum rewards, #env-steps
    // ---- AnymalTerrain only (null otherwise)
    float* netf;          // [3*NB][N] net contact force per body, world frame (gym net_contact_force tensor)
    float* commands;      // [4][N] x vel, y vel, yaw vel, heading (anymal_terrain.py:140)
    float* last_actions;  // [12][N]
    float* last_dof_vel;  // [12][N]
    float* feet_air_time; // [4][N]
    float* episode_sums;  // [13][N]
    float* env_origins;   // [3][N]
    float* friction;      // [N] per-env shape friction (100 buckets, :236-239,279-281)
    int* terrain_levels;  // [N]
    int* terrain_types;   // [N]
    float* ep_stats;      // [16] this step: sum of the 13 episode sums over resetting envs, #resets, sum terrain levels
    float* ep_means;      // [16] extras["episode"]: rew_* means / max_episode_length_s, terrain_level mean (:421-425)
    float* targets;       // [ND][N] position targets of the Articulation task's drives (gym.set_dof_position_target_tensor); null otherwise
    float* scene = nullptr;     // [13 * kSceneMaxFree][N] root states of the free boxes of the Articulation task's scene (core/scene_engine.hpp); null otherwise
    int* scene_nc = nullptr;    // [2][N] scene contacts taken in the last sub-step, refused for want of a slot since reset
    float* scene_warm = nullptr;   // [4 * 48][N] per contact slot of the last sub-step: feature id (int bits; 0 = empty), impulses (normal, two tangents)
    float* ep_cum;        // [16] the same sums accumulated since init, never re-zeroed: 13 episode sums of the envs that reset, [13] their count,
                          //      [14] sum of the terrain levels of all envs over the steps, [15] the steps -- what a multi-GPU job all-reduces every K
                          //      steps to form job-wide extras["episode"] (parallel.py TaskExtrasReducer)
};

}  // namespace mi

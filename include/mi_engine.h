/*
 * mi_engine.h -- C ABI of the MI355X-native vectorised RL-environment engine (libmi_engine.so).
 *
 * The reference (isaac-sim/IsaacGymEnvs) has no FFI of its own: its hot path crosses from Python into the
 * closed `isaacgym` pybind module.  This header is the C boundary a replacement for that module binds to.
 * Each entry point cites the reference interface it replaces (paths relative to /root/reference/).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated otherwise; no torch types cross this boundary;
 *   - the caller owns all memory: the engine state lives in one caller-allocated "arena"
 *     (mi_engine_arena_bytes), tensors are described as (byte offset, shape, element strides) views of it,
 *     exactly like `gym.acquire_*_tensor` + `gymtorch.wrap_tensor` expose simulator memory
 *     (isaacgymenvs/tasks/ant.py:77-95);
 *   - every launch is enqueued on the caller's HIP stream (`stream` = hipStream_t, NULL = default stream), no
 *     host synchronisation inside;
 *   - return value 0 = ok, negative = error, message via mi_last_error() (thread local);
 *   - quaternions xyzw; fp32 state; reset/progress buffers int64 (isaacgymenvs/tasks/base/vec_task.py:316-323).
 */
#ifndef MI_ENGINE_H
#define MI_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_MAX_DOF 32
#define MI_ABI_VERSION 4

typedef struct MiEngine MiEngine;

/* gymapi.SimParams subset parsed by VecTask.__parse_sim_params (vec_task.py:514-562) plus solver constants */
typedef struct {
    float dt;
    int32_t substeps;          /* sim.substeps */
    int32_t iters;             /* sim.physx.num_position_iterations */
    float gravity[3];
    float contact_offset;      /* sim.physx.contact_offset */
    float rest_offset;         /* sim.physx.rest_offset */
    float max_depen_vel;       /* sim.physx.max_depenetration_velocity */
    float erp;                 /* penetration/limit error reduction per sub-step (engine constant, default 0.5) */
    float plane_mu;            /* env.plane.staticFriction (ant.py:128-133) */
    float ground_z;
    float cfm;                 /* constraint regularisation (default 1e-6) */
    float warm;                /* warm-start factor (default 1) */
} MiSimParams;

/* task parameters of Ant / Humanoid: the scalars the reference passes into compute_*_observations/_reward
 * (ant.py:214-242, humanoid.py:218-251) and uses in reset_idx / pre_physics_step (ant.py:252-285) */
typedef struct {
    float dt, dof_vel_scale, contact_force_scale, angular_velocity_scale, power_scale;
    float heading_weight, up_weight, actions_cost, energy_cost, joints_at_limit_cost;
    float death_cost, termination_height, max_episode_length, clip_actions, max_motor_effort, start_height;
    float gear[MI_MAX_DOF];
    float dof_lower[MI_MAX_DOF], dof_upper[MI_MAX_DOF], initial_dof_pos[MI_MAX_DOF];
    float targets[3];
    float inv_start_rot[4];
    float basis_vec0[3], basis_vec1[3];
    float reset_pos_noise, reset_vel_noise;
} MiLocoParams;

/* task parameters of Cartpole (cartpole.py:40-47) */
typedef struct {
    float reset_dist, max_push_effort, max_episode_length, clip_actions;
} MiCartpoleParams;

/* task parameters of AnymalTerrain: what the reference's __init__ reads from cfg["env"] (anymal_terrain.py:47-105).
 * Reward scales are already multiplied by the control dt (:104-105). */
typedef struct {
    float lin_vel_scale, ang_vel_scale, dof_pos_scale, dof_vel_scale, height_meas_scale, action_scale;
    float rew_termination, rew_lin_vel_xy, rew_lin_vel_z, rew_ang_vel_z, rew_ang_vel_xy, rew_orient, rew_torque, rew_joint_acc,
          rew_base_height, rew_air_time, rew_collision, rew_stumble, rew_action_rate, rew_hip;
    float command_x[2], command_y[2], command_yaw[2];
    float base_init_state[13];
    float default_dof_pos[12];
    float kp, kd, torque_limit;
    float dt;                       /* control dt = decimation * sim.dt */
    float max_episode_length_s;
    int32_t max_episode_length, push_interval, allow_knee_contacts, decimation, add_noise;
    float noise_lin_vel, noise_ang_vel, noise_gravity, noise_dof_pos, noise_dof_vel, noise_height;
    int32_t curriculum;
    float clip_actions;
    float friction_range[2];
    float terrain_mu;
} MiAnymalParams;

/* task parameters of Anymal (flat ground): what the reference's __init__ reads from cfg["env"] (anymal.py:42-97).
 * Reward scales are the YAML values already multiplied by dt (anymal.py:96-97). */
typedef struct {
    float lin_vel_scale, ang_vel_scale, dof_pos_scale, dof_vel_scale, action_scale; /* learn.*Scale, control.actionScale */
    float rew_lin_vel_xy, rew_ang_vel_z, rew_torque;
    float command_x[2], command_y[2], command_yaw[2];   /* randomCommandVelocityRanges */
    float base_init_state[13];                          /* baseInitState pos, rot, vLinear, vAngular */
    float default_dof_pos[12];                          /* defaultJointAngles in dof order */
    float kp, kd, torque_limit;                         /* control.stiffness / damping (DOF_MODE_POS drive); URDF effort */
    int32_t max_episode_length;                         /* int(episodeLength_s / dt + 0.5) (anymal.py:92) */
    float clip_actions;
} MiAnymalFlatParams;

/* task parameters of Quadcopter (quadcopter.py:45-101, 203-240): constants the reference hard-codes in the task file */
typedef struct {
    float max_episode_length;            /* env.maxEpisodeLength */
    float dt;                            /* sim.dt */
    float dof_lower[8], dof_upper[8];    /* rotor joint limits of the generated asset (+-30 degrees) */
    float max_thrust;                    /* 2 (quadcopter.py:88) */
    float dof_action_speed_scale;        /* 8 pi (:283) */
    float thrust_action_speed_scale;     /* 200 (:287) */
    float drive_stiffness, drive_damping; /* DOF_MODE_POS drive, 1000 / 0 (:236-238) */
    float max_angular_velocity;          /* asset option, 4 pi (:208) */
    float init_height;                   /* default_pose.p.z = 1 (:226) */
    float clip_actions;
} MiQuadcopterParams;

/* "Articulation": ANY articulated robot the asset parser accepts, on the ground plane, with no task logic of its own -- what gym.load_asset of a
 * file with a new kinematic tree becomes (reference call sites: amp/humanoid_amp_base.py:177-200 mjcf/amp_humanoid.xml, franka_cube_stack.py:189-193
 * franka_panda_gripper.urdf, ...).  The robot is compiled at run time from its parsed description (isaacgymenvs_amd/assets/runtime.py); the stock
 * library carries the AMP humanoid.  Driven through mi_engine_simulate (gym.simulate) only: efforts in dof_actuation_force, position targets in
 * dof_position_targets for the dofs whose drive gains are non-zero (gym DOF_MODE_POS: dof stiffness / damping are the drive's gains).
 * mi_engine_step is refused -- the observation / reward functions of such a task stay the caller's (the reference's own torch code). */
/* The SCENE beside a fixed-base articulated actor: what the other actors of the reference's table-top envs become (franka_cube_stack.py:204-233,
 * 323-339: gym.create_box assets -- the table and its stand with fix_base_link, two free cubes -- created in every env beside the arm).  Free boxes
 * are rigid bodies of the same sub-step (contacts with the actor's collision spheres, the static boxes, the ground plane and each other are rows of
 * the one Gauss-Seidel solve, csrc/core/scene_engine.hpp); their root states live in tensor "scene_state" [N, MI_SCENE_MAX_FREE, 13]. */
#define MI_SCENE_MAX_FREE 4
#define MI_SCENE_MAX_STATIC 4
#define MI_SCENE_WARM_SLOTS 48 /* entries per env of tensor "scene_warm": one per contact slot of the scene solve (24 actor + 24 box contacts) */
typedef struct {
    int32_t n_free, n_static;              /* 0, 0: no scene -- the actor alone on the ground plane (its spheres against the plane) */
    int32_t arm_gravity;                   /* 0: asset option disable_gravity on the articulated actor (franka_cube_stack.py:199); the boxes feel the sim's */
    int32_t pad;
    float free_half[MI_SCENE_MAX_FREE][3]; /* half sizes */
    float free_mass[MI_SCENE_MAX_FREE];
    float free_inertia[MI_SCENE_MAX_FREE][3];  /* principal inertias along the box axes */
    float free_mu[MI_SCENE_MAX_FREE];      /* shape friction (combined with the other side's by averaging) */
    float free_init[MI_SCENE_MAX_FREE][7]; /* start pose (create_actor): position, quaternion xyzw */
    float static_pos[MI_SCENE_MAX_STATIC][3], static_quat[MI_SCENE_MAX_STATIC][4], static_half[MI_SCENE_MAX_STATIC][3], static_mu[MI_SCENE_MAX_STATIC];
    float arm_mu;                          /* friction of the actor's shapes */
} MiScene;

typedef struct {
    float kp[MI_MAX_DOF], kd[MI_MAX_DOF];  /* per-dof position-drive gains; kp = kd = 0: no drive on that dof */
    float max_angular_velocity;            /* asset option: clamp of the base's angular speed (rad/s); <= 0: none */
    float init_root[13];                   /* actor start pose (create_actor) + zero velocities */
    MiScene scene;                         /* free / static boxes beside the actor; fixed-base actors only */
    float drive_vmax[MI_MAX_DOF];          /* the asset's joint velocity limits (URDF <limit velocity=>); <= 0: none.  Scenes only: the solved joint velocities
                                            * are clamped to them (the simulator's maxJointVelocity), and a position drive's error is clamped to vmax * kd / kp,
                                            * the error at which its spring and damper balance at that speed (franka_panda_gripper.urdf:247 fingers: 0.2 m/s) */
} MiArticulationParams;

/* task parameters of Ingenuity (ingenuity.py:45-97, 233-282): constants the reference hard-codes in the task file */
typedef struct {
    float max_episode_length;            /* env.maxEpisodeLength */
    float dt;                            /* sim.dt */
    float thrust_upper_limit;            /* 2000 (ingenuity.py:91) */
    float thrust_lateral_component;      /* 0.2 (:92) */
    float thrust_action_speed_scale;     /* 2000 (:337) */
    float max_angular_velocity;          /* asset option, 4 pi (:248) */
    float init_height;                   /* default_pose.p.z = 1 (:254) */
    float rotor_speed;                   /* 50: speed the two visual rotors are given at every reset (:298-299) */
    int32_t target_period;               /* 500: a new target whenever progress_buf % 500 == 0 (:324) */
    float clip_actions;
} MiIngenuityParams;

/* task parameters of BallBalance (ball_balance.py:56-62, 136-306): the lengths of the generated asset and the constants the
 * reference hard-codes in the task file */
typedef struct {
    float max_episode_length;            /* env.maxEpisodeLength */
    float dt;                            /* sim.dt */
    float action_speed_scale;            /* env.actionSpeedScale */
    float dof_lower[6], dof_upper[6];    /* joint limits of the generated asset: +-45 degrees upper, -70 .. 90 degrees lower leg joints */
    float tray_height;                   /* bbot_pose.p.z (ball_balance.py:251-252) */
    float ball_init_pos[3];              /* (0.2, 0, 2) (:303-306) */
    float clip_actions;
    /* physics of the sub-step (csrc/core/bbot_engine.hpp) */
    float pin_stiffness, pin_damping;    /* attractors, 5e7 / 5e3 (:287-288) */
    float drive_kp, drive_kd;            /* DOF_MODE_POS drive of the actuated dofs, 4000 / 100 (:276-277) */
    int32_t actuated_mask;               /* bit d: dof d is position driven; dofs 1, 3, 5 (:271) */
    float ball_radius, ball_mass, ball_inertia, mu;   /* sphere radius 0.1, density 200 (:263-266); combined friction */
    float tray_radius, tray_half;        /* collision cylinder of the tray (:139-140) */
    float pin_offset[3];                 /* attractor offset in the lower leg's frame (:299) */
    float pin_target[3][3];              /* attractor targets, env frame (:293-297) */
    float sensor_pos[3][3];              /* force-sensor origins in the tray frame (:256-259) */
} MiBallBalanceParams;

/* scalars of compute_hand_reward (shadow_hand.py:746-756) */
typedef struct {
    float max_episode_length;
    float dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale;
    float success_tolerance, reach_goal_bonus, fall_dist, fall_penalty;
    int32_t max_consecutive_successes;
    float av_factor;
    int32_t ignore_z_rot;
} MiHandRewardParams;

/* task parameters of ShadowHand (shadow_hand.py:45-110, cfg/task/ShadowHand.yaml) and of AllegroHand (allegro_hand.py:42-128,
 * cfg/task/AllegroHand.yaml: the same fields; 16 driven dofs -> actuated[0..15], full state 88 wide, obs_type 0 / 2 / 3) */
typedef struct {
    MiHandRewardParams rew;
    float vel_obs_scale, force_torque_obs_scale;
    float reset_position_noise, reset_dof_pos_noise, reset_dof_vel_noise;
    float act_moving_average, dof_speed_scale, dt;
    int32_t use_relative_control;
    float clip_actions;
    float object_init_pos[3], goal_init_pos[3];
    float hand_pos[3], hand_quat[4];
    float cube_half, cube_mass, cube_inertia, mu;
    int32_t actuated[20];
    /* observationType (shadow_hand.py:97-110): 0 full_state (211), 1 openai (42), 2 full_no_vel (77), 3 full (157).
     * For types 1-3 obs_buf[:, k] = full_state[:, obs_map[k]] (the layouts of shadow_hand.py:472-526 are column subsets of
     * compute_full_state's); asymmetric_obs != 0 additionally exposes the full state as "states_buf" (:584). */
    int32_t obs_type, num_obs, asymmetric_obs;
    int16_t obs_map[160];
    /* random object forces (shadow_hand.py:69-72, 196-201, 700-708); force_scale <= 0 disables them */
    float force_scale, force_prob_range[2], force_decay, force_decay_interval;
    /* objectType (shadow_hand.py:86-96): 0 "block" (the cube_* fields), 2 "egg" = ellipsoid with semi-axes object_dims
     * (mjcf/open_ai_assets/hand/egg.xml:10), 1 "pen" = capsule along the object's z axis, object_dims = {radius, half length, -}
     * (pen.xml:20); object_inertia = principal inertias about the object's axes; cube_mass is the object's mass for every shape. */
    int32_t object_shape;
    float object_dims[3], object_inertia[3];
} MiHandParams;

typedef struct {
    int32_t num_obs, num_actions, num_dofs, num_bodies, num_sensors, num_contact_spheres, fixed_base, task_params_bytes;
} MiTaskInfo;

enum { MI_F32 = 0, MI_I64 = 1, MI_U8 = 2, MI_I32 = 3 };

typedef struct {
    char name[48];
    int32_t dtype;       /* MI_F32 ... */
    int32_t ndim;
    int64_t shape[4];
    int64_t stride[4];   /* in elements */
    int64_t byte_offset; /* from the arena base */
} MiTensorDesc;

/* ---- discovery ------------------------------------------------------------------------------------------- */
int mi_abi_version(void);
/* task in {"Cartpole","Ant","Humanoid","AnymalTerrain","ShadowHand","Anymal","Quadcopter","Ingenuity","BallBalance","AllegroHand"}: replaces isaacgym_task_map lookup (isaacgymenvs/tasks/__init__.py:88-114)
 * + gym.get_asset_{dof,rigid_body}_count (ant.py:155-156) */
int mi_task_info(const char* task, MiTaskInfo* out);
size_t mi_engine_arena_bytes(const char* task, int num_envs);

/* ---- lifecycle: replaces gym.create_sim / load_asset / create_env / create_actor / prepare_sim
 *      (vec_task.py:261-262, ant.py:116-212) and VecTask.allocate_buffers (vec_task.py:301-324) ------------- */
int mi_engine_create(const char* task, const MiSimParams* sim, const void* task_params, size_t task_params_bytes,
                     int num_envs, int env_id_offset /* global id of env 0: rank * num_envs */, uint64_t seed,
                     void* arena, size_t arena_bytes, MiEngine** out);
/* fills the arena: initial root/dof state, reset_buf = 1, progress = 0 ... (vec_task.py:310-323) */
int mi_engine_init_state(MiEngine* e, void* stream);
void mi_engine_destroy(MiEngine* e);

/* ---- tensor views: replaces gym.acquire_{actor_root_state,dof_state,force_sensor,dof_force}_tensor +
 *      gymtorch.wrap_tensor (ant.py:77-95, humanoid.py:78-98) and the VecTask buffers -------------------------- */
int mi_engine_num_tensors(const MiEngine* e);
int mi_engine_tensor_desc(const MiEngine* e, int index, MiTensorDesc* out);

/* ---- the hot path ------------------------------------------------------------------------------------------
 * One fused launch = VecTask.step (vec_task.py:360-408): clamp actions -> pre_physics_step ->
 * control_freq_inv x gym.simulate -> post_physics_step (progress++, reset flagged envs, observations, reward)
 * -> timeout mask.  `actions` is [num_envs, num_actions] row-major fp32. */
int mi_engine_step(MiEngine* e, const float* actions, void* stream);
/* reset_idx(env_ids) for explicit resets (ant.py:252-279, cartpole.py:144-157; used by VecTask.reset_done,
 * vec_task.py:440-455).  env_ids: int64 device pointer. */
int mi_engine_reset_idx(MiEngine* e, const int64_t* env_ids, int n, void* stream);
/* physics only: gym.simulate(sim) + refresh_* (vec_task.py:382; ant.py:233-235) with the efforts currently in
 * the "dof_actuation_force" tensor (gym.set_dof_actuation_force_tensor, ant.py:285) */
int mi_engine_simulate(MiEngine* e, void* stream);
/* gym.refresh_rigid_body_state_tensor(sim) (shadow_hand.py:440, anymal_terrain.py:314): fills the "rigid_body_state" tensor
 * [num_envs, num_bodies, 13] -- position, quaternion xyzw, linear velocity of the body frame's origin, angular velocity, world frame --
 * of every body of the articulation from the current root / dof state (gym.acquire_rigid_body_state_tensor, shadow_hand.py:150-175).
 * On demand only: step / simulate never touch the tensor. */
int mi_engine_refresh_rigid_body_states(MiEngine* e, void* stream);
/* gym.refresh_jacobian_tensors / gym.refresh_mass_matrix_tensors (franka_cube_stack.py:388-392,551-552; read by the operational-space
 * controller of :595-612) into CALLER tensors on the engine's device, row-major like the simulator's.  nv = num_dofs for a fixed base,
 * 6 + num_dofs otherwise; generalised velocity = [root linear velocity of the root frame's origin, root angular velocity (world frame;
 * floating bases only)] ++ dof velocities.
 *   jacobians     out [num_envs][num_bodies][6][nv]: rows 0-2 linear velocity of the body frame's origin, 3-5 angular velocity (world) per
 *                 unit generalised velocity -- J qd equals the velocity block of mi_engine_refresh_rigid_body_states.  (The simulator drops
 *                 the base link of a fixed-base actor: its tensor is out[:, 1:].)
 *   mass matrices out [num_envs][nv][nv]: the joint-space inertia the sub-step factors, joint armatures on the diagonal.
 * On demand only, from the current root / dof state. */
int mi_engine_compute_jacobians(MiEngine* e, float* out, void* stream);
int mi_engine_compute_mass_matrices(MiEngine* e, float* out, void* stream);
/* AnymalTerrain only: the terrain the reference builds with `Terrain(cfg["env"]["terrain"], num_envs)` and hands to
 * gym.add_triangle_mesh (anymal_terrain.py:203-215).  height_samples: DEVICE int16 [rows*cols] (Terrain.heightsamples,
 * row-major), env_origins: DEVICE fp32 [num_levels*num_terrains*3] (Terrain.env_origins).  Both stay owned by the
 * caller and must outlive the engine.  Must be called before mi_engine_init_state. */
int mi_engine_set_terrain(MiEngine* e, const int16_t* height_samples, int rows, int cols, float horizontal_scale,
                          float vertical_scale, float border_size, const float* env_origins, int num_levels,
                          int num_terrains, float env_length, int max_init_level);
/* optional knobs: "clip_obs" (env.clipObservations, vec_task.py:115), "control_freq_inv" (env.controlFrequencyInv, :111),
 * "gravity_x|y|z" (gym.set_sim_params after sim_params.gravity randomisation, vec_task.py:720-732),
 * "self_collision" 0 | 1 (the collision filter of gym.create_actor: humanoid.py:194 passes 0 = the actor collides with itself;
 * default 1 for the tasks whose arena carries "self_contact_impulse", rejected with 1 elsewhere),
 * "multi_wave" 0 | 32 (physics sub-step of Ant / Anymal / AnymalTerrain spread over the four waves of a workgroup of that many
 * envs -- same results up to summation order, see csrc/core/engine_mw.hpp; other tasks ignore it),
 * "fused_post" 0 | 1 (Ant on the limb-per-wave form: post_physics_step inside the step's sub-step launch -- with "fused_sub" the whole
 * control step is then ONE launch, the post step spread over the four role waves, csrc/mw_kernels.hpp loco_post_role; Humanoid on limb
 * waves: the step's last sub-step launch carries it on its role waves, csrc/mwc_kernels.hpp; same buffers),
 * "fused_sub" 0 | 1 (Ant / AnymalTerrain on the limb-per-wave form: all physics sub-steps of a control step -- vec_task.py:379-382,
 * anymal_terrain.py:443-451 -- in one launch, the state staying on chip between them; same buffers; HIP backend only),
 * "steps" (the control-step counter: observation-ring parity and per-step RNG counters; restore it together with the arena),
 * "actor_tensors" 0 | 1 (Ant, Humanoid, Anymal: whether the sub-step reads the per-env `actor_scale` / `dof_limit_shift` tensors of the
 * `actor_params` domain randomisation, vec_task.py:752-828; default 0 = the model's constants, no loads; ShadowHand always reads its own),
 * "terrain_slope_threshold" (AnymalTerrain: terrain.slopeTreshold of the reference's height-field -> triangle-mesh conversion,
 * anymal_terrain.py:576; steeper cell edges are levelled to their lower end in the ground query; 0 = off),
 * "terrain_walls" 0 | 1 (AnymalTerrain, default 1: the vertical faces that the slope correction gives the triangle mesh of
 * gym.add_triangle_mesh, anymal_terrain.py:198-211, collide from the side -- csrc/core/engine.hpp HeightfieldGround::contact),
 * "drive_force_limit" 0 | 1 (ShadowHand / AllegroHand, default 1: the position drives deliver at most the force range of their
 * actuators -- MJCF forcerange, shared.xml:250-269; the `effort` dof property, allegro_hand.py:264 -- the clamp being solved with the
 * joint-limit and contact rows, csrc/core/hand_engine.hpp drive_clamp_update; 0 = unlimited drives) */
int mi_engine_set_option(MiEngine* e, const char* key, double value);
/* Observation (which = 0) / action (which = 1) noise of the domain randomisation, applied inside the step kernels (replaces the
 * `noise_lambda` closures the reference builds in VecTask.apply_randomizations, vec_task.py:650-718, and runs as torch ops on the
 * buffers every step, :371-372,397-399).  value = op(x, corr + white): `white` ~ N(a, b) or U(a, b) fresh every step, `corr` a
 * per-(env, element) normal draw made once, scaled by (a_corr, b_corr); the host passes ranges already blended by the schedule.
 * dist 0 switches the noise off.  Cartpole, Ant, Humanoid, ShadowHand / AllegroHand (obs_buf only: states_buf stays clean). */
typedef struct MiNoiseParams {
    int32_t dist;          /* 0 off, 1 gaussian, 2 uniform */
    int32_t op;            /* 0 additive, 1 scaling */
    float a, b;            /* gaussian: mean, std; uniform: low, high */
    float a_corr, b_corr;  /* the same for the correlated part */
    uint32_t epoch;        /* stream id of the correlated draws (0: sampled once, as the reference does) */
} MiNoiseParams;
int mi_engine_set_noise(MiEngine* e, int which, const MiNoiseParams* params);
/* reads back any of the keys of mi_engine_set_option (replaces gym.get_sim_params / gym.get_frame_count, vec_task.py:620,723) */
int mi_engine_get_option(const MiEngine* e, const char* key, double* value);
/* which slot of the "obs_out" ring ([2, N, num_obs], clamped copy of obs_buf = what VecTask.step returns as
 * obs_dict["obs"], vec_task.py:402) the most recent step wrote */
int mi_engine_last_ring(const MiEngine* e);

/* ---- stand-alone replacements of the reference's @torch.jit.script functions (row-major contiguous
 *      [n, k] fp32 tensors, int64 reset/progress, same argument meaning and order as the reference) ---------- */
/* compute_ant_observations (ant.py:374-408) / compute_humanoid_observations (humanoid.py:378-413);
 * dof_force may be NULL for Ant.  potentials is in/out (the function returns the new potentials and the old
 * one as prev_potentials, ant.py:390-391). */
int mi_compute_locomotion_observations(const char* task, int n, const MiLocoParams* p, const float* root_states,
                                       const float* targets, float* potentials, float* prev_potentials,
                                       const float* inv_start_rot, const float* dof_pos, const float* dof_vel,
                                       const float* dof_force, const float* dof_limits_lower,
                                       const float* dof_limits_upper, const float* sensor_force_torques,
                                       const float* actions, const float* basis_vec0, const float* basis_vec1,
                                       float* obs_buf, float* up_vec, float* heading_vec, void* stream);
/* compute_ant_reward (ant.py:325-371) / compute_humanoid_reward (humanoid.py:323-375) */
int mi_compute_locomotion_reward(const char* task, int n, const MiLocoParams* p, const float* obs_buf,
                                 const int64_t* reset_buf_in, const int64_t* progress_buf, const float* actions,
                                 const float* potentials, const float* prev_potentials, float* rew_buf,
                                 int64_t* reset_buf_out, void* stream);
/* compute_cartpole_reward (cartpole.py:180-196) */
int mi_compute_cartpole_reward(int n, const MiCartpoleParams* p, const float* pole_angle, const float* pole_vel,
                               const float* cart_vel, const float* cart_pos, const int64_t* reset_buf_in,
                               const int64_t* progress_buf, float* rew_buf, int64_t* reset_buf_out, void* stream);

/* compute_quadcopter_reward (quadcopter.py:348-386): root_positions / root_linvels / root_angvels [n,3], root_quats [n,4]
 * (xyzw), reset_buf / progress_buf int64 [n] -> reward [n], reset int64 [n] */
int mi_compute_quadcopter_reward(int n, const float* root_positions, const float* root_quats, const float* root_linvels,
                                 const float* root_angvels, const int64_t* reset_buf_in, const int64_t* progress_buf,
                                 float max_episode_length, float* rew_buf, int64_t* reset_buf_out, void* stream);
/* compute_anymal_observations (anymal.py:354-386): root_states [n,13], commands [n,3], dof_pos / dof_vel / actions
 * [n,12] -> obs [n,48].  gravity_vec is the constant (0,0,-1) of anymal.py:143; default_dof_pos and the scales come
 * from p. */
int mi_compute_anymal_observations(int n, const MiAnymalFlatParams* p, const float* root_states, const float* commands,
                                   const float* dof_pos, const float* dof_vel, const float* actions, float* obs_buf,
                                   void* stream);
/* compute_anymal_reward (anymal.py:311-351): torques [n,12], contact_forces [n,num_bodies,3] (base = body 0, knees =
 * the *_THIGH bodies), episode_lengths int64 [n] -> rew [n], reset int64 [n] */
int mi_compute_anymal_reward(int n, const MiAnymalFlatParams* p, const float* root_states, const float* commands,
                             const float* torques, const float* contact_forces, int num_bodies,
                             const int64_t* episode_lengths, float* rew_buf, int64_t* reset_buf, void* stream);

/* ---- ShadowHand task functions (the hand/cube physics is not in the engine yet; these run on caller tensors) ------- */
/* compute_hand_reward (shadow_hand.py:746-800).  rew_buf out; reset_buf, reset_goal_buf, progress_buf (int64), successes
 * (fp32 [n]) and consecutive_successes (fp32 [1]) are updated IN PLACE = the tuple the jitted function returns.
 * workspace2: 2 floats of device scratch for the cross-env sums (:792-793). */
int mi_compute_hand_reward(int n, const MiHandRewardParams* p, float* rew_buf, int64_t* reset_buf, int64_t* reset_goal_buf,
                           int64_t* progress_buf, float* successes, float* consecutive_successes, const float* object_pos,
                           const float* object_rot, const float* target_pos, const float* target_rot, const float* actions,
                           int num_actions, float* workspace2, void* stream);
/* compute_full_state (shadow_hand.py:528-584): writes 3*num_dofs + 24 + 19*num_fingertips + num_actions floats per env
 * (211 for the Shadow hand) into obs_buf rows of `obs_stride` floats.  object_state [n,13], goal_pose [n,7],
 * fingertip_state [n,nf,13], fingertip_force_torque [n,6*nf]; all row-major contiguous. */
int mi_compute_hand_full_state(int n, int num_dofs, int num_fingertips, int num_actions, float vel_obs_scale,
                               float force_torque_obs_scale, const float* dof_pos, const float* dof_vel, const float* dof_force,
                               const float* dof_lower, const float* dof_upper, const float* object_state, const float* goal_pose,
                               const float* fingertip_state, const float* fingertip_force_torque, const float* actions,
                               float* obs_buf, int obs_stride, void* stream);
/* randomize_rotation (shadow_hand.py:803-806): out_quat [n,4] = q(rand0*pi, x_unit) * q(rand1*pi, y_unit) */
int mi_randomize_rotation(int n, const float* rand0, const float* rand1, const float* x_unit, const float* y_unit, float* out_quat,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Stand-alone replacements of the reference's remaining @torch.jit.script task functions (kernels in
 * csrc/kernels_jit_twins.hip).  Same argument order as the jitted signatures; every tensor is a contiguous row-major
 * device buffer of the shape the reference passes ([n,3] positions, [n,4] xyzw quaternions, int64 reset/progress buffers).
 * Buffers the reference only reads for their shape may be NULL where noted.  All return 0 or -1 (mi_last_error()).
 * ------------------------------------------------------------------------------------------------------------------ */

/* compute_bbot_reward (ball_balance.py:459-476); tray_positions is unused there and may be NULL */
int mi_compute_bbot_reward(int n, const float* tray_positions, const float* ball_positions, const float* ball_velocities, float ball_radius,
                           const int64_t* reset_buf_in, const int64_t* progress_buf, float max_episode_length, float* rew_buf,
                           int64_t* reset_buf_out, void* stream);
/* compute_ingenuity_reward (ingenuity.py:410-442); root_linvels and reset_buf_in are not read there and may be NULL */
int mi_compute_ingenuity_reward(int n, const float* root_positions, const float* target_root_positions, const float* root_quats,
                                const float* root_linvels, const float* root_angvels, const int64_t* reset_buf_in, const int64_t* progress_buf,
                                float max_episode_length, float* rew_buf, int64_t* reset_buf_out, void* stream);

typedef struct MiFrankaCabinetRewardParams {   /* float arguments of compute_franka_reward, franka_cabinet.py:494-496 */
    float dist_reward_scale, rot_reward_scale, around_handle_reward_scale, open_reward_scale;
    float finger_dist_reward_scale, action_penalty_scale, distX_offset, max_episode_length;
} MiFrankaCabinetRewardParams;
/* compute_franka_reward (franka_cabinet.py:488-553): actions [n,num_actions<=16], cabinet_dof_pos [n,num_cabinet_dofs>=4]
 * (column 3 = drawer_top_joint), the four *_axis tensors [n,3] */
int mi_compute_franka_cabinet_reward(int n, const MiFrankaCabinetRewardParams* p, const int64_t* reset_buf_in, const int64_t* progress_buf,
                                     const float* actions, int num_actions, const float* cabinet_dof_pos, int num_cabinet_dofs,
                                     const float* franka_grasp_pos, const float* drawer_grasp_pos, const float* franka_grasp_rot,
                                     const float* drawer_grasp_rot, const float* franka_lfinger_pos, const float* franka_rfinger_pos,
                                     const float* gripper_forward_axis, const float* drawer_inward_axis, const float* gripper_up_axis,
                                     const float* drawer_up_axis, float* rew_buf, int64_t* reset_buf_out, void* stream);
/* compute_grasp_transforms (franka_cabinet.py:556-568): two tf_combine's (torch_jit_utils.py:148) */
int mi_compute_grasp_transforms(int n, const float* hand_rot, const float* hand_pos, const float* franka_local_grasp_rot,
                                const float* franka_local_grasp_pos, const float* drawer_rot, const float* drawer_pos,
                                const float* drawer_local_grasp_rot, const float* drawer_local_grasp_pos, float* global_franka_rot,
                                float* global_franka_pos, float* global_drawer_rot, float* global_drawer_pos, void* stream);

/* axisangle2quat (franka_cube_stack.py:40-71): vec [n,3] -> quat [n,4] */
int mi_axisangle2quat(int n, const float* vec, float eps, float* quat, void* stream);
typedef struct MiFrankaCubeStackRewardParams {   /* reward_settings dict (franka_cube_stack.py:82-87, 461) + max_episode_length */
    float r_dist_scale, r_lift_scale, r_align_scale, r_stack_scale, table_height, max_episode_length;
} MiFrankaCubeStackRewardParams;
/* compute_franka_reward (franka_cube_stack.py:697-752); the `states` dict entries are passed as separate tensors:
 * cubeA_size / cubeB_size [n], the positions [n,3] */
int mi_compute_franka_cube_stack_reward(int n, const MiFrankaCubeStackRewardParams* p, const int64_t* reset_buf_in, const int64_t* progress_buf,
                                        const float* cubeA_size, const float* cubeB_size, const float* cubeA_pos, const float* cubeA_pos_relative,
                                        const float* eef_lf_pos, const float* eef_rf_pos, const float* cubeA_to_cubeB_pos, float* rew_buf,
                                        int64_t* reset_buf_out, void* stream);

/* AllegroHand: compute_hand_reward (allegro_hand.py:663-718) is the same function as shadow_hand.py:746-800 -> mi_compute_hand_reward.
 * randomize_rotation_pen (allegro_hand.py:728-732 == shadow_hand.py:809-813): rand1 and y_unit are unused by the reference but must be valid pointers */
int mi_randomize_rotation_pen(int n, const float* rand0, const float* rand1, float max_angle, const float* x_unit, const float* y_unit,
                              const float* z_unit, float* out_quat, void* stream);

/* Trifinger: lgsk_kernel (trifinger.py:1260-1274) elementwise over n floats */
int mi_lgsk_kernel(int n, const float* x, float scale, float eps, float* out, void* stream);
/* gen_keypoints (trifinger.py:1277-1290): pose rows of pose_stride >= 7 floats (pos3, quat4), size3 = 3 HOST floats -> keypoints [n,8,3] */
int mi_gen_keypoints(int n, const float* pose, int pose_stride, const float* size3, float* keypoints, void* stream);
typedef struct MiTrifingerRewardParams {   /* scalar arguments of compute_trifinger_reward, trifinger.py:1296-1308 */
    int episode_length;
    float dt, finger_move_penalty_weight, finger_reach_object_weight, object_dist_weight, object_rot_weight;
    int64_t env_steps_count;
    int use_keypoints;
    float keypoint_size[3];    /* gen_keypoints' default (0.065, 0.065, 0.065) */
} MiTrifingerRewardParams;
/* compute_trifinger_reward (trifinger.py:1292-1383): object_goal_poses [n,7], object states [n,13], fingertip states [n,3,13];
 * obs_buf / reset_buf inputs of the reference are not read there and are not passed; the two info arrays [n] may be NULL */
int mi_compute_trifinger_reward(int n, const MiTrifingerRewardParams* p, const int64_t* progress_buf, const float* object_goal_poses,
                                const float* object_state, const float* last_object_state, const float* fingertip_state,
                                const float* last_fingertip_state, float* rew_buf, int64_t* reset_buf_out, float* info_finger_movement_penalty,
                                float* info_finger_reach_object_reward, void* stream);
/* compute_trifinger_observations_states (trifinger.py:1386-1420): obs_buf [n, 2 num_dofs + 14 + num_actions]; states_buf (may be NULL)
 * [n, obs + 6 + fingertip_state_cols + num_dofs + tip_wrench_cols] when asymmetric_obs, else a copy of obs_buf */
int mi_compute_trifinger_observations_states(int n, int asymmetric_obs, int num_dofs, int num_actions, int fingertip_state_cols,
                                             int tip_wrench_cols, const float* dof_position, const float* dof_velocity, const float* object_state,
                                             const float* object_goal_poses, const float* actions, const float* fingertip_state,
                                             const float* joint_torques, const float* tip_wrenches, float* obs_buf, float* states_buf,
                                             void* stream);

/* Trifinger cuboid-pose samplers (trifinger.py:1427-1512).  The reference draws from torch.rand / torch.randn inside the jitted function; these
 * take the SAME draws as a tensor (columns in the order the reference calls the generator) and apply the same map:
 * random_xy: rand2 [n,2] (radius draw, angle draw) -> xy [n,2];  random_z: rand1 [n] -> z [n];  default_orientation -> quat [n,4];
 * random_orientation: randn4 [n,4] -> quat;  random_orientation_within_angle: rand3 [n,3], base [n,4] -> quat;
 * random_angular_vel: randn4 [n,4] = axis draws 3 + magnitude draw 1 -> angvel [n,3];  random_yaw_orientation: rand1 [n] -> quat */
int mi_trifinger_random_xy(int n, const float* rand2, float max_com_distance_to_center, float* xy, void* stream);
int mi_trifinger_random_z(int n, const float* rand1, float min_height, float max_height, float* z, void* stream);
int mi_trifinger_default_orientation(int n, float* quat, void* stream);
int mi_trifinger_random_orientation(int n, const float* randn4, float* quat, void* stream);
int mi_trifinger_random_orientation_within_angle(int n, const float* rand3, const float* base, float max_angle, float* quat, void* stream);
int mi_trifinger_random_angular_vel(int n, const float* randn4, float magnitude_stdev, float* angvel, void* stream);
int mi_trifinger_random_yaw_orientation(int n, const float* rand1, float* quat, void* stream);

/* HumanoidAMP: dof_to_obs (amp/humanoid_amp_base.py:462-492): pose [n,28] -> dof_obs [n,52] */
int mi_amp_dof_to_obs(int n, const float* pose, float* dof_obs, void* stream);
/* compute_humanoid_observations (amp/humanoid_amp_base.py:494-528) == build_amp_observations (humanoid_amp.py:299-330):
 * key_body_pos [n,num_key_bodies<=8,3] -> obs [n, 13 + 52 + 28 + 3 num_key_bodies] */
int mi_compute_humanoid_amp_observations(int n, const float* root_states, const float* dof_pos, const float* dof_vel, const float* key_body_pos,
                                         int num_key_bodies, int local_root_obs, float* obs, void* stream);
/* compute_humanoid_reward (amp/humanoid_amp_base.py:530-534): ones; obs_buf is read for its shape only and may be NULL */
int mi_compute_humanoid_amp_reward(int n, const float* obs_buf, float* rew_buf, void* stream);
/* compute_humanoid_reset (amp/humanoid_amp_base.py:536-564): contact_buf / rigid_body_pos [n,num_bodies<=64,3]; contact_body_ids is a
 * HOST int64 list; reset_buf_in is read for its shape only and may be NULL */
int mi_compute_humanoid_amp_reset(int n, const int64_t* reset_buf_in, const int64_t* progress_buf, const float* contact_buf,
                                  const int64_t* contact_body_ids, int num_contact_body_ids, const float* rigid_body_pos, int num_bodies,
                                  float max_episode_length, int enable_early_termination, float termination_height, int64_t* reset_buf_out,
                                  int64_t* terminated_out, void* stream);

typedef struct MiDextremeRewardParams {   /* scalar arguments of compute_hand_reward, dextreme/allegro_hand_dextreme.py:1598-1606 */
    float max_episode_length, dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale, action_delta_penalty_scale;
    float success_tolerance, reach_goal_bonus, fall_dist, fall_penalty;
    int max_consecutive_successes;
    float av_factor;
    int num_success_hold_steps;
} MiDextremeRewardParams;
/* compute_hand_reward (dextreme/allegro_hand_dextreme.py:1598-1663), in place on reset_buf / reset_goal_buf / progress_buf /
 * hold_count_buf / successes / consecutive_successes[1] like the reference's return tuple; reward_terms8 (may be NULL) [8,n] =
 * dist_rew, rot_rew, action_penalty, action_delta_penalty, velocity_penalty, reach_goal_rew, fall_rew, timeout_rew;
 * workspace2 = 2 device floats of scratch */
int mi_compute_hand_reward_dextreme(int n, const MiDextremeRewardParams* p, float* rew_buf, int64_t* reset_buf, int64_t* reset_goal_buf,
                                    int64_t* progress_buf, int64_t* hold_count_buf, const float* cur_targets, const float* prev_targets,
                                    const float* hand_dof_vel, int num_dofs, float* successes, float* consecutive_successes,
                                    const float* object_pos, const float* object_rot, const float* target_pos, const float* target_rot,
                                    const float* actions, int num_actions, float* reward_terms8, float* workspace2, void* stream);

/* Measurement aid (bench.py "box"; no reference counterpart): how fast THIS device runs the two things the step kernels are bound by -- the
 * issue rate of a lone wave (a chain of `fma_iters` dependent v_fma_f32 on one wave) and the latency of dependent loads (`hops` pointer-chase
 * steps through a `chase_bytes` buffer, a device scratch buffer of at least that size, >= 256 KB, whose contents are overwritten).  out6[0] = us per
 * 1000 dependent FMAs on a lone wave, out6[1] = ns per dependent load, out6[2] = us per 1000 dependent FMAs with one such wave on every
 * SIMD of the chip (the clocks under load), out6[3] = ns per workgroup barrier of a four-wave workgroup with an LDS word exchanged (one
 * workgroup per CU), out6[4] = ns per dependent LDS read, out6[5] = ns per dependent
 * v_rcp_f32 + v_sin_f32 pair (a wave on every SIMD); host floats, the call synchronises the stream.  The boxes of one pool differ by up to
 * 1.5x on the engine's kernels; these two numbers say which kind of box a benchmark line came from. */
int mi_device_probe(void* scratch, long long chase_bytes, int fma_iters, int hops, float* out6, void* stream);

/* Test aid: fills the LDS of every CU with `pattern` (e.g. a NaN's bits).  LDS keeps what the last kernel on a CU left there; a step kernel that
 * read a slot before writing it would depend on what ran before it -- the benchmark-size parity tests poison the LDS first. */
int mi_debug_poison_lds(unsigned pattern, void* stream);

const char* mi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_ENGINE_H */

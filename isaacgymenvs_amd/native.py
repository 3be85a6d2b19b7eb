"""ctypes binding of libmi_engine.so (C ABI: include/mi_engine.h) + its hipcc build recipe.

PyTorch is only plumbing here: it owns the device arena and the stream; every compute entry point is a raw-pointer
C call.  There is no CPU fallback: if the library is missing or no ROCm device is visible the constructors raise.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_ENGINE_LIB lets tools/ab_bench.sh time two builds of the library inside one GPU session (never set in tests)
LIB_PATH = os.environ.get("MI_ENGINE_LIB") or os.path.join(_HERE, "libmi_engine.so")
# CPU product backend (sim_device="cpu": the reference's CPU pipeline, BASELINE config 1), built with g++ from the same engine sources
CPU_LIB_PATH = os.path.join(_HERE, "libmi_engine_cpu.so")
CPU_TASKS = ("Cartpole", "Ant", "Humanoid", "AnymalTerrain", "ShadowHand", "Anymal", "Quadcopter", "Ingenuity", "BallBalance", "AllegroHand",
             "Articulation")
CSRC = os.path.join(_HERE, "csrc")
BUILD_DIR = os.path.join(CSRC, "build")
# one translation unit per robot model (they compile in parallel) + the C ABI
SOURCES = ["mi_engine.hip", "kernels_cartpole.hip", "kernels_ant.hip", "kernels_humanoid.hip", "kernels_anymal.hip", "kernels_shadow_hand.hip",
           "kernels_shadow_hand_pen.hip", "kernels_shadow_hand_egg.hip", "kernels_quadcopter.hip", "kernels_ingenuity.hip", "kernels_ball_balance.hip", "kernels_jit_twins.hip",
           "kernels_mw_ant.hip", "kernels_mw_anymal.hip", "kernels_humanoid_sc2.hip", "kernels_humanoid_mwc.hip",
           "kernels_shadow_hand_mw.hip", "kernels_shadow_hand_mw_pen.hip", "kernels_shadow_hand_mw_egg.hip", "kernels_body_states.hip",
           # the sub-step kernels that read the `actor_params` factor tensors (Sim<Scaled<M>>): their own objects, beside the plain ones
           # AllegroHand: the hand task kernels instantiated for the Allegro model, one object shape per translation unit
           "kernels_allegro_hand.hip", "kernels_allegro_hand_pen.hip", "kernels_allegro_hand_egg.hip",
           "kernels_allegro_hand_mw.hip", "kernels_allegro_hand_mw_pen.hip", "kernels_allegro_hand_mw_egg.hip",
           "kernels_scaled_ant.hip", "kernels_scaled_humanoid.hip", "kernels_scaled_humanoid_mwc.hip", "kernels_scaled_humanoid_sc2.hip",
           "kernels_scaled_anymal.hip",
           # the ShadowHand's sub-step kernels that read per-body link-mass factors (option hand_body_mass), one shape per unit, both forms
           "kernels_scaled_shadow_hand_box.hip", "kernels_scaled_shadow_hand_pen.hip", "kernels_scaled_shadow_hand_egg.hip",
           "kernels_scaled_shadow_hand_mw_box.hip", "kernels_scaled_shadow_hand_mw_pen.hip", "kernels_scaled_shadow_hand_mw_egg.hip",
           # the Articulation task: the ONE translation unit that holds the run-time-compiled robot (assets/runtime.py rebuilds only this)
           "kernels_articulation.hip"]
MI_MAX_DOF = 32
# include/mi_engine.h MI_ABI_VERSION: bumped whenever the arena layout, a parameter struct or an export changes (2: round 4 -- cumulative
# episode statistics tensors, per-body actor scales, rigid_body_state; 3: round 5 -- compensated cumulative statistics (episode_cum_stats [32],
# reward_workspace [8]), hand_pair_count, hand_body_mass_scale; 4: round 5, scenes -- MiArticulationParams.scene / drive_vmax, scene_state, scene_contacts,
# scene_warm).  A library of another version is refused when it is loaded, and a state
# checkpoint (VecTask.get_env_state) carries the version + arena size it was taken with.
MI_ABI_VERSION = 4

# -fno-slp-vectorize: pairing scalars into v_pk_* ops lengthens live ranges in the fully unrolled sub-step
# (Ant: 230 -> 40 spilled VGPRs without it)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-signed-zeros", "-fno-trapping-math",
               "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("MI_EXTRA_HIPCC_FLAGS", "").split()
# more spilled SGPRs than this in a physics kernel fails the build: heavy SGPR spilling was the regime in which
# gfx950 builds of the sub-step returned run-to-run different results ("compiler regime")
MAX_SGPR_SPILL = 160
# Per kernel family (fragment of the mangled name, first match wins), what the build accepts -- the measured count plus head room, so that an edit
# which moves a kernel towards that regime fails the BUILD, not a benchmark (one 160-SGPR gate let the Humanoid's limb-wave kernels, at
# 84 / 79 spilled SGPRs, pass as "the hot kernels spill 0").  Measured, round 5: the one-launch kernels of Ant / ANYmal 0-4, the hands' finger waves
# 0, the Humanoid's limb waves 84 (sub-step) / 79 (sub-step + post step) and 74 in the Sim<Scaled<M>> form, the one-wave forms (small batches, CPU
# twin) 10-155 (round 6: 162 for the ANYmal's height-field one-wave kernel once the friction step of a sliding contact became isotropic -- core/engine.hpp
# friction_disc: a few more literals per contact block; the small-batch form, budget 176).  Anything unnamed falls under MAX_SGPR_SPILL.
SGPR_SPILL_BUDGETS = [
    ("substep_mw_fused", 8), ("substep_mw_kernel", 8), ("substep_mw_post_kernel", 8),                 # Ant, ANYmal: limb per wave
    ("hand_substep_mw", 8),                                                                           # ShadowHand / AllegroHand: finger per wave
    ("substep_mwc_post_kernel", 100), ("substep_mwc_kernel", 100),                                    # Humanoid: limb per wave on the compact store
    # the one-wave forms: small batches, tasks without a multi-wave form, the run-time robot of the Articulation task (its 160 is the old gate)
    ("articulation_substep_kernel", 160), ("hand_substep_kernel", 120), ("substep_sc2_kernel", 120), ("substep_kernel", 176),
]


def sgpr_spill_budget(kernel_name):
    for frag, cap in SGPR_SPILL_BUDGETS:
        if frag in kernel_name:
            return cap
    return MAX_SGPR_SPILL


def over_sgpr_budget(usage):
    """{kernel: usage} of the physics kernels whose spilled-SGPR count exceeds their family's budget"""
    return {k: dict(u, budget=sgpr_spill_budget(k)) for k, u in usage.items() if "substep" in k and u.get("SGPRs Spill", 0) > sgpr_spill_budget(k)}


class MiSimParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("substeps", C.c_int32), ("iters", C.c_int32), ("gravity", C.c_float * 3),
                ("contact_offset", C.c_float), ("rest_offset", C.c_float), ("max_depen_vel", C.c_float),
                ("erp", C.c_float), ("plane_mu", C.c_float), ("ground_z", C.c_float), ("cfm", C.c_float),
                ("warm", C.c_float)]


class MiLocoParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "dt", "dof_vel_scale", "contact_force_scale", "angular_velocity_scale", "power_scale", "heading_weight",
        "up_weight", "actions_cost", "energy_cost", "joints_at_limit_cost", "death_cost", "termination_height",
        "max_episode_length", "clip_actions", "max_motor_effort", "start_height")] + [
        ("gear", C.c_float * MI_MAX_DOF), ("dof_lower", C.c_float * MI_MAX_DOF), ("dof_upper", C.c_float * MI_MAX_DOF),
        ("initial_dof_pos", C.c_float * MI_MAX_DOF), ("targets", C.c_float * 3), ("inv_start_rot", C.c_float * 4),
        ("basis_vec0", C.c_float * 3), ("basis_vec1", C.c_float * 3), ("reset_pos_noise", C.c_float),
        ("reset_vel_noise", C.c_float)]


class MiCartpoleParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("reset_dist", "max_push_effort", "max_episode_length", "clip_actions")]


class MiAnymalParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "lin_vel_scale", "ang_vel_scale", "dof_pos_scale", "dof_vel_scale", "height_meas_scale", "action_scale",
        "rew_termination", "rew_lin_vel_xy", "rew_lin_vel_z", "rew_ang_vel_z", "rew_ang_vel_xy", "rew_orient", "rew_torque",
        "rew_joint_acc", "rew_base_height", "rew_air_time", "rew_collision", "rew_stumble", "rew_action_rate", "rew_hip")] + [
        ("command_x", C.c_float * 2), ("command_y", C.c_float * 2), ("command_yaw", C.c_float * 2),
        ("base_init_state", C.c_float * 13), ("default_dof_pos", C.c_float * 12),
        ("kp", C.c_float), ("kd", C.c_float), ("torque_limit", C.c_float), ("dt", C.c_float),
        ("max_episode_length_s", C.c_float), ("max_episode_length", C.c_int32), ("push_interval", C.c_int32),
        ("allow_knee_contacts", C.c_int32), ("decimation", C.c_int32), ("add_noise", C.c_int32)] + [
        (n, C.c_float) for n in ("noise_lin_vel", "noise_ang_vel", "noise_gravity", "noise_dof_pos", "noise_dof_vel",
                                 "noise_height")] + [
        ("curriculum", C.c_int32), ("clip_actions", C.c_float), ("friction_range", C.c_float * 2), ("terrain_mu", C.c_float)]


class MiAnymalFlatParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "lin_vel_scale", "ang_vel_scale", "dof_pos_scale", "dof_vel_scale", "action_scale",
        "rew_lin_vel_xy", "rew_ang_vel_z", "rew_torque")] + [
        ("command_x", C.c_float * 2), ("command_y", C.c_float * 2), ("command_yaw", C.c_float * 2),
        ("base_init_state", C.c_float * 13), ("default_dof_pos", C.c_float * 12),
        ("kp", C.c_float), ("kd", C.c_float), ("torque_limit", C.c_float), ("max_episode_length", C.c_int32),
        ("clip_actions", C.c_float)]


class MiQuadcopterParams(C.Structure):
    _fields_ = [("max_episode_length", C.c_float), ("dt", C.c_float), ("dof_lower", C.c_float * 8), ("dof_upper", C.c_float * 8),
                ("max_thrust", C.c_float), ("dof_action_speed_scale", C.c_float), ("thrust_action_speed_scale", C.c_float),
                ("drive_stiffness", C.c_float), ("drive_damping", C.c_float), ("max_angular_velocity", C.c_float),
                ("init_height", C.c_float), ("clip_actions", C.c_float)]


class MiIngenuityParams(C.Structure):
    _fields_ = [("max_episode_length", C.c_float), ("dt", C.c_float), ("thrust_upper_limit", C.c_float),
                ("thrust_lateral_component", C.c_float), ("thrust_action_speed_scale", C.c_float), ("max_angular_velocity", C.c_float),
                ("init_height", C.c_float), ("rotor_speed", C.c_float), ("target_period", C.c_int32), ("clip_actions", C.c_float)]


class MiBallBalanceParams(C.Structure):
    _fields_ = [("max_episode_length", C.c_float), ("dt", C.c_float), ("action_speed_scale", C.c_float), ("dof_lower", C.c_float * 6),
                ("dof_upper", C.c_float * 6), ("tray_height", C.c_float), ("ball_init_pos", C.c_float * 3), ("clip_actions", C.c_float),
                ("pin_stiffness", C.c_float), ("pin_damping", C.c_float), ("drive_kp", C.c_float), ("drive_kd", C.c_float),
                ("actuated_mask", C.c_int32), ("ball_radius", C.c_float), ("ball_mass", C.c_float), ("ball_inertia", C.c_float),
                ("mu", C.c_float), ("tray_radius", C.c_float), ("tray_half", C.c_float), ("pin_offset", C.c_float * 3),
                ("pin_target", (C.c_float * 3) * 3), ("sensor_pos", (C.c_float * 3) * 3)]


MI_SCENE_MAX_FREE, MI_SCENE_MAX_STATIC = 4, 4


class MiScene(C.Structure):
    """include/mi_engine.h MiScene: the free / static boxes beside a fixed-base articulated actor (csrc/core/scene_engine.hpp)"""
    _fields_ = [("n_free", C.c_int32), ("n_static", C.c_int32), ("arm_gravity", C.c_int32), ("pad", C.c_int32),
                ("free_half", (C.c_float * 3) * MI_SCENE_MAX_FREE), ("free_mass", C.c_float * MI_SCENE_MAX_FREE),
                ("free_inertia", (C.c_float * 3) * MI_SCENE_MAX_FREE), ("free_mu", C.c_float * MI_SCENE_MAX_FREE),
                ("free_init", (C.c_float * 7) * MI_SCENE_MAX_FREE),
                ("static_pos", (C.c_float * 3) * MI_SCENE_MAX_STATIC), ("static_quat", (C.c_float * 4) * MI_SCENE_MAX_STATIC),
                ("static_half", (C.c_float * 3) * MI_SCENE_MAX_STATIC), ("static_mu", C.c_float * MI_SCENE_MAX_STATIC), ("arm_mu", C.c_float)]


class MiArticulationParams(C.Structure):
    _fields_ = [("kp", C.c_float * MI_MAX_DOF), ("kd", C.c_float * MI_MAX_DOF), ("max_angular_velocity", C.c_float), ("init_root", C.c_float * 13),
                ("scene", MiScene), ("drive_vmax", C.c_float * MI_MAX_DOF)]


class MiHandRewardParams(C.Structure):
    _fields_ = [("max_episode_length", C.c_float), ("dist_reward_scale", C.c_float), ("rot_reward_scale", C.c_float),
                ("rot_eps", C.c_float), ("action_penalty_scale", C.c_float), ("success_tolerance", C.c_float),
                ("reach_goal_bonus", C.c_float), ("fall_dist", C.c_float), ("fall_penalty", C.c_float),
                ("max_consecutive_successes", C.c_int32), ("av_factor", C.c_float), ("ignore_z_rot", C.c_int32)]


class MiHandParams(C.Structure):
    _fields_ = [("rew", MiHandRewardParams), ("vel_obs_scale", C.c_float), ("force_torque_obs_scale", C.c_float),
                ("reset_position_noise", C.c_float), ("reset_dof_pos_noise", C.c_float), ("reset_dof_vel_noise", C.c_float),
                ("act_moving_average", C.c_float), ("dof_speed_scale", C.c_float), ("dt", C.c_float),
                ("use_relative_control", C.c_int32), ("clip_actions", C.c_float), ("object_init_pos", C.c_float * 3),
                ("goal_init_pos", C.c_float * 3), ("hand_pos", C.c_float * 3), ("hand_quat", C.c_float * 4),
                ("cube_half", C.c_float), ("cube_mass", C.c_float), ("cube_inertia", C.c_float), ("mu", C.c_float),
                ("actuated", C.c_int32 * 20),
                ("obs_type", C.c_int32), ("num_obs", C.c_int32), ("asymmetric_obs", C.c_int32), ("obs_map", C.c_int16 * 160),
                ("force_scale", C.c_float), ("force_prob_range", C.c_float * 2), ("force_decay", C.c_float),
                ("force_decay_interval", C.c_float),
                ("object_shape", C.c_int32), ("object_dims", C.c_float * 3), ("object_inertia", C.c_float * 3)]


class MiFrankaCabinetRewardParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale",
                                         "finger_dist_reward_scale", "action_penalty_scale", "distX_offset", "max_episode_length")]


class MiFrankaCubeStackRewardParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("r_dist_scale", "r_lift_scale", "r_align_scale", "r_stack_scale", "table_height",
                                         "max_episode_length")]


class MiTrifingerRewardParams(C.Structure):
    _fields_ = [("episode_length", C.c_int32), ("dt", C.c_float), ("finger_move_penalty_weight", C.c_float),
                ("finger_reach_object_weight", C.c_float), ("object_dist_weight", C.c_float), ("object_rot_weight", C.c_float),
                ("env_steps_count", C.c_int64), ("use_keypoints", C.c_int32), ("keypoint_size", C.c_float * 3)]


class MiDextremeRewardParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("max_episode_length", "dist_reward_scale", "rot_reward_scale", "rot_eps", "action_penalty_scale",
                                         "action_delta_penalty_scale", "success_tolerance", "reach_goal_bonus", "fall_dist", "fall_penalty")] + \
               [("max_consecutive_successes", C.c_int32), ("av_factor", C.c_float), ("num_success_hold_steps", C.c_int32)]


class MiNoiseParams(C.Structure):
    _fields_ = [("dist", C.c_int32), ("op", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("a_corr", C.c_float), ("b_corr", C.c_float),
                ("epoch", C.c_uint32)]


class MiTaskInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_obs", "num_actions", "num_dofs", "num_bodies", "num_sensors",
                                         "num_contact_spheres", "fixed_base", "task_params_bytes")]


class MiTensorDesc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("stride", C.c_int64 * 4), ("byte_offset", C.c_int64)]


# every symbol include/mi_engine.h declares (checked by tests/test_abi.py)
EXPORTS = ["mi_abi_version", "mi_task_info", "mi_engine_arena_bytes", "mi_engine_create", "mi_engine_init_state",
           "mi_engine_destroy", "mi_engine_num_tensors", "mi_engine_tensor_desc", "mi_engine_step",
           "mi_engine_reset_idx", "mi_engine_simulate", "mi_engine_refresh_rigid_body_states", "mi_engine_compute_jacobians", "mi_engine_compute_mass_matrices", "mi_engine_set_option", "mi_engine_get_option", "mi_engine_set_noise", "mi_engine_last_ring", "mi_engine_set_terrain",
           "mi_compute_locomotion_observations", "mi_compute_locomotion_reward", "mi_compute_cartpole_reward",
           "mi_compute_hand_reward", "mi_compute_hand_full_state", "mi_randomize_rotation",
           "mi_compute_anymal_observations", "mi_compute_anymal_reward", "mi_compute_quadcopter_reward",
           "mi_compute_bbot_reward", "mi_compute_ingenuity_reward", "mi_compute_franka_cabinet_reward", "mi_compute_grasp_transforms",
           "mi_axisangle2quat", "mi_compute_franka_cube_stack_reward", "mi_randomize_rotation_pen", "mi_lgsk_kernel", "mi_gen_keypoints",
           "mi_compute_trifinger_reward", "mi_compute_trifinger_observations_states", "mi_trifinger_random_xy", "mi_trifinger_random_z",
           "mi_trifinger_default_orientation", "mi_trifinger_random_orientation", "mi_trifinger_random_orientation_within_angle",
           "mi_trifinger_random_angular_vel", "mi_trifinger_random_yaw_orientation", "mi_amp_dof_to_obs",
           "mi_compute_humanoid_amp_observations", "mi_compute_humanoid_amp_reward", "mi_compute_humanoid_amp_reset", "mi_compute_hand_reward_dextreme",
           "mi_device_probe", "mi_debug_poison_lds", "mi_last_error"]


def auto_multi_wave(task, num_envs):
    """Launch shape of the physics sub-step (csrc/core/engine_mw.hpp, engine_mwc.hpp, hand_engine_mw.hpp), envs per workgroup; 0 = one wave
    per workgroup.  Locomotion: the limb-per-wave form while its 4 * N / E waves still find a SIMD each (1024 on an MI355X) -- measured 1.4x
    faster at 4096 envs, slower from 16384 envs on (profiles/r2b_mw_ab.txt); tasks without a multi-wave form ignore the option."""
    mw = 16 if num_envs <= 4096 else (32 if num_envs <= 8192 else 0)
    if task == "AnymalTerrain" and 8192 < num_envs <= 16384:
        mw = 32          # with its five sim steps in one launch the leg waves still win at 16384 envs: 0.260 against 0.294 ms per step (profiles/r3zz_env_sweep.txt)
    if task == "Humanoid":
        mw = 32          # limb waves + pair wave (csrc/mwc_kernels.hpp), at any env count
    if task in ("ShadowHand", "AllegroHand"):
        # finger per wave (csrc/hand_mw_kernels.hpp): 32 envs per workgroup (two half-filled waves per SIMD) while that fills the chip (one
        # workgroup per CU up to 8192 envs), full 64-env waves (half the wave instructions per env) from there on -- profiles/r3l_hand_mw_ab_0_32_64.txt
        mw = 64 if num_envs >= 8192 else 32
    return mw


def select_multi_wave(engine, task, num_envs, mw="auto"):
    """sets the engine's multi_wave option ("auto": auto_multi_wave); a value the task's kernels do not take falls back to 32"""
    if mw == "auto":
        mw = auto_multi_wave(task, num_envs)
    for cand in (int(mw), 32):
        try:
            engine.set_option("multi_wave", cand)
            break
        except RuntimeError:
            continue
    if task in ("Ant", "AnymalTerrain", "Anymal"):
        # all physics sub-steps of a control step in one launch (csrc/mw_kernels.hpp substep_mw_fused_kernel; bit-identical buffers):
        # Ant@4096 0.0424 -> 0.0408 ms, AnymalTerrain@4096 0.1273 -> 0.1187 ms per step (tools/fused_sub_ab.py, profiles/r3x_fused_sub_ab.txt)
        try:
            engine.set_option("fused_sub", 1)
        except RuntimeError:
            pass
    if task == "Humanoid":
        # post_physics_step on the role waves of the step's last sub-step launch (csrc/mwc_kernels.hpp substep_mwc_post_kernel; bit-identical
        # buffers): Humanoid@8192 0.1805 -> 0.1695, @4096 0.1758 -> 0.1704, @16384 0.3488 -> 0.3405 ms per step (profiles/r4j_humanoid_fused_post_ab.txt)
        try:
            engine.set_option("fused_post", 1)
        except RuntimeError:
            pass
    if task == "Ant":
        # post_physics_step inside the step's ONE launch, spread over the four role waves (csrc/mw_kernels.hpp loco_post_role; bit-identical
        # buffers): Ant@1024 0.0394 -> 0.0351, @4096 0.0394 -> 0.0365, @8192 0.0478 -> 0.0431 ms per step (profiles/r4i_ant_fused_post_ab.txt).
        # (Round 3's form -- the whole post step on ONE wave of the last sub-step launch -- only paid below 2048 envs, profiles/r3r_*, r3s_*.)
        try:
            engine.set_option("fused_post", 1)
        except RuntimeError:
            pass


    # (AnymalTerrain has a "fused_post" form too -- 39 of the post pass's 48 observation columns written by the height-scan kernel's threads, one per
    #  (env, column): csrc/kernels_anymal.hip, bit-identical -- measured at +1.5 % on a slow box, +0.5 % on a fast one at 4096 envs, -4.5 % at 16384, with
    #  27 % more HBM-side traffic (every column thread re-reads its inputs): left off, profiles/r4t_anymal_obs_columns_ab.txt)


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    root = os.path.join(_HERE, "csrc")
    for d, _, fs in os.walk(root):
        if os.path.abspath(d).startswith(os.path.abspath(BUILD_DIR)):
            continue
        for f in fs:
            if os.path.getmtime(os.path.join(d, f)) > t:
                return True
    return os.path.getmtime(os.path.join(_HERE, "..", "include", "mi_engine.h")) > t


def _newest_include(src):
    """newest mtime among the files `src` reaches through #include "..." (relative to the including file or to csrc/), itself excluded"""
    import re
    seen, todo, newest = set(), [src], 0.0
    while todo:
        path = todo.pop()
        if path in seen or not os.path.exists(path):
            continue
        seen.add(path)
        if path != src:
            newest = max(newest, os.path.getmtime(path))
        with open(path) as f:
            text = f.read()
        for m in re.finditer(r'#include\s+"([^"]+)"', text):
            for base in (os.path.dirname(path), CSRC):
                q = os.path.normpath(os.path.join(base, m.group(1)))
                if os.path.exists(q):
                    todo.append(q)
                    break
    return newest


def _obj_stale(src, obj, newest_header):
    return (not os.path.exists(obj)) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header)


def build(force=False, verbose=False):
    """Generate the per-robot constexpr headers and compile the HIP library for gfx950 (cross-compiles w/o a GPU).

    Every translation unit is compiled by its own hipcc process (in parallel), then linked into libmi_engine.so.
    The per-kernel resource usage (VGPRs, spills, scratch) that hipcc reports lands in csrc/build/*.log and is
    summarised in csrc/build/resource_usage.txt -- spilled SGPRs are treated as a build error for the step kernels
    (an SGPR-spilling build of these kernels was observed to miscompute on gfx950).
    """
    from .registry import generate_headers
    generate_headers()
    cpu_job = build_cpu(force=force, wait=False)        # g++ job of the CPU backend runs beside the hipcc jobs
    if not force and not needs_build():
        _finish_cpu(cpu_job)
        return LIB_PATH
    try:
        os.makedirs(BUILD_DIR, exist_ok=True)
        hdrs = []
        for d, _, fs in os.walk(CSRC):
            if os.path.abspath(d).startswith(os.path.abspath(BUILD_DIR)):
                continue
            hdrs += [os.path.join(d, f) for f in fs if f.endswith((".hpp", ".h"))]
        hdrs.append(os.path.join(_HERE, "..", "include", "mi_engine.h"))
        procs, objs = [], []
        for name in SOURCES:
            src = os.path.join(CSRC, name)
            obj = os.path.join(BUILD_DIR, name.replace(".hip", ".o"))
            objs.append(obj)
            # stale against the headers THIS translation unit reaches through #include "..." (a generated model header touches only the units
            # of that robot: a Shadow-Hand edit no longer recompiles the Ant's 90-second unit)
            if not force and not _obj_stale(src, obj, _newest_include(src)):
                continue
            cmd = [hipcc_path()] + HIPCC_FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            log = open(obj.replace(".o", ".log"), "w")
            procs.append((name, subprocess.Popen(cmd, cwd=CSRC, stdout=log, stderr=subprocess.STDOUT), log))
        failed = []
        for name, p, log in procs:
            rc = p.wait()
            log.close()
            if rc != 0:
                failed.append(name)
        if failed:
            for name in failed:
                with open(os.path.join(BUILD_DIR, name.replace(".hip", ".log"))) as f:
                    print(f.read()[-4000:])
            raise RuntimeError(f"hipcc failed for {failed}")
        # the resource check runs on the object logs BEFORE anything is linked: a library that fails it never exists on disk,
        # so a later needs_build() / lib() cannot silently pick it up
        usage = resource_usage()
        with open(os.path.join(BUILD_DIR, "resource_usage.txt"), "w") as f:
            for k, u in usage.items():
                f.write(f"{k}: {u}\n")
        bad = over_sgpr_budget(usage)
        if bad:
            if os.path.exists(LIB_PATH):
                os.remove(LIB_PATH)
            raise RuntimeError(f"step kernels spill SGPRs (known-bad regime on gfx950): {bad}")
        tmp = LIB_PATH + ".tmp"
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, LIB_PATH)
    except BaseException:
        # the g++ job of the CPU backend must not be left running / half written, and its own failure must not be hidden
        try:
            _finish_cpu(cpu_job)
        except Exception as e:      # noqa: BLE001 -- report it beside the hipcc error
            print(f"(CPU backend build also failed: {e})")
        raise
    _finish_cpu(cpu_job)
    return LIB_PATH


def resource_usage(build_dir=None, sources=None):
    """Parse the -Rpass-analysis=kernel-resource-usage remarks of the last build: {kernel: {field: value}}."""
    import re
    out = {}
    for name in (sources if sources is not None else SOURCES):
        logp = os.path.join(build_dir or BUILD_DIR, name.replace(".hip", ".log"))
        if not os.path.exists(logp):
            continue
        cur = None
        for line in open(logp):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+([A-Za-z /\[\]]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    return out


def _bind_lifecycle(L):
    """ctypes signatures of the engine lifecycle entry points (the part of include/mi_engine.h both libraries export)."""
    L.mi_abi_version.restype = C.c_int
    if L.mi_abi_version() != MI_ABI_VERSION:
        raise RuntimeError(f"{getattr(L, '_name', 'library')}: ABI version {L.mi_abi_version()}, this package speaks {MI_ABI_VERSION} -- rebuild "
                           f"(python -c 'import __graft_entry__ as g; g.build()'; a cached isaacgymenvs_amd/_variants/ library is rebuilt on demand)")
    L.mi_last_error.restype = C.c_char_p
    L.mi_engine_arena_bytes.restype = C.c_size_t
    L.mi_engine_arena_bytes.argtypes = [C.c_char_p, C.c_int]
    L.mi_task_info.argtypes = [C.c_char_p, C.POINTER(MiTaskInfo)]
    L.mi_engine_create.argtypes = [C.c_char_p, C.POINTER(MiSimParams), C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                   C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.mi_engine_init_state.argtypes = [C.c_void_p, C.c_void_p]
    L.mi_engine_destroy.argtypes = [C.c_void_p]
    L.mi_engine_destroy.restype = None
    L.mi_engine_num_tensors.argtypes = [C.c_void_p]
    L.mi_engine_tensor_desc.argtypes = [C.c_void_p, C.c_int, C.POINTER(MiTensorDesc)]
    L.mi_engine_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mi_engine_reset_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mi_engine_simulate.argtypes = [C.c_void_p, C.c_void_p]
    L.mi_engine_refresh_rigid_body_states.argtypes = [C.c_void_p, C.c_void_p]
    L.mi_engine_compute_jacobians.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mi_engine_compute_mass_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mi_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.mi_engine_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
    L.mi_engine_set_noise.argtypes = [C.c_void_p, C.c_int, C.POINTER(MiNoiseParams)]
    L.mi_engine_last_ring.argtypes = [C.c_void_p]
    L.mi_engine_set_terrain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                        C.c_int, C.c_int, C.c_float, C.c_int]
    return L


_lib_cpu = None


def cpu_needs_build():
    """Same staleness rule as the HIP library: any source or header under csrc/ (the CPU backend includes the engine core, the task
    headers and the arena layout the kernels use) or the public header newer than the library."""
    if not os.path.exists(CPU_LIB_PATH):
        return True
    t = os.path.getmtime(CPU_LIB_PATH)
    for d, _, fs in os.walk(CSRC):
        if os.path.abspath(d).startswith(os.path.abspath(BUILD_DIR)):
            continue
        for f in fs:
            if f.endswith((".hpp", ".h", ".cpp")) and os.path.getmtime(os.path.join(d, f)) > t:
                return True
    return os.path.getmtime(os.path.join(_HERE, "..", "include", "mi_engine.h")) > t


# translation units of the CPU backend: (source, object suffix, optimisation level, defines).  The hand + object sub-step is compiled once per
# hand and object shape (about a minute each at -O1; they run side by side and beside the hipcc jobs)
# last column: the robot models whose constants the unit instantiates (None: all -- it holds the arena layout); a run-time variant of a model
# (assets/runtime.py) recompiles only those
_HAND_MODEL = {0: "shadow_hand", 1: "allegro_hand"}
CPU_UNITS = [("mi_engine_cpu.cpp", "", "-O2", [], ("cartpole", "ant", "humanoid", "quadcopter", "ingenuity", "balance_bot")),
             ("cpu_anymal.cpp", "", "-O2", [], ("anymal",)), ("cpu_articulation.cpp", "", "-O2", [], ("articulation",))] + \
            [("cpu_hand.cpp", f"_{h}", "-O2", [f"-DMI_CPU_HAND={h}"], (_HAND_MODEL[h],)) for h in (0, 1)] + \
            [("cpu_hand_physics.cpp", f"_{h}_{sh}", "-O1", [f"-DMI_CPU_HAND={h}", f"-DMI_CPU_SHAPE={sh}"], (_HAND_MODEL[h],)) for h in (0, 1) for sh in (0, 1, 2)]
# -fno-gnu-unique: the generated models' constexpr tables that are indexed at run time (ModelArticulation::dof_lower, sph_rad, ...) would otherwise be
# STB_GNU_UNIQUE symbols, which the dynamic loader binds ACROSS libraries even under RTLD_LOCAL -- two run-time variants of the Articulation robot loaded
# into one process (a Franka, then a Kuka) then read each other's tables (found in round 5: NaN joint limits in the second robot's reset)
CPU_FLAGS = ["-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-gnu-unique"]


def cpu_object(src, suffix):
    return os.path.join(BUILD_DIR, "cpu_" + src.replace(".cpp", "") + suffix + ".o")


class _CpuJob:
    """the g++ processes of one CPU-backend build; finish() waits for them and links"""

    def __init__(self, procs, objs):
        self.procs, self.objs = procs, objs

    def finish(self):
        failed = []
        for name, p, log in self.procs:
            rc = p.wait()
            log.close()
            if rc != 0:
                failed.append((name, log.name))
        if failed:
            for name, logp in failed:
                with open(logp) as f:
                    print(f.read()[-4000:])
            raise RuntimeError(f"g++ failed for the CPU backend: {[n for n, _ in failed]}")
        tmp = CPU_LIB_PATH + ".tmp"
        subprocess.check_call(["g++", "-shared", "-fPIC", "-fopenmp"] + self.objs + ["-o", tmp])
        os.replace(tmp, CPU_LIB_PATH)


def build_cpu(force=False, wait=True):
    """g++ build of the CPU backend (csrc/cpu/: the engine core, the task headers and the arena layout the kernels use, on the host, OpenMP
    over envs): one object per translation unit, compiled in parallel, linked into libmi_engine_cpu.so.
    Returns the job when wait=False (build() overlaps it with the hipcc jobs)."""
    from .registry import generate_headers
    generate_headers()
    if not force and not cpu_needs_build():
        return None
    os.makedirs(BUILD_DIR, exist_ok=True)
    hdrs = []
    for d, _, fs in os.walk(CSRC):
        if os.path.abspath(d).startswith(os.path.abspath(BUILD_DIR)):
            continue
        hdrs += [os.path.join(d, f) for f in fs if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(_HERE, "..", "include", "mi_engine.h"))
    newest = max(os.path.getmtime(h) for h in hdrs)
    procs, objs = [], []
    for src, suffix, opt, defs, _models in CPU_UNITS:
        srcp = os.path.join(CSRC, "cpu", src)
        obj = cpu_object(src, suffix)
        objs.append(obj)
        if not force and not _obj_stale(srcp, obj, _newest_include(srcp)):
            continue
        cmd = ["g++", opt] + CPU_FLAGS + defs + ["-c", srcp, "-o", obj]
        log = open(obj.replace(".o", ".log"), "w")
        procs.append((os.path.basename(obj), subprocess.Popen(cmd, cwd=CSRC, stdout=log, stderr=subprocess.STDOUT), log))
    job = _CpuJob(procs, objs)
    if not wait:
        return job
    job.finish()
    return None


def _finish_cpu(job):
    if job is not None:
        job.finish()


def lib_cpu():
    """Load the CPU backend.  Fails loudly when it has not been built (there is no fallback onto anything else)."""
    global _lib_cpu
    if _lib_cpu is not None:
        return _lib_cpu
    if not os.path.exists(CPU_LIB_PATH):
        raise RuntimeError(f"{CPU_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    _lib_cpu = _bind_lifecycle(C.CDLL(CPU_LIB_PATH))
    return _lib_cpu


_lib = None
_variant_libs = {}


def variant_lib(path):
    """a library built for a run-time asset (assets/runtime.py): same ABI, another robot model compiled in"""
    if path not in _variant_libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing")
        _variant_libs[path] = _bind_lifecycle(C.CDLL(path))
    return _variant_libs[path]


def lib():
    """Load the HIP library.  Fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    _bind_lifecycle(L)
    L.mi_compute_locomotion_observations.argtypes = [C.c_char_p, C.c_int, C.POINTER(MiLocoParams)] + [C.c_void_p] * 18
    L.mi_compute_locomotion_reward.argtypes = [C.c_char_p, C.c_int, C.POINTER(MiLocoParams)] + [C.c_void_p] * 9
    L.mi_compute_cartpole_reward.argtypes = [C.c_int, C.POINTER(MiCartpoleParams)] + [C.c_void_p] * 9
    L.mi_compute_quadcopter_reward.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3
    L.mi_compute_anymal_observations.argtypes = [C.c_int, C.POINTER(MiAnymalFlatParams)] + [C.c_void_p] * 7
    L.mi_compute_anymal_reward.argtypes = [C.c_int, C.POINTER(MiAnymalFlatParams)] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4
    L.mi_compute_hand_reward.argtypes = [C.c_int, C.POINTER(MiHandRewardParams)] + [C.c_void_p] * 11 + [C.c_int, C.c_void_p, C.c_void_p]
    L.mi_compute_hand_full_state.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float] + [C.c_void_p] * 11 + [C.c_int, C.c_void_p]
    L.mi_randomize_rotation.argtypes = [C.c_int] + [C.c_void_p] * 6
    V, F, I = C.c_void_p, C.c_float, C.c_int
    L.mi_compute_bbot_reward.argtypes = [I, V, V, V, F, V, V, F, V, V, V]
    L.mi_compute_ingenuity_reward.argtypes = [I] + [V] * 7 + [F, V, V, V]
    L.mi_compute_franka_cabinet_reward.argtypes = [I, C.POINTER(MiFrankaCabinetRewardParams), V, V, V, I, V, I] + [V] * 13
    L.mi_compute_grasp_transforms.argtypes = [I] + [V] * 13
    L.mi_axisangle2quat.argtypes = [I, V, F, V, V]
    L.mi_compute_franka_cube_stack_reward.argtypes = [I, C.POINTER(MiFrankaCubeStackRewardParams)] + [V] * 12
    L.mi_randomize_rotation_pen.argtypes = [I, V, V, F, V, V, V, V, V]
    L.mi_lgsk_kernel.argtypes = [I, V, F, F, V, V]
    L.mi_gen_keypoints.argtypes = [I, V, I, C.POINTER(C.c_float), V, V]
    L.mi_compute_trifinger_reward.argtypes = [I, C.POINTER(MiTrifingerRewardParams)] + [V] * 11
    L.mi_compute_trifinger_observations_states.argtypes = [I, I, I, I, I, I] + [V] * 11
    L.mi_trifinger_random_xy.argtypes = [I, V, F, V, V]
    L.mi_trifinger_random_z.argtypes = [I, V, F, F, V, V]
    L.mi_trifinger_default_orientation.argtypes = [I, V, V]
    L.mi_trifinger_random_orientation.argtypes = [I, V, V, V]
    L.mi_trifinger_random_orientation_within_angle.argtypes = [I, V, V, F, V, V]
    L.mi_trifinger_random_angular_vel.argtypes = [I, V, F, V, V]
    L.mi_trifinger_random_yaw_orientation.argtypes = [I, V, V, V]
    L.mi_amp_dof_to_obs.argtypes = [I, V, V, V]
    L.mi_compute_humanoid_amp_observations.argtypes = [I, V, V, V, V, I, I, V, V]
    L.mi_compute_humanoid_amp_reward.argtypes = [I, V, V, V]
    L.mi_compute_humanoid_amp_reset.argtypes = [I, V, V, V, C.POINTER(C.c_int64), I, V, I, F, I, F, V, V, V]
    L.mi_compute_hand_reward_dextreme.argtypes = [I, C.POINTER(MiDextremeRewardParams)] + [V] * 8 + [I] + [V] * 7 + [I, V, V, V]
    _lib = L
    return L


def check(rc, L=None):
    if rc != 0:
        raise RuntimeError("mi_engine: " + ((L or lib()).mi_last_error() or b"?").decode())


def task_info(task):
    """Static facts of a task (observation / action widths ...).  Answered by whichever library is present: the table is shared
    (csrc/arena_layout.hpp), so a CPU-only install can still size its buffers."""
    info = MiTaskInfo()
    L = lib() if os.path.exists(LIB_PATH) else lib_cpu()
    check(L.mi_task_info(task.encode(), C.byref(info)), L)
    return info


_DT = None


def _dtypes():
    global _DT
    if _DT is None:
        import torch
        _DT = {0: torch.float32, 1: torch.int64, 2: torch.uint8, 3: torch.int32}
    return _DT


class Engine:
    """Owns a torch uint8 arena on `device` and the native engine handle bound to it."""

    def __init__(self, task, sim_params: MiSimParams, task_params, num_envs, device, seed=0, env_id_offset=0, terrain=None, lib_path=None):
        import torch
        dev = torch.device(device)
        if dev.type == "cpu":
            # the reference's CPU pipeline (sim_device=cpu): the engine's own host build, OpenMP over envs -- not the test oracle
            if task not in CPU_TASKS:
                raise RuntimeError(f"unknown task {task} for the CPU backend ({', '.join(CPU_TASKS)})")
            L = lib_cpu() if lib_path is None else variant_lib(lib_path)
        elif dev.type == "cuda":
            if not torch.cuda.is_available():
                raise RuntimeError("no ROCm device visible to PyTorch")
            L = lib() if lib_path is None else variant_lib(lib_path)
        else:
            raise RuntimeError(f"unsupported sim device {device!r}")
        self.L = L
        self._raw_stream = None
        self.task, self.N, self.device = task, num_envs, dev
        nbytes = L.mi_engine_arena_bytes(task.encode(), num_envs)
        if nbytes == 0:
            check(-1, L)
        self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        if os.environ.get("MI_ARENA_POISON"):
            # tests only: an arena full of NaN bit patterns instead of zeros -- whatever mi_engine_init_state does not write shows
            # (tests/test_gpu_step_time_sanity.py::test_init_state_writes_everything_the_step_reads)
            self.arena.view(torch.int32)[:] = 0x7FC00000
        self._sim, self._tp = sim_params, task_params
        h = C.c_void_p()
        check(L.mi_engine_create(task.encode(), C.byref(sim_params), C.cast(C.byref(task_params), C.c_void_p),
                                 C.sizeof(task_params), num_envs, env_id_offset, seed & 0xFFFFFFFFFFFFFFFF,
                                 self.arena.data_ptr(), nbytes, C.byref(h)), L)
        self.h = h
        self.tensors = {}
        dts = _dtypes()
        for i in range(L.mi_engine_num_tensors(h)):
            d = MiTensorDesc()
            check(L.mi_engine_tensor_desc(h, i, C.byref(d)), L)
            dt = dts[d.dtype]
            esz = torch.empty(0, dtype=dt).element_size()
            shape = [d.shape[k] for k in range(d.ndim)]
            stride = [d.stride[k] for k in range(d.ndim)]
            extent = 1 + sum((s - 1) * st for s, st in zip(shape, stride))
            flat = self.arena[d.byte_offset:d.byte_offset + extent * esz].view(dt)
            self.tensors[d.name.decode()] = torch.as_strided(flat, shape, stride)
        if terrain is not None:
            # terrain: object with heightsamples [rows, cols] int16, env_origins [levels, types, 3], horizontal_scale,
            # vertical_scale, border_size, env_length (isaacgymenvs_amd/tasks/terrain.py) + max_init_level
            self.height_samples = torch.as_tensor(terrain.heightsamples, dtype=torch.int16).contiguous().to(dev)
            self.terrain_origins = torch.as_tensor(terrain.env_origins, dtype=torch.float32).contiguous().to(dev)
            check(L.mi_engine_set_terrain(h, self.height_samples.data_ptr(), self.height_samples.shape[0],
                                          self.height_samples.shape[1], float(terrain.horizontal_scale),
                                          float(terrain.vertical_scale), float(terrain.border_size),
                                          self.terrain_origins.data_ptr(), self.terrain_origins.shape[0],
                                          self.terrain_origins.shape[1], float(terrain.env_length),
                                          int(getattr(terrain, "max_init_level", 0))))
            if float(getattr(terrain, "slope_threshold", 0.0) or 0.0) > 0.0:
                check(L.mi_engine_set_option(h, b"terrain_slope_threshold", float(terrain.slope_threshold)), L)
        if dev.type == "cuda":
            with torch.cuda.device(dev):
                check(L.mi_engine_init_state(h, self._stream()), L)
        else:
            check(L.mi_engine_init_state(h, None), L)

    def _stream(self):
        """the caller's current HIP stream on the engine's device, as a raw pointer (every launch goes there)"""
        if self.device.type != "cuda":
            return None
        raw = self._raw_stream
        if raw is None:
            import torch
            # (a C call that returns the pointer: ~0.2 us against ~3 us for torch.cuda.current_stream(...).cuda_stream -- the host side of a
            #  step is ~25 us against 37 us of GPU time for Ant@4096, tools/debug/host_enqueue_time.py)
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            if raw is None:
                raw = lambda idx: torch.cuda.current_stream(idx).cuda_stream      # noqa: E731
            self._raw_stream = raw
            self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        return raw(self._dev_index)

    def step(self, actions):
        if self.L.mi_engine_step(self.h, actions.data_ptr(), self._stream()) != 0:
            check(-1, self.L)

    def simulate(self):
        check(self.L.mi_engine_simulate(self.h, self._stream()), self.L)

    def refresh_rigid_body_states(self):
        """gym.refresh_rigid_body_state_tensor: fills tensors["rigid_body_state"] [N, num_bodies, 13] from the current root / dof state"""
        check(self.L.mi_engine_refresh_rigid_body_states(self.h, self._stream()), self.L)

    def _nv(self):
        info = MiTaskInfo()          # from the library the engine lives in: a run-time variant carries its own robot's table
        check(self.L.mi_task_info(self.task.encode(), C.byref(info)), self.L)
        return info.num_bodies, info.num_dofs + (0 if info.fixed_base else 6)

    def compute_jacobians(self, out=None):
        """gym.refresh_jacobian_tensors: [N, num_bodies, 6, nv] (rows: linear, angular velocity of the body origin, world frame; nv = 6 base +
        dofs for a floating base) -- include/mi_engine.h mi_engine_compute_jacobians.  `out`: a contiguous fp32 tensor to fill."""
        import torch
        nb, nv = self._nv()
        if out is None:
            out = torch.empty((self.N, nb, 6, nv), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == self.N * nb * 6 * nv
        check(self.L.mi_engine_compute_jacobians(self.h, out.data_ptr(), self._stream()), self.L)
        return out

    def compute_mass_matrices(self, out=None):
        """gym.refresh_mass_matrix_tensors: [N, nv, nv] joint-space inertia (armatures on the diagonal)"""
        import torch
        _, nv = self._nv()
        if out is None:
            out = torch.empty((self.N, nv, nv), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == self.N * nv * nv
        check(self.L.mi_engine_compute_mass_matrices(self.h, out.data_ptr(), self._stream()), self.L)
        return out

    def reset_idx(self, env_ids):
        if env_ids.numel():
            check(self.L.mi_engine_reset_idx(self.h, env_ids.data_ptr(), env_ids.numel(), self._stream()), self.L)

    def set_option(self, key, value):
        check(self.L.mi_engine_set_option(self.h, key.encode(), float(value)), self.L)

    def set_noise(self, which, dist="off", op="additive", a=0.0, b=0.0, a_corr=0.0, b_corr=0.0, epoch=0):
        """In-kernel observation (which=0) / action (which=1) noise; see MiNoiseParams in include/mi_engine.h."""
        p = MiNoiseParams(dist={"off": 0, "gaussian": 1, "uniform": 2}[dist], op={"additive": 0, "scaling": 1}[op], a=a, b=b,
                          a_corr=a_corr, b_corr=b_corr, epoch=epoch)
        check(self.L.mi_engine_set_noise(self.h, int(which), C.byref(p)), self.L)

    def get_option(self, key):
        out = C.c_double()
        check(self.L.mi_engine_get_option(self.h, key.encode(), C.byref(out)), self.L)
        return out.value

    def last_ring(self):
        return self.L.mi_engine_last_ring(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

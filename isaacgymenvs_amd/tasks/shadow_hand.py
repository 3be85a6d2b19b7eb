"""ShadowHand: 24-DoF Shadow hand (20 actuated, 4 tendon-coupled) re-orienting a cube in hand
(reference isaacgymenvs/tasks/shadow_hand.py).

Host side only: config -> MiHandParams and the reference's attribute names as views of the engine arena.  Supported
subset: objectType "block" (the cube: an isotropic free body), "egg" (ellipsoid) and "pen" (capsule) with principal inertias, all four observation layouts (full_state 211, full 157,
full_no_vel 77, openai 42), asymmetric observations (states_buf = full state), absolute or relative position control,
random object forces (forceScale > 0), in-kernel resets.  Physics simplifications are listed
in DESIGN.md (hand geometry sampled by spheres against the exact box, no hand self-collision, soft tendons, drive force
limits not clamped).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import native
from ..registry import load_extras, load_model
from .base.vec_task import VecTask

CUBE_SIZE = 0.05          # assets/urdf/objects/cube_multicolor.urdf: box 0.05
CUBE_DENSITY = 567.0


OBJECT_SHAPE_ID = {"block": 0, "pen": 1, "egg": 2}   # include/mi_engine.h MiHandParams.object_shape (1 is refused by the engine)
EGG_SEMI_AXES = (0.03, 0.03, 0.04)
EGG_DENSITY = 1000.0
PEN_RADIUS, PEN_HALF_LENGTH, PEN_DENSITY = 0.008, 0.1, 1000.0     # mjcf/open_ai_assets/hand/pen.xml:20 (capsule "0.008 0.1"), default density


def capsule_mass_inertia(r, hl, density):
    """mass and principal inertias (x = y transverse, z along the axis) of a solid capsule: cylinder 2 hl long + two hemispheres."""
    mc = density * np.pi * r * r * 2.0 * hl
    ms = density * 4.0 / 3.0 * np.pi * r ** 3
    izz = 0.5 * mc * r * r + 0.4 * ms * r * r
    ixx = mc * (3.0 * r * r + (2.0 * hl) ** 2) / 12.0 + ms * (0.4 * r * r + hl * hl + 0.75 * hl * r)
    return mc + ms, (ixx, ixx, izz)


NUM_OBS = {"openai": 42, "full_no_vel": 77, "full": 157, "full_state": 211}     # shadow_hand.py:101-106
OBS_TYPE_ID = {"full_state": 0, "openai": 1, "full_no_vel": 2, "full": 3}


def obs_columns(obs_type):
    """Columns of compute_full_state's 211-vector (shadow_hand.py:528-584) that make up the other observation layouts
    (compute_fingertip_observations(True) :472-485, compute_full_observations :498-526).  Full-state layout: dof pos 0:24,
    dof vel 24:48, dof force 48:72, object pose 72:79, linvel 79:82, angvel 82:85, goal pose 85:92, rel. rotation 92:96,
    fingertip states 96:161 (5 x 13), fingertip force-torques 161:191, actions 191:211."""
    r = lambda a, b: list(range(a, b))
    tip_pos = [96 + 13 * t + k for t in range(5) for k in range(3)]
    if obs_type == "openai":
        return tip_pos + r(72, 75) + r(92, 96) + r(191, 211)
    if obs_type == "full_no_vel":
        return r(0, 24) + r(72, 79) + r(85, 92) + r(92, 96) + tip_pos + r(191, 211)
    if obs_type == "full":
        return r(0, 24) + r(24, 48) + r(72, 79) + r(79, 82) + r(82, 85) + r(85, 92) + r(92, 96) + r(96, 161) + r(191, 211)
    if obs_type == "full_state":
        return r(0, 211)
    raise ValueError("Unknown type of observations!\nobservationType should be one of: [openai, full_no_vel, full, full_state]")


def hand_max_episode_length(cfg):
    """episodeLength, or resetTime seconds worth of control steps when resetTime > 0 (shadow_hand.py:81,139-141)."""
    env = cfg["env"]
    reset_time = float(env.get("resetTime", -1.0))
    if reset_time > 0.0:
        return int(round(reset_time / (int(env.get("controlFrequencyInv", 1)) * float(cfg["sim"]["dt"]))))
    return env["episodeLength"]


def hand_params_from_cfg(cfg, model="shadow_hand", hand_quat=None, object_offset=(0.0, -0.39, 0.10), pen_offset_z=0.02,
                         cube=(CUBE_SIZE, CUBE_DENSITY), obs_layout=None):
    """cfg -> MiHandParams.  The keyword arguments are what differs for the AllegroHand task (tasks/allegro_hand.py): the model with its
    extras, the actor's orientation, where the object starts relative to the hand, the cube asset, the observation layouts."""
    env = cfg["env"]
    ex = load_extras(model)
    cube_size, cube_density = cube
    num_obs, obs_type_id, columns = obs_layout or (NUM_OBS, OBS_TYPE_ID, obs_columns)
    p = native.MiHandParams()
    r = p.rew
    r.max_episode_length = float(hand_max_episode_length(cfg))
    r.dist_reward_scale = float(env["distRewardScale"]); r.rot_reward_scale = float(env["rotRewardScale"])
    r.rot_eps = float(env["rotEps"]); r.action_penalty_scale = float(env["actionPenaltyScale"])
    r.success_tolerance = float(env["successTolerance"]); r.reach_goal_bonus = float(env["reachGoalBonus"])
    r.fall_dist = float(env["fallDistance"]); r.fall_penalty = float(env["fallPenalty"])
    r.max_consecutive_successes = int(env["maxConsecutiveSuccesses"])
    r.av_factor = float(env.get("averFactor", 0.1))
    r.ignore_z_rot = 0                                            # object_type == "pen" (shadow_hand.py:421)
    p.vel_obs_scale, p.force_torque_obs_scale = 0.2, 10.0         # shadow_hand.py:61-62
    p.reset_position_noise = float(env["resetPositionNoise"])
    p.reset_dof_pos_noise = float(env["resetDofPosRandomInterval"])
    p.reset_dof_vel_noise = float(env["resetDofVelRandomInterval"])
    p.act_moving_average = float(env["actionsMovingAverage"])
    p.dof_speed_scale = float(env["dofSpeedScale"])
    p.dt = float(cfg["sim"]["dt"])
    p.use_relative_control = int(bool(env["useRelativeControl"]))
    ca = env.get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    hand_pos = (0.0, 0.0, 0.5)                                    # get_axis_params(0.5, up_axis_idx), :306-307
    obj = tuple(hand_pos[k] + object_offset[k] for k in range(3))   # shadow_hand.py:309-315 (0, -0.39, 0.10); allegro_hand.py:287-294
    for k in range(3):
        p.hand_pos[k] = hand_pos[k]
        p.object_init_pos[k] = obj[k]
        p.goal_init_pos[k] = obj[k] - (0.04 if k == 2 else 0.0)   # goal_states = object_init_state, z -= 0.04 (:393-395)
    for k in range(4):
        p.hand_quat[k] = float((hand_quat if hand_quat is not None else ex["mount_quat"])[k])
    p.cube_half = cube_size / 2
    p.cube_mass = cube_density * cube_size ** 3
    p.cube_inertia = p.cube_mass * cube_size ** 2 / 6.0
    p.mu = 1.0
    p.object_shape = OBJECT_SHAPE_ID[env.get("objectType", "block")]
    if env.get("objectType", "block") == "egg":
        # mjcf/open_ai_assets/hand/egg.xml:10: ellipsoid, semi-axes 0.03 0.03 0.04, no mass given -> AssetOptions.density default 1000 kg/m^3
        a, b, c = EGG_SEMI_AXES
        m = EGG_DENSITY * 4.0 / 3.0 * np.pi * a * b * c
        p.cube_mass = m                                           # the object's mass, whatever its shape
        p.cube_half = max(EGG_SEMI_AXES)
        inertia = (m / 5.0 * (b * b + c * c), m / 5.0 * (a * a + c * c), m / 5.0 * (a * a + b * b))
        p.cube_inertia = sum(inertia) / 3.0
        for k in range(3):
            p.object_dims[k] = EGG_SEMI_AXES[k]
            p.object_inertia[k] = inertia[k]
    if env.get("objectType", "block") == "pen":
        m, inertia = capsule_mass_inertia(PEN_RADIUS, PEN_HALF_LENGTH, PEN_DENSITY)
        p.cube_mass = m
        p.cube_half = PEN_HALF_LENGTH
        p.cube_inertia = sum(inertia) / 3.0
        p.object_dims[0], p.object_dims[1], p.object_dims[2] = PEN_RADIUS, PEN_HALF_LENGTH, 0.0
        for k in range(3):
            p.object_inertia[k] = inertia[k]
        r.ignore_z_rot = 1                                        # shadow_hand.py:421
        obj = (hand_pos[0] + object_offset[0], hand_pos[1] + object_offset[1], hand_pos[2] + pen_offset_z)   # the pen starts lower (:316-317)
        for k in range(3):
            p.object_init_pos[k] = obj[k]
            p.goal_init_pos[k] = obj[k] - (0.04 if k == 2 else 0.0)
    for a, d in enumerate(ex["actuated_dofs"]):
        p.actuated[a] = int(d)
    ot = env["observationType"]
    cols = columns(ot)
    assert len(cols) == num_obs[ot]
    p.obs_type, p.num_obs = obs_type_id[ot], len(cols)
    p.asymmetric_obs = int(bool(env.get("asymmetric_observations", False)))
    if ot != "full_state":
        for k, c in enumerate(cols):
            p.obs_map[k] = c
    p.force_scale = float(env.get("forceScale", 0.0))                          # :69-72
    fr = env.get("forceProbRange", [0.001, 0.1])
    p.force_prob_range[0], p.force_prob_range[1] = float(fr[0]), float(fr[1])
    p.force_decay = float(env.get("forceDecay", 0.99))
    p.force_decay_interval = float(env.get("forceDecayInterval", 0.08))
    return p


class ShadowHand(VecTask):
    native_task = "ShadowHand"
    model_name = "shadow_hand"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        env = cfg["env"]
        if env["objectType"] not in ("block", "egg", "pen"):                        # shadow_hand.py:86-87
            raise AssertionError("objectType must be one of block, egg, pen")
        if env["observationType"] not in NUM_OBS:                                   # shadow_hand.py:97-99
            raise Exception("Unknown type of observations!\nobservationType should be one of: [openai, full_no_vel, full, full_state]")
        self.randomize = cfg["task"]["randomize"]
        self.randomization_params = cfg["task"].get("randomization_params", {})
        self.reset_time = env.get("resetTime", -1.0)
        self.max_episode_length = hand_max_episode_length(cfg)
        self.obs_type = env["observationType"]
        self.object_type = env["objectType"]
        self.asymmetric_obs = bool(env.get("asymmetric_observations", False))
        self.force_scale = env.get("forceScale", 0.0)
        self.num_obs_dict = dict(NUM_OBS)
        self.fingertips = ["robot0:ffdistal", "robot0:mfdistal", "robot0:rfdistal", "robot0:lfdistal", "robot0:thdistal"]
        self.num_fingertips = 5
        cfg["env"]["numObservations"] = self.num_obs_dict[self.obs_type]           # :113-116
        cfg["env"]["numStates"] = 211 if self.asymmetric_obs else 0
        cfg["env"]["numActions"] = 20
        cfg["env"].setdefault("plane", {"staticFriction": 1.0})
        self.spec = load_model("shadow_hand")
        self.num_shadow_hand_dofs = self.spec.nd
        self.num_shadow_hand_bodies = self.spec.nb
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        t = self.engine.tensors
        dev = self.device
        ex = load_extras("shadow_hand")
        self.dof_state = t["dof_state"]
        self.shadow_hand_dof_pos, self.shadow_hand_dof_vel = self.dof_state[..., 0], self.dof_state[..., 1]
        self.dof_force_tensor = t["dof_force"]
        self.vec_sensor_tensor = torch.as_strided(t["force_sensor"], (self.num_envs, 30), (1, self.num_envs))
        self.object_state = t["object_state"]
        self.object_pose, self.object_pos, self.object_rot = self.object_state[:, 0:7], self.object_state[:, 0:3], self.object_state[:, 3:7]
        self.object_linvel, self.object_angvel = self.object_state[:, 7:10], self.object_state[:, 10:13]
        self.goal_states = t["goal_states"]
        self.goal_pose, self.goal_pos, self.goal_rot = self.goal_states, self.goal_states[:, 0:3], self.goal_states[:, 3:7]
        self.fingertip_state = t["fingertip_state"]
        self.fingertip_pos = self.fingertip_state[:, :, 0:3]
        self.cur_targets, self.prev_targets = t["cur_targets"], t["prev_targets"]
        self.actions = t["actions"]
        self.successes, self.consecutive_successes = t["successes"], t["consecutive_successes"]
        self.reset_goal_buf = t["reset_goal_buf"]
        if self.asymmetric_obs:
            self.states_buf = t["states_buf"]                                         # compute_full_state(True), :584
        self.rb_forces_object, self.random_force_prob = t["rb_forces_object"], t["random_force_prob"]
        lo = np.minimum(self.spec.dof_lower, self.spec.dof_upper); up = np.maximum(self.spec.dof_lower, self.spec.dof_upper)
        self.shadow_hand_dof_lower_limits = torch.tensor(lo, dtype=torch.float32, device=dev)
        self.shadow_hand_dof_upper_limits = torch.tensor(up, dtype=torch.float32, device=dev)
        self.actuated_dof_indices = torch.tensor(ex["actuated_dofs"], dtype=torch.long, device=dev)
        self.fingertip_handles = torch.tensor([self.spec.body_names.index(n) for n in self.fingertips], dtype=torch.long, device=dev)
        self.extras["consecutive_successes"] = self.consecutive_successes[0]       # shadow_hand.py:424 (.mean() of a 1-vector)

    def _viewer_extras(self, env):
        """the manipulated object (drawn by its bounding ball) and, translucent in spirit, the goal pose beside it"""
        obj = self.object_pos[env].detach().cpu().numpy().astype(np.float64)
        goal = self.goal_pos[env].detach().cpu().numpy().astype(np.float64)
        r = 0.035
        return np.stack([obj, goal]), np.array([r, r]), np.array([[0.9, 0.75, 0.2], [0.55, 0.9, 0.55]])

    def _task_params(self):
        return hand_params_from_cfg(self.cfg)

    #: `actor_params` entries of the two actors -> columns of the `actor_scale` tensor (csrc/core/hand_engine.hpp HS_*; reference
    #: cfg/task/ShadowHand.yaml:89-161).  dof_properties.stiffness is the position drives' kp (DOF_MODE_POS).
    HAND_SCALE_COLUMNS = {("hand", "rigid_body_properties", "mass"): 0, ("hand", "dof_properties", "damping"): 1,
                          ("hand", "dof_properties", "stiffness"): 2, ("hand", "tendon_properties", "stiffness"): 3,
                          ("hand", "tendon_properties", "damping"): 4, ("object", "rigid_body_properties", "mass"): 5,
                          ("object", "scale", "scale"): 6}

    def _actor_scale_column(self, actor, group, attr):
        if (actor, group, attr) == ("hand", "rigid_body_properties", "mass") and "hand_body_mass_scale" in self.engine.tensors:
            # one draw per env and BODY, as the reference samples the hand's rigid-body property list (vec_task.py:783-828; ShadowHand.yaml:104-110):
            # the `hand_body_mass_scale` tensor, read by the Sim<Scaled<M>> kernels once option "hand_body_mass" is on.  (Until round 5: one
            # factor per env, column 0 of `actor_scale`; that column stays 1.)
            return ("hand_body_mass_scale", 0, np.asarray(self.spec.mass, np.float64))
        return self.HAND_SCALE_COLUMNS.get((actor, group, attr))

    def _on_body_tensor_written(self):
        if not getattr(self, "_hand_body_mass_on", False):
            self.engine.set_option("hand_body_mass", 1)
            self._hand_body_mass_on = True

    def _actor_reference_value(self, group, attr, actor=None):
        """what an `additive` draw is relative to (a `scaling` draw is the factor itself)"""
        ex = load_extras("shadow_hand")
        if group == "scale":
            return 1.0
        if group == "tendon_properties":
            return float(ex["tendon_limit_stiffness"] if attr == "stiffness" else ex["tendon_damping"])
        if actor == "object":
            return float(self._task_params_struct.cube_mass)
        if (group, attr) == ("dof_properties", "stiffness"):
            return float(np.mean(ex["dof_kp"]))
        return super()._actor_reference_value(group, attr, actor)

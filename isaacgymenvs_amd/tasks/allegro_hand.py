"""AllegroHand: the 16-DoF Allegro hand re-orienting a cube in hand (reference isaacgymenvs/tasks/allegro_hand.py).

The reference task is the ShadowHand task line by line (same reset_idx / reset_target_pose / pre_physics_step / compute_hand_reward) with
another robot: 16 dofs, all of them position-driven with the gains the task writes into the dof properties (allegro_hand.py:256-264:
stiffness 3, damping 0.1, armature 0.001), no tendons, no fingertip or force-sensor columns in the observations (88 / 72 / 50 wide,
:103-107), the 6.5 cm cube of cube_multicolor_allegro.urdf.  The engine side is the same template instantiated for that model
(csrc/hand_task_kernels.hpp, AllegroHandTask); the hand's mesh collision shapes are sampled by spheres (assets/mesh.py).

Host side only: config -> MiHandParams and the reference's attribute names as views of the engine arena.  Not modelled: the drives'
effort limit (0.5 N m, :259) and joint friction (0.01, :262), like the ShadowHand's force ranges; hand self-collision.
"""
from __future__ import annotations

import numpy as np
import torch

from ..assets.model import quat_mul
from ..registry import load_extras, load_model
from .base.vec_task import VecTask
from .shadow_hand import OBJECT_SHAPE_ID, hand_max_episode_length, hand_params_from_cfg  # noqa: F401

CUBE_SIZE, CUBE_DENSITY = 0.065, 400.0      # assets/urdf/objects/cube_multicolor_allegro.urdf (cfg asset.assetFileNameBlock)
NUM_OBS = {"full_no_vel": 50, "full": 72, "full_state": 88}          # allegro_hand.py:103-107
OBS_TYPE_ID = {"full_state": 0, "full_no_vel": 2, "full": 3}


def obs_columns(obs_type):
    """Columns of compute_full_state's 88-vector (allegro_hand.py:485-507: dof pos 0:16, dof vel 16:32, dof force 32:48, object pose 48:55,
    linvel 55:58, angvel 58:61, goal pose 61:68, relative rotation 68:72, actions 72:88) that make up compute_full_observations' layouts
    (:441-460)."""
    r = lambda a, b: list(range(a, b))
    if obs_type == "full_no_vel":
        return r(0, 16) + r(48, 55) + r(61, 68) + r(68, 72) + r(72, 88)
    if obs_type == "full":
        return r(0, 32) + r(48, 61) + r(61, 72) + r(72, 88)
    if obs_type == "full_state":
        return r(0, 88)
    raise Exception("Unknown type of observations!\nobservationType should be one of: [openai, full_no_vel, full, full_state]")


def _axis_angle(axis, angle):
    a = np.asarray(axis, float)
    return np.concatenate([a * np.sin(angle / 2), [np.cos(angle / 2)]])


def hand_start_quat():
    """allegro_hand.py:283: Quat.from_axis_angle(y, pi) * from_axis_angle(x, 0.47 pi) * from_axis_angle(z, 0.25 pi) (xyzw)"""
    return quat_mul(quat_mul(_axis_angle((0, 1, 0), np.pi), _axis_angle((1, 0, 0), 0.47 * np.pi)), _axis_angle((0, 0, 1), 0.25 * np.pi))


def allegro_params_from_cfg(cfg):
    # object start: hand position + (0, pose_dy, pose_dz) with pose_dy, pose_dz = -0.2, 0.06 hard-coded (:288), the pen 0.02 above the
    # hand (:293-294); the YAML's startObjectPoseDY / DZ are not read by the reference either
    return hand_params_from_cfg(cfg, model="allegro_hand", hand_quat=hand_start_quat(), object_offset=(0.0, -0.2, 0.06), pen_offset_z=0.02,
                                cube=(CUBE_SIZE, CUBE_DENSITY), obs_layout=(NUM_OBS, OBS_TYPE_ID, obs_columns))


class AllegroHand(VecTask):
    native_task = "AllegroHand"
    model_name = "allegro_hand"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        env = cfg["env"]
        if env["objectType"] not in ("block", "egg", "pen"):                        # allegro_hand.py:85-86
            raise AssertionError("objectType must be one of block, egg, pen")
        if env["observationType"] not in NUM_OBS:                                   # :99-101
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
        cfg["env"]["numObservations"] = self.num_obs_dict[self.obs_type]           # :117-119
        cfg["env"]["numStates"] = 88 if self.asymmetric_obs else 0
        cfg["env"]["numActions"] = 16
        cfg["env"].setdefault("plane", {"staticFriction": 1.0})
        self.spec = load_model("allegro_hand")
        self.num_shadow_hand_dofs = self.spec.nd                                     # the reference keeps the ShadowHand task's attribute names
        self.num_shadow_hand_bodies = self.spec.nb
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        t = self.engine.tensors
        dev = self.device
        ex = load_extras("allegro_hand")
        self.dof_state = t["dof_state"]
        self.shadow_hand_dof_pos, self.shadow_hand_dof_vel = self.dof_state[..., 0], self.dof_state[..., 1]
        self.dof_force_tensor = t["dof_force"]
        self.object_state = t["object_state"]
        self.object_pose, self.object_pos, self.object_rot = self.object_state[:, 0:7], self.object_state[:, 0:3], self.object_state[:, 3:7]
        self.object_linvel, self.object_angvel = self.object_state[:, 7:10], self.object_state[:, 10:13]
        self.goal_states = t["goal_states"]
        self.goal_pose, self.goal_pos, self.goal_rot = self.goal_states, self.goal_states[:, 0:3], self.goal_states[:, 3:7]
        self.cur_targets, self.prev_targets = t["cur_targets"], t["prev_targets"]
        self.actions = t["actions"]
        self.successes, self.consecutive_successes = t["successes"], t["consecutive_successes"]
        self.reset_goal_buf = t["reset_goal_buf"]
        if self.asymmetric_obs:
            self.states_buf = t["states_buf"]                                         # compute_full_state(True), :462-484
        self.rb_forces_object, self.random_force_prob = t["rb_forces_object"], t["random_force_prob"]
        lo = np.minimum(self.spec.dof_lower, self.spec.dof_upper); up = np.maximum(self.spec.dof_lower, self.spec.dof_upper)
        self.shadow_hand_dof_lower_limits = torch.tensor(lo, dtype=torch.float32, device=dev)
        self.shadow_hand_dof_upper_limits = torch.tensor(up, dtype=torch.float32, device=dev)
        self.actuated_dof_indices = torch.tensor(ex["actuated_dofs"], dtype=torch.long, device=dev)
        self.extras["consecutive_successes"] = self.consecutive_successes[0]       # allegro_hand.py:406

    def _viewer_extras(self, env):
        """the manipulated object (drawn by its bounding ball) and the goal pose beside it (as tasks/shadow_hand.py)"""
        obj = self.object_pos[env].detach().cpu().numpy().astype(np.float64)
        goal = self.goal_pos[env].detach().cpu().numpy().astype(np.float64)
        r = 0.045
        return np.stack([obj, goal]), np.array([r, r]), np.array([[0.9, 0.75, 0.2], [0.55, 0.9, 0.55]])

    def _task_params(self):
        return allegro_params_from_cfg(self.cfg)

    #: `actor_params` entries of the two actors -> columns of the `actor_scale` tensor (csrc/core/hand_engine.hpp HS_*; reference
    #: cfg/task/AllegroHand.yaml:97-161).  dof_properties.stiffness is the position drives' kp (DOF_MODE_POS).
    HAND_SCALE_COLUMNS = {("hand", "rigid_body_properties", "mass"): 0, ("hand", "dof_properties", "damping"): 1,
                          ("hand", "dof_properties", "stiffness"): 2, ("object", "rigid_body_properties", "mass"): 5,
                          ("object", "scale", "scale"): 6}

    def _actor_scale_column(self, actor, group, attr):
        return self.HAND_SCALE_COLUMNS.get((actor, group, attr))

    def _actor_reference_value(self, group, attr, actor=None):
        """what an `additive` draw is relative to (a `scaling` draw is the factor itself)"""
        ex = load_extras("allegro_hand")
        if group == "scale":
            return 1.0
        if actor == "object":
            return float(self._task_params_struct.cube_mass)
        if (group, attr) == ("dof_properties", "stiffness"):
            return float(np.mean(ex["dof_kp"]))
        return super()._actor_reference_value(group, attr, actor)

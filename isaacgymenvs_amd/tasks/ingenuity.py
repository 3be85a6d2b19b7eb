"""Ingenuity: the Mars helicopter flying to random targets that move every 500 steps (reference isaacgymenvs/tasks/ingenuity.py).

Host side only: config -> MiIngenuityParams and the reference's attribute names as views of the engine arena.  The asset the
reference generates in code (:120-231) is restated in assets/procedural.py (without its GLB meshes, which are not part of the
reference tree) and compiled into models/ingenuity.json; pre_physics_step (:321-354), the thrust forces, compute_observations
(:386-391) and compute_ingenuity_reward (:407-445) run in csrc/kernels_ingenuity.hip.

The reference views one root tensor as [N, 2, 13] (craft, marker); here the two actors' root states are the tensors
`root_states` and `marker_states`.
"""
from __future__ import annotations

import math

import numpy as np

from .. import native
from ..registry import load_model
from .base.vec_task import VecTask


def ingenuity_params_from_cfg(cfg):
    p = native.MiIngenuityParams()
    p.max_episode_length = float(cfg["env"]["maxEpisodeLength"])
    p.dt = float(cfg["sim"]["dt"])
    p.thrust_upper_limit = 2000.0                        # ingenuity.py:91
    p.thrust_lateral_component = 0.2                     # :92
    p.thrust_action_speed_scale = 2000.0                 # :337
    p.max_angular_velocity = 4 * math.pi                 # :248
    p.init_height = 1.0                                  # :254
    p.rotor_speed = 50.0                                 # :298-299
    p.target_period = 500                                # :324
    ca = cfg["env"].get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    return p


class Ingenuity(VecTask):
    native_task = "Ingenuity"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        self.max_episode_length = cfg["env"]["maxEpisodeLength"]
        self.debug_viz = cfg["env"].get("enableDebugVis", False)
        cfg["env"]["numObservations"] = 13               # :49-50: target offset 3 + quaternion 4 + linear 3 + angular velocity 3
        cfg["env"]["numActions"] = 6                     # :52-55: one thrust vector per rotor
        cfg["env"].setdefault("plane", {"staticFriction": 1.0})
        cfg["sim"]["gravity"] = [0.0, 0.0, -3.721]       # create_sim overrides the YAML with Mars gravity (:109-112)
        self.spec = load_model("ingenuity")
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        self.dt = self.sim_params.dt
        t = self.engine.tensors
        self.root_states = t["root_states"]
        self.root_positions, self.root_quats = self.root_states[..., 0:3], self.root_states[..., 3:7]
        self.root_linvels, self.root_angvels = self.root_states[..., 7:10], self.root_states[..., 10:13]
        self.marker_states = t["marker_states"]
        self.marker_positions = self.marker_states[..., 0:3]
        self.target_root_positions = t["target_root_positions"]
        self.dof_states = t["dof_state"]
        self.dof_positions, self.dof_velocities = self.dof_states[..., 0], self.dof_states[..., 1]
        self.initial_root_states = t["initial_root_states"]
        self.thrusts, self.forces = t["thrusts"], t["forces"]
        self.num_dofs = self.spec.nd
        self.thrust_lower_limit, self.thrust_upper_limit, self.thrust_lateral_component = 0, 2000, 0.2

    def _task_params(self):
        return ingenuity_params_from_cfg(self.cfg)

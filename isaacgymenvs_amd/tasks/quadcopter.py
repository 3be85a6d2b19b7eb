"""Quadcopter: free-flying chassis with four tilting rotors hovering at (0, 0, 1) (reference isaacgymenvs/tasks/quadcopter.py).

Host side only: config -> MiQuadcopterParams and the reference's attribute names as views of the engine arena.  The asset
the reference generates in code (:119-198) is restated in assets/procedural.py and compiled into models/quadcopter.json;
pre_physics_step (:276-292), the position drives + thrust forces, compute_observations (:320-331) and
compute_quadcopter_reward (:348-386) run in csrc/kernels_quadcopter.hip.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import native
from ..registry import load_model
from .base.vec_task import VecTask


def quadcopter_params_from_cfg(cfg, spec):
    p = native.MiQuadcopterParams()
    p.max_episode_length = float(cfg["env"]["maxEpisodeLength"])
    p.dt = float(cfg["sim"]["dt"])
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    for d in range(8):
        p.dof_lower[d], p.dof_upper[d] = float(lo[d]), float(up[d])
    p.max_thrust = 2.0                                   # quadcopter.py:88
    p.dof_action_speed_scale = 8 * math.pi               # :283
    p.thrust_action_speed_scale = 200.0                  # :287
    p.drive_stiffness, p.drive_damping = 1000.0, 0.0     # :236-238
    p.max_angular_velocity = 4 * math.pi                 # :208
    p.init_height = 1.0                                  # :226
    ca = cfg["env"].get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    return p


class Quadcopter(VecTask):
    native_task = "Quadcopter"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        self.max_episode_length = cfg["env"]["maxEpisodeLength"]
        self.debug_viz = cfg["env"].get("enableDebugVis", False)
        cfg["env"]["numObservations"] = 21               # :52-57: root state 13 + 8 dof positions
        cfg["env"]["numActions"] = 12                    # 8 rotor dof targets + 4 thrusts
        cfg["env"].setdefault("plane", {"staticFriction": 1.0})
        self.spec = load_model("quadcopter")
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        self.dt = self.sim_params.dt
        t = self.engine.tensors
        self.root_states = t["root_states"]
        self.root_positions, self.root_quats = self.root_states[..., 0:3], self.root_states[..., 3:7]
        self.root_linvels, self.root_angvels = self.root_states[..., 7:10], self.root_states[..., 10:13]
        self.dof_states = t["dof_state"]
        self.dof_positions, self.dof_velocities = self.dof_states[..., 0], self.dof_states[..., 1]
        self.initial_root_states = t["initial_root_states"]
        self.dof_position_targets, self.thrusts, self.forces = t["dof_position_targets"], t["thrusts"], t["forces"]
        self.num_dofs = self.spec.nd
        p = self._task_params_struct
        self.dof_lower_limits = torch.tensor(p.dof_lower[:], device=self.device)
        self.dof_upper_limits = torch.tensor(p.dof_upper[:], device=self.device)
        self.dof_ranges = self.dof_upper_limits - self.dof_lower_limits
        self.thrust_lower_limits = torch.zeros(4, device=self.device)
        self.thrust_upper_limits = p.max_thrust * torch.ones(4, device=self.device)

    def _task_params(self):
        return quadcopter_params_from_cfg(self.cfg, self.spec)

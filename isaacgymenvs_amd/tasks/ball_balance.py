"""BallBalance: a tray on three actuated legs keeps a dropped ball near its centre (reference isaacgymenvs/tasks/ball_balance.py).

Host side only: config + the lengths of the generated asset -> MiBallBalanceParams, and the reference's attribute names as views of
the engine arena.  The asset the reference generates in code (:136-224) is restated in assets/procedural.py and compiled into
models/balance_bot.json; pre_physics_step (:395-413), the physics with the attractor-pinned feet and the ball (csrc/core/
bbot_engine.hpp), compute_observations (:322-337) and compute_bbot_reward (:459-476) run in csrc/kernels_ball_balance.hip.

The reference views one root tensor as [N, 2, 13] (bot, ball); here the two actors' root states are the tensors `root_states`
(the tray) and `ball_states`.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import native
from ..assets.procedural import balance_bot_dims
from ..registry import load_model
from .base.vec_task import VecTask


def ball_balance_params_from_cfg(cfg, spec):
    d = balance_bot_dims()
    p = native.MiBallBalanceParams()
    p.max_episode_length = float(cfg["env"]["maxEpisodeLength"])
    p.dt = float(cfg["sim"]["dt"])
    p.action_speed_scale = float(cfg["env"]["actionSpeedScale"])
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    for k in range(6):
        p.dof_lower[k], p.dof_upper[k] = float(lo[k]), float(up[k])
    p.tray_height = d["tray_height"]                      # ball_balance.py:251-252
    p.ball_init_pos[0], p.ball_init_pos[1], p.ball_init_pos[2] = 0.2, 0.0, 2.0     # :303-306
    ca = cfg["env"].get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    p.pin_stiffness, p.pin_damping = 5e7, 5e3             # :287-288
    p.drive_kp, p.drive_kd = 4000.0, 100.0                # :276-277
    p.actuated_mask = (1 << 1) | (1 << 3) | (1 << 5)      # :271
    radius, density = 0.1, 200.0                          # :263-265
    p.ball_radius = radius
    p.ball_mass = density * 4.0 / 3.0 * math.pi * radius ** 3
    p.ball_inertia = 0.4 * p.ball_mass * radius ** 2
    p.mu = 1.0                                            # default shape friction on both sides
    p.tray_radius, p.tray_half = d["tray_radius"], 0.5 * d["tray_thickness"]
    p.pin_offset[0], p.pin_offset[1], p.pin_offset[2] = 0.0, 0.0, 0.5 * d["leg_length"]     # :299
    for j, a in enumerate(d["leg_angles"]):
        x, y = d["leg_outer_offset"] * math.cos(a), d["leg_outer_offset"] * math.sin(a)
        p.pin_target[j][0], p.pin_target[j][1], p.pin_target[j][2] = x, y, d["leg_radius"]  # :293-297
        p.sensor_pos[j][0], p.sensor_pos[j][1], p.sensor_pos[j][2] = x, y, 0.0              # :256-259
    return p


class BallBalance(VecTask):
    native_task = "BallBalance"
    model_name = "balance_bot"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        self.max_episode_length = cfg["env"]["maxEpisodeLength"]
        self.action_speed_scale = cfg["env"]["actionSpeedScale"]
        self.debug_viz = cfg["env"].get("enableDebugVis", False)
        cfg["env"]["numObservations"] = 24               # :66-76
        cfg["env"]["numActions"] = 3                     # :78-79: target velocities of the three actuated dofs
        cfg["env"].setdefault("plane", {"staticFriction": 1.0})
        cfg["sim"]["gravity"] = [0.0, 0.0, -9.81]        # create_sim (:125-128)
        self.spec = load_model("balance_bot")
        dims = balance_bot_dims()
        self.tray_height, self.leg_radius, self.leg_length = dims["tray_height"], dims["leg_radius"], dims["leg_length"]
        self.leg_outer_offset, self.leg_angles = dims["leg_outer_offset"], dims["leg_angles"]
        self.ball_radius = 0.1
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        self.dt = self.sim_params.dt
        t = self.engine.tensors
        self.root_states = t["root_states"]              # the bot's root = the tray
        self.tray_positions = self.root_states[..., 0:3]
        self.ball_states = t["ball_states"]
        self.ball_positions, self.ball_orientations = self.ball_states[..., 0:3], self.ball_states[..., 3:7]
        self.ball_linvels, self.ball_angvels = self.ball_states[..., 7:10], self.ball_states[..., 10:13]
        self.dof_states = t["dof_state"]
        self.dof_positions, self.dof_velocities = self.dof_states[..., 0], self.dof_states[..., 1]
        self.vec_sensor_tensor = t["force_sensor"]
        self.sensor_forces, self.sensor_torques = self.vec_sensor_tensor[..., 0:3], self.vec_sensor_tensor[..., 3:6]
        self.initial_root_states = t["initial_root_states"]
        self.dof_position_targets = t["dof_position_targets"]
        self.num_bbot_dofs = self.spec.nd
        p = self._task_params_struct
        self.bbot_dof_lower_limits = torch.tensor(p.dof_lower[:], device=self.device)
        self.bbot_dof_upper_limits = torch.tensor(p.dof_upper[:], device=self.device)

    def _task_params(self):
        return ball_balance_params_from_cfg(self.cfg, self.spec)

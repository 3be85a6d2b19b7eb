"""Cartpole: fixed-base slider + free pole, effort on DoF 0 (reference isaacgymenvs/tasks/cartpole.py)."""
from __future__ import annotations

import numpy as np

from .. import native
from .base.vec_task import VecTask


def cartpole_params_from_cfg(cfg):
    p = native.MiCartpoleParams()
    p.reset_dist = float(cfg["env"]["resetDist"])
    p.max_push_effort = float(cfg["env"]["maxEffort"])
    p.max_episode_length = 500.0  # cartpole.py:44
    ca = cfg["env"].get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    return p


class Cartpole(VecTask):
    native_task = "Cartpole"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        self.reset_dist = cfg["env"]["resetDist"]
        self.max_push_effort = cfg["env"]["maxEffort"]
        self.max_episode_length = 500
        self.cfg["env"]["numObservations"] = 4
        self.cfg["env"]["numActions"] = 1
        self.cfg["env"].setdefault("plane", {"staticFriction": 1.0})
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        t = self.engine.tensors
        self.num_dof = 2
        self.dof_state = t["dof_state"]
        self.dof_pos = self.dof_state[..., 0]
        self.dof_vel = self.dof_state[..., 1]

    def _task_params(self):
        return cartpole_params_from_cfg(self.cfg)

"""Anymal: ANYmal-C on flat ground, PD position drives, velocity-command tracking (reference isaacgymenvs/tasks/anymal.py).

Host side only: config -> MiAnymalFlatParams and the reference's attribute names.  pre_physics_step (:226-229), the PD
drive, the physics, reset_idx (:274-301), compute_anymal_observations (:354-386) and compute_anymal_reward (:311-351) run
in csrc/kernels_anymal.hip (anymal_flat_post_kernel).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import native
from ..registry import load_model
from .base.vec_task import VecTask


def anymal_flat_params_from_cfg(cfg, dof_names):
    env, learn, ctrl = cfg["env"], cfg["env"]["learn"], cfg["env"]["control"]
    p = native.MiAnymalFlatParams()
    p.lin_vel_scale = learn["linearVelocityScale"]; p.ang_vel_scale = learn["angularVelocityScale"]      # :47-51
    p.dof_pos_scale = learn["dofPositionScale"]; p.dof_vel_scale = learn["dofVelocityScale"]
    p.action_scale = ctrl["actionScale"]
    dt = float(cfg["sim"]["dt"])                                                                          # :90
    p.rew_lin_vel_xy = float(learn["linearVelocityXYRewardScale"]) * dt                                   # :96-97
    p.rew_ang_vel_z = float(learn["angularVelocityZRewardScale"]) * dt
    p.rew_torque = float(learn["torqueRewardScale"]) * dt
    rng = env["randomCommandVelocityRanges"]
    for k in range(2):
        p.command_x[k] = rng["linear_x"][k]; p.command_y[k] = rng["linear_y"][k]; p.command_yaw[k] = rng["yaw"][k]
    b = env["baseInitState"]
    for k, val in enumerate(list(b["pos"]) + list(b["rot"]) + list(b["vLinear"]) + list(b["vAngular"])):
        p.base_init_state[k] = float(val)
    for i, n in enumerate(dof_names):
        p.default_dof_pos[i] = float(env["defaultJointAngles"][n])
    # DOF_MODE_POS drive with these gains (:203-206); the drive force saturates at the URDF effort limit (80 N m)
    p.kp, p.kd, p.torque_limit = float(ctrl["stiffness"]), float(ctrl["damping"]), 80.0
    p.max_episode_length = int(float(learn["episodeLength_s"]) / dt + 0.5)                               # :92
    ca = env.get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    return p


class Anymal(VecTask):
    native_task = "Anymal"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        env = cfg["env"]
        self.spec = load_model("anymal")
        self.dof_names = list(self.spec.dof_names)
        self.num_dof, self.num_bodies = self.spec.nd, self.spec.nb
        self.lin_vel_scale = env["learn"]["linearVelocityScale"]
        self.ang_vel_scale = env["learn"]["angularVelocityScale"]
        self.dof_pos_scale = env["learn"]["dofPositionScale"]
        self.dof_vel_scale = env["learn"]["dofVelocityScale"]
        self.action_scale = env["control"]["actionScale"]
        self.command_x_range = env["randomCommandVelocityRanges"]["linear_x"]
        self.command_y_range = env["randomCommandVelocityRanges"]["linear_y"]
        self.command_yaw_range = env["randomCommandVelocityRanges"]["yaw"]
        self.named_default_joint_angles = env["defaultJointAngles"]
        self.randomize = cfg["task"].get("randomize", False)
        self.randomization_params = cfg["task"].get("randomization_params", {})
        b = env["baseInitState"]
        self.base_init_state = list(b["pos"]) + list(b["rot"]) + list(b["vLinear"]) + list(b["vAngular"])
        cfg["env"]["numObservations"] = 48                                                                # :84-85
        cfg["env"]["numActions"] = 12
        # env.control.controlFrequencyInv is never read by the reference (VecTask looks at env.controlFrequencyInv -> 1)
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        self.dt = self.sim_params.dt
        self.max_episode_length_s = env["learn"]["episodeLength_s"]
        self.max_episode_length = int(self.max_episode_length_s / self.dt + 0.5)
        self.Kp, self.Kd = env["control"]["stiffness"], env["control"]["damping"]
        self.rew_scales = {"lin_vel_xy": env["learn"]["linearVelocityXYRewardScale"] * self.dt,
                           "ang_vel_z": env["learn"]["angularVelocityZRewardScale"] * self.dt,
                           "torque": env["learn"]["torqueRewardScale"] * self.dt}
        t = self.engine.tensors
        self.root_states = t["root_states"]
        self.dof_state = t["dof_state"]
        self.dof_pos, self.dof_vel = self.dof_state[..., 0], self.dof_state[..., 1]
        self.contact_forces = t["net_contact_force"]
        self.torques = t["dof_force"]                       # gym.acquire_dof_force_tensor (:103)
        self.commands = t["commands"]
        self.commands_x, self.commands_y, self.commands_yaw = self.commands[..., 0], self.commands[..., 1], self.commands[..., 2]
        self.actions = t["actions"]
        self.initial_root_states = t["initial_root_states"]
        p = self._task_params_struct
        self.default_dof_pos = torch.tensor([p.default_dof_pos[i] for i in range(self.num_dof)], device=self.device).repeat(self.num_envs, 1)
        self.gravity_vec = torch.tensor([0.0, 0.0, -1.0], device=self.device).repeat(self.num_envs, 1)
        extremity = "SHANK" if env["urdfAsset"]["collapseFixedJoints"] else "FOOT"
        self.feet_indices = torch.tensor([i for i, n in enumerate(self.spec.body_names) if extremity in n], device=self.device)
        self.knee_indices = torch.tensor([i for i, n in enumerate(self.spec.body_names) if "THIGH" in n], device=self.device)
        self.base_index = 0

    def _task_params(self):
        return anymal_flat_params_from_cfg(self.cfg, self.dof_names)

    def _actor_scale_column(self, actor, group, attr):
        """`actor_params.anymal` (Anymal.yaml:121-165): the dofs' `stiffness` / `damping` properties are the position drives' gains
        (anymal.py:203-206 writes env.control.stiffness / damping into every dof), so their factors scale kp / kd per dof in the sub-step
        (csrc/step_kernels.hpp efforts_for_substep); link masses per body as for Ant / Humanoid."""
        if (group, attr) == ("dof_properties", "stiffness"):
            return self.spec.nb + self.spec.nd, np.full(self.spec.nd, float(self.Kp))
        if (group, attr) == ("dof_properties", "damping"):
            return self.spec.nb, np.full(self.spec.nd, float(self.Kd))
        return super()._actor_scale_column(actor, group, attr)

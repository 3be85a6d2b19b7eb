"""Shared host side of the Ant and Humanoid tasks (reference tasks/ant.py, tasks/humanoid.py).

All per-step maths (pre_physics_step, physics, reset_idx, compute_*_observations, compute_*_reward) run inside the
fused HIP kernel; this class resolves config -> MiLocoParams and exposes the reference's attribute names as views.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import native
from ..registry import load_model
from .base.vec_task import VecTask


def loco_params_from_cfg(cfg, model_name, start_height):
    """Build MiLocoParams from a task config dict + compiled model (shared with tests / oracle)."""
    env = cfg["env"]
    spec = load_model(model_name)
    nd = spec.nd
    p = native.MiLocoParams()
    p.dt = float(cfg["sim"]["dt"])
    p.dof_vel_scale = float(env["dofVelocityScale"])
    p.contact_force_scale = float(env["contactForceScale"])
    p.angular_velocity_scale = float(env.get("angularVelocityScale", 0.1))  # humanoid.py:50
    p.power_scale = float(env["powerScale"])
    p.heading_weight = float(env["headingWeight"])
    p.up_weight = float(env["upWeight"])
    p.actions_cost = float(env["actionsCost"])
    p.energy_cost = float(env["energyCost"])
    p.joints_at_limit_cost = float(env["jointsAtLimitCost"])
    p.death_cost = float(env["deathCost"])
    p.termination_height = float(env["terminationHeight"])
    p.max_episode_length = float(env["episodeLength"])
    ca = env.get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    # motor efforts in actuator-file order, used positionally against dof order exactly like the reference
    # (ant.py:158-161,283 ; humanoid.py:158-160,283)
    gears = [float(g) for g in spec.act_gear]
    p.max_motor_effort = max(gears)
    p.start_height = float(start_height)
    lo = np.minimum(spec.dof_lower, spec.dof_upper)  # swap if lower > upper (ant.py:199-206)
    up = np.maximum(spec.dof_lower, spec.dof_upper)
    init = np.where(lo > 0, lo, np.where(up < 0, up, 0.0))  # ant.py:98-101
    for d in range(nd):
        p.gear[d] = gears[d]
        p.dof_lower[d], p.dof_upper[d], p.initial_dof_pos[d] = float(lo[d]), float(up[d]), float(init[d])
    p.targets[0], p.targets[1], p.targets[2] = 1000.0, 0.0, 0.0  # ant.py:110
    p.inv_start_rot[0] = p.inv_start_rot[1] = p.inv_start_rot[2] = 0.0
    p.inv_start_rot[3] = 1.0
    p.basis_vec0[0], p.basis_vec0[1], p.basis_vec0[2] = 1.0, 0.0, 0.0  # heading_vec (ant.py:104)
    p.basis_vec1[0], p.basis_vec1[1], p.basis_vec1[2] = 0.0, 0.0, 1.0  # up_vec
    p.reset_pos_noise, p.reset_vel_noise = 0.2, 0.1  # ant.py:257-258
    return p


class LocomotionTask(VecTask):
    model_name = ""
    start_height = 0.0

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        env = cfg["env"]
        self.max_episode_length = env["episodeLength"]
        self.randomization_params = cfg["task"].get("randomization_params", {})
        self.randomize = cfg["task"]["randomize"]
        self.dof_vel_scale = env["dofVelocityScale"]
        self.contact_force_scale = env["contactForceScale"]
        self.power_scale = env["powerScale"]
        self.heading_weight = env["headingWeight"]
        self.up_weight = env["upWeight"]
        self.actions_cost_scale = env["actionsCost"]
        self.energy_cost_scale = env["energyCost"]
        self.joints_at_limit_cost_scale = env["jointsAtLimitCost"]
        self.death_cost = env["deathCost"]
        self.termination_height = env["terminationHeight"]
        self.debug_viz = env.get("enableDebugVis", False)
        self.plane_static_friction = env["plane"]["staticFriction"]
        self.plane_dynamic_friction = env["plane"]["dynamicFriction"]
        self.plane_restitution = env["plane"]["restitution"]
        if float(self.plane_restitution) != 0.0:
            # every shipped config has restitution 0 (Ant.yaml:34, Humanoid.yaml:37; the `actor_params` restitution entries of Humanoid.yaml:102
            # SCALE that zero, i.e. change nothing): the contact model has no restitution term, so anything else is refused, not ignored
            raise NotImplementedError(f"plane restitution {self.plane_restitution}: the MI355X engine's contact model has no restitution (the reference configs use 0)")
        info = native.task_info(self.native_task)
        self.cfg["env"]["numObservations"] = info.num_obs
        self.cfg["env"]["numActions"] = info.num_actions
        self.num_dof = info.num_dofs
        self.num_bodies = info.num_bodies
        self.up_axis_idx = 2
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        t = self.engine.tensors
        p = self._task_params_struct
        dev = self.device
        # baseline of the `rigid_shape_properties.friction` randomisation: the compiled model's own shape friction (Ant 1.5)
        fr = load_model(self.model_name).sph_friction
        self.model_shape_friction = float(fr[0]) if len(fr) else 1.0
        # reference attribute names (ant.py:77-114)
        self.root_states = t["root_states"]
        self.initial_root_states = t["initial_root_states"]
        self.dof_state = t["dof_state"]
        self.dof_pos = self.dof_state[..., 0]
        self.dof_vel = self.dof_state[..., 1]
        self.vec_sensor_tensor = t["force_sensor"].view(self.num_envs, -1) if False else \
            torch.as_strided(t["force_sensor"], (self.num_envs, t["force_sensor"].shape[1] * 6), (1, self.num_envs))
        self.dof_force_tensor = t["dof_force"]
        self.actions = t["actions"]
        self.potentials = t["potentials"]
        self.prev_potentials = t["prev_potentials"]
        self.up_vec = t["up_vec"]
        self.heading_vec = t["heading_vec"]
        nd = self.num_dof
        self.dof_limits_lower = torch.tensor([p.dof_lower[d] for d in range(nd)], device=dev)
        self.dof_limits_upper = torch.tensor([p.dof_upper[d] for d in range(nd)], device=dev)
        self.initial_dof_pos = torch.tensor([p.initial_dof_pos[d] for d in range(nd)], device=dev).repeat(self.num_envs, 1)
        self.initial_dof_vel = torch.zeros_like(self.initial_dof_pos)
        self.joint_gears = torch.tensor([p.gear[d] for d in range(nd)], device=dev)
        self.motor_efforts = self.joint_gears
        self.max_motor_effort = p.max_motor_effort
        self.start_rotation = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)
        self.inv_start_rot = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev).repeat(self.num_envs, 1)
        self.basis_vec0 = torch.tensor([1.0, 0.0, 0.0], device=dev).repeat(self.num_envs, 1)
        self.basis_vec1 = torch.tensor([0.0, 0.0, 1.0], device=dev).repeat(self.num_envs, 1)
        self.targets = torch.tensor([1000.0, 0.0, 0.0], device=dev).repeat(self.num_envs, 1)
        self.target_dirs = torch.tensor([1.0, 0.0, 0.0], device=dev).repeat(self.num_envs, 1)

    def _task_params(self):
        return loco_params_from_cfg(self.cfg, self.model_name, self.start_height)

    def _post_step_extras(self):
        # compute_true_objective (ant.py:245-250): forward velocity
        self.extras["true_objective"] = self.root_states[:, 7]

"""AnymalTerrain: ANYmal-C (13 bodies / 12 DoFs after collapsing fixed joints) on procedurally generated rough terrain,
PD position control with decimation, velocity-command tracking (reference isaacgymenvs/tasks/anymal_terrain.py).

Host side only: config -> MiAnymalParams, terrain generation (tasks/terrain.py), attribute names of the reference.  The
per-step maths (PD torques x decimation, physics on the height field, termination, 13 reward terms, curriculum, resets,
height scan, noise) run in csrc/kernels_anymal.hip.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import native
from ..registry import load_model
from .base.vec_task import VecTask
from .terrain import Terrain

# episode_sums keys in kernel order (anymal_terrain.py:166-168)
SUM_KEYS = ("lin_vel_xy", "lin_vel_z", "ang_vel_z", "ang_vel_xy", "orient", "torques", "joint_acc", "base_height",
            "air_time", "collision", "stumble", "action_rate", "hip")


def anymal_params_from_cfg(cfg, dof_names):
    env, learn, ctrl = cfg["env"], cfg["env"]["learn"], cfg["env"]["control"]
    p = native.MiAnymalParams()
    p.lin_vel_scale = learn["linearVelocityScale"]; p.ang_vel_scale = learn["angularVelocityScale"]
    p.dof_pos_scale = learn["dofPositionScale"]; p.dof_vel_scale = learn["dofVelocityScale"]
    p.height_meas_scale = learn["heightMeasurementScale"]; p.action_scale = ctrl["actionScale"]
    decimation = int(ctrl["decimation"])
    dt = decimation * float(cfg["sim"]["dt"])                                    # :95
    for name, key in (("rew_termination", "terminalReward"), ("rew_lin_vel_xy", "linearVelocityXYRewardScale"),
                      ("rew_lin_vel_z", "linearVelocityZRewardScale"), ("rew_ang_vel_z", "angularVelocityZRewardScale"),
                      ("rew_ang_vel_xy", "angularVelocityXYRewardScale"), ("rew_orient", "orientationRewardScale"),
                      ("rew_torque", "torqueRewardScale"), ("rew_joint_acc", "jointAccRewardScale"),
                      ("rew_base_height", "baseHeightRewardScale"), ("rew_air_time", "feetAirTimeRewardScale"),
                      ("rew_collision", "kneeCollisionRewardScale"), ("rew_stumble", "feetStumbleRewardScale"),
                      ("rew_action_rate", "actionRateRewardScale"), ("rew_hip", "hipRewardScale")):
        setattr(p, name, float(learn[key]) * dt)                                 # :104-105
    rng = env["randomCommandVelocityRanges"]
    for k in range(2):
        p.command_x[k] = rng["linear_x"][k]; p.command_y[k] = rng["linear_y"][k]; p.command_yaw[k] = rng["yaw"][k]
    b = env["baseInitState"]
    for k, val in enumerate(list(b["pos"]) + list(b["rot"]) + list(b["vLinear"]) + list(b["vAngular"])):
        p.base_init_state[k] = float(val)
    for i, n in enumerate(dof_names):
        p.default_dof_pos[i] = float(env["defaultJointAngles"][n])
    p.kp, p.kd, p.torque_limit = float(ctrl["stiffness"]), float(ctrl["damping"]), 80.0   # :444-445
    p.dt = dt
    p.max_episode_length_s = float(learn["episodeLength_s"])
    p.max_episode_length = int(p.max_episode_length_s / dt + 0.5)
    p.push_interval = int(learn["pushInterval_s"] / dt + 0.5) if learn.get("pushRobots", True) else 0
    p.allow_knee_contacts = int(bool(learn["allowKneeContacts"]))
    p.decimation = decimation
    p.add_noise = int(bool(learn["addNoise"]))
    nl = float(learn["noiseLevel"])
    p.noise_lin_vel = learn["linearVelocityNoise"] * nl * p.lin_vel_scale       # :174-186
    p.noise_ang_vel = learn["angularVelocityNoise"] * nl * p.ang_vel_scale
    p.noise_gravity = learn["gravityNoise"] * nl
    p.noise_dof_pos = learn["dofPositionNoise"] * nl * p.dof_pos_scale
    p.noise_dof_vel = learn["dofVelocityNoise"] * nl * p.dof_vel_scale
    p.noise_height = learn["heightMeasurementNoise"] * nl * p.height_meas_scale
    p.curriculum = int(bool(env["terrain"]["curriculum"]))
    ca = env.get("clipActions", np.inf)
    p.clip_actions = float(ca) if np.isfinite(ca) else 3.0e38
    p.friction_range[0], p.friction_range[1] = learn["frictionRange"]
    p.terrain_mu = float(env["terrain"]["staticFriction"])
    return p


class _FlatTerrain:
    """terrainType 'plane' (anymal_terrain.py:194-201): a flat height field and a single origin at (0, 0, 0)."""
    horizontal_scale, vertical_scale, border_size, env_length = 0.1, 0.005, 20.0, 8.0

    def __init__(self):
        self.heightsamples = np.zeros((512, 512), np.int16)
        self.env_origins = np.zeros((1, 1, 3))


class AnymalTerrain(VecTask):
    native_task = "AnymalTerrain"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = cfg
        env = cfg["env"]
        self.spec = load_model("anymal")
        self.dof_names = list(self.spec.dof_names)
        self.num_dof, self.num_bodies = self.spec.nd, self.spec.nb
        self.decimation = env["control"]["decimation"]
        self.dt = self.decimation * cfg["sim"]["dt"]
        self.max_episode_length_s = env["learn"]["episodeLength_s"]
        self.max_episode_length = int(self.max_episode_length_s / self.dt + 0.5)
        self.curriculum = env["terrain"]["curriculum"]
        self.custom_origins = env["terrain"]["terrainType"] == "trimesh"
        if env["terrain"]["terrainType"] not in ("plane", "trimesh"):
            raise ValueError("terrainType must be 'plane' or 'trimesh'")
        cfg["env"]["plane"] = {"staticFriction": env["terrain"]["staticFriction"]}   # terrain friction -> sim params
        if float(env["terrain"].get("restitution", 0.0)) != 0.0:    # AnymalTerrain.yaml:17 ships 0; the contact model has no restitution term
            raise NotImplementedError(f"terrain restitution {env['terrain']['restitution']}: not modelled by the MI355X engine (the reference config uses 0)")
        # the reference task never calls apply_randomizations (task.randomize is read by nobody in anymal_terrain.py; its own
        # randomisation is the per-env friction buckets and the observation noise, both in-kernel here)
        self.randomize = False
        if self.custom_origins:
            # same terrain on every rank of a multi-GPU job: the seed of the terrain stream is the job seed, not seed+rank
            self.terrain = Terrain(env["terrain"], num_robots=env["numEnvs"], seed=int(cfg.get("_terrain_seed", cfg.get("_seed", 0))))
            if not self.curriculum:
                env["terrain"]["maxInitMapLevel"] = env["terrain"]["numLevels"] - 1     # :259
            self.terrain.max_init_level = int(env["terrain"]["maxInitMapLevel"])
        else:
            self.terrain = _FlatTerrain()
            self.terrain.max_init_level = 0
        super().__init__(config=self.cfg, rl_device=rl_device, sim_device=sim_device,
                         graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        self.dt = self.decimation * cfg["sim"]["dt"]          # the reference's self.dt is the control dt (:95)
        t = self.engine.tensors
        self.root_states = t["root_states"]
        self.dof_state = t["dof_state"]
        self.dof_pos, self.dof_vel = self.dof_state[..., 0], self.dof_state[..., 1]
        # the joint state as of the reference task's last gym.refresh_dof_state_tensor: what its PD law, observations and reward read (one sim step
        # behind `dof_state` after a step -- anymal_terrain.py:441-455, vec_task.py:379-382; engine option dof_state_lag, default 1)
        self.dof_state_refreshed = t["dof_state_refreshed"]
        self.contact_forces = t["net_contact_force"]
        self.commands = t["commands"]
        self.torques = t["dof_actuation_force"]
        self.actions = t["actions"]
        self.last_actions, self.last_dof_vel = t["last_actions"], t["last_dof_vel"]
        self.feet_air_time = t["feet_air_time"]
        self.env_origins = t["env_origins"]
        self.terrain_levels, self.terrain_types = t["terrain_levels"], t["terrain_types"]
        self.episode_sums = {k: t["episode_sums"][:, i] for i, k in enumerate(SUM_KEYS)}
        self.height_samples = self.engine.height_samples
        self.feet_indices = torch.tensor([i for i, n in enumerate(self.spec.body_names) if env["urdfAsset"]["footName"] in n],
                                         device=self.device)
        self.knee_indices = torch.tensor([i for i, n in enumerate(self.spec.body_names) if env["urdfAsset"]["kneeName"] in n],
                                         device=self.device)
        self.base_index = 0
        p = self._task_params_struct
        self.default_dof_pos = torch.tensor([p.default_dof_pos[i] for i in range(self.num_dof)], device=self.device).repeat(self.num_envs, 1)
        means = t["episode_means"]
        # extras["episode"] (:421-425): persistent views of what the extras kernel refreshes each step
        self.extras["episode"] = {"rew_" + k: means[i] for i, k in enumerate(SUM_KEYS)}
        self.extras["episode"]["terrain_level"] = means[14]

    def _task_params(self):
        return anymal_params_from_cfg(self.cfg, self.dof_names)

    def create_sim(self):
        tp = self._task_params()
        self.engine = native.Engine(self.native_task, self.sim_params, tp, self.num_envs, self.device, seed=self.engine_seed,
                                    env_id_offset=self.rank * self.num_envs, terrain=self.terrain)
        self._task_params_struct = tp
        if np.isfinite(self.clip_obs):
            self.engine.set_option("clip_obs", self.clip_obs)
        self.engine.set_option("control_freq_inv", self.control_freq_inv)
        self._select_multi_wave()
        self.sim = self.engine

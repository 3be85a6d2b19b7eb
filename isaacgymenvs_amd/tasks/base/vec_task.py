"""VecTask: the public vectorised-env API of the reference, hosted on the native HIP engine.

Mirrors reference isaacgymenvs/tasks/base/vec_task.py (Env :67-205, VecTask :207-455): same constructor
signature, attributes, buffer dtypes and `step/reset/reset_done/reset_idx` contract.  What differs is *where the
work runs*: the reference's step() is ~30 torch ops around the closed `gym.simulate`; here the whole step is one
fused HIP kernel (csrc/mi_engine.hip) and this class only moves pointers.

Buffers are strided torch views of the engine's SoA arena (the analogue of gymtorch.wrap_tensor): writing through
`env.dof_pos[env_ids] = ...` edits simulator state in place, as with the reference's CPU pipeline.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Tuple

import numpy as np
import torch

from ... import native
from ...utils import spaces


class _TorchNoise:
    """Observation / action noise as torch ops, for the tasks whose kernels do not apply it themselves: value = op(x, corr + white)
    with `corr` a normal draw per element made once (scaled by the correlated range) and `white` fresh on every call."""

    def __init__(self, spec, corr=None):
        self.spec, self.corr = spec, corr

    def __call__(self, x):
        s = self.spec
        if self.corr is None or self.corr.shape != x.shape:
            self.corr = torch.randn_like(x)
        if s["dist"] == "gaussian":
            nz = self.corr * s["b_corr"] + s["a_corr"] + torch.randn_like(x) * s["b"] + s["a"]
        else:
            nz = self.corr * (s["b_corr"] - s["a_corr"]) + s["a_corr"] + torch.rand_like(x) * (s["b"] - s["a"]) + s["a"]
        return x + nz if s["op"] == "additive" else x * nz


class Env:
    """reference vec_task.py:67-205 (device resolution, spaces, clip values)."""

    def __init__(self, config: Dict[str, Any], rl_device: str, sim_device: str, graphics_device_id: int, headless: bool):
        split_device = sim_device.split(":")
        self.device_type = split_device[0]
        self.device_id = int(split_device[1]) if len(split_device) > 1 else 0
        self.device = "cpu"
        if config["sim"]["use_gpu_pipeline"]:
            if self.device_type.lower() in ("cuda", "gpu"):
                self.device = "cuda:" + str(self.device_id)
            else:
                print("GPU Pipeline can only be used with GPU simulation. Forcing CPU Pipeline.")
                config["sim"]["use_gpu_pipeline"] = False
        # self.device == "cpu": the reference runs PhysX-CPU here (vec_task.py:78-88); this engine runs its own host build (OpenMP over
        # envs, csrc/cpu/mi_engine_cpu.cpp) for the tasks that have one -- native.Engine raises for the others
        self.rl_device = rl_device
        self.headless = headless
        enable_camera_sensors = config["env"].get("enableCameraSensors", False)
        self.graphics_device_id = graphics_device_id
        if not enable_camera_sensors and self.headless:
            self.graphics_device_id = -1
        self.num_environments = config["env"]["numEnvs"]
        self.num_agents = config["env"].get("numAgents", 1)
        self.num_observations = config["env"].get("numObservations", 0)
        self.num_states = config["env"].get("numStates", 0)
        self.obs_space = spaces.Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = spaces.Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.num_actions = config["env"]["numActions"]
        self.control_freq_inv = config["env"].get("controlFrequencyInv", 1)
        self.act_space = spaces.Box(np.ones(self.num_actions) * -1., np.ones(self.num_actions) * 1.)
        self.clip_obs = config["env"].get("clipObservations", np.inf)
        self.clip_actions = config["env"].get("clipActions", np.inf)
        self.total_train_env_frames: int = 0
        self.control_steps: int = 0
        self.render_fps: int = config["env"].get("renderFPS", -1)
        self.last_frame_time: float = 0.0
        self.record_frames: bool = False

    # -- API properties (vec_task.py:162-185)
    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self) -> int:
        return self.num_environments

    @property
    def num_acts(self) -> int:
        return self.num_actions

    @property
    def num_obs(self) -> int:
        return self.num_observations

    def set_train_info(self, env_frames, *args, **kwargs):  # vec_task.py:187-194
        self.total_train_env_frames = env_frames

    def get_env_state(self):  # vec_task.py:196-200 (default None); overridden below with real physics state
        return None

    def set_env_state(self, env_state):
        pass


class VecTask(Env):
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 24}

    #: name of the native task this class drives (subclasses set it)
    native_task: str = ""

    def __init__(self, config, rl_device, sim_device, graphics_device_id, headless,
                 virtual_screen_capture: bool = False, force_render: bool = False):
        super().__init__(config, rl_device, sim_device, graphics_device_id, headless)
        self.virtual_screen_capture = virtual_screen_capture
        self.force_render = force_render
        self.viewer = None
        self.sim_params = self._parse_sim_params(self.cfg["physics_engine"], self.cfg["sim"])
        self.dt: float = self.sim_params.dt
        self.first_randomization = True
        self.last_rand_step = 0
        self.dr_randomizations = {}
        self._torch_noise = {}
        # multi-GPU sharding: global env ids are rank*num_envs + i (seed offset: reference utils/utils.py:94)
        self.rank = int(os.getenv("RANK", "0")) if config.get("_multi_gpu", False) else 0
        self.engine_seed = int(config.get("_seed", 0))
        self.sim_initialized = False
        self.create_sim()
        self.sim_initialized = True
        self.allocate_buffers()
        self.obs_dict = {}
        self._job_extras = None
        # OPT-IN: the reference's VecTask.step has no collective, and a collective inside step() obliges every rank to call step() the
        # same number of times.  `multi_gpu=True` alone therefore shards and nothing else; job-wide extras are asked for with
        # cfg["_job_extras"] = True (or MI_JOB_EXTRAS=1), or later with env.enable_job_extras(interval).
        if config.get("_multi_gpu", False) and (bool(config.get("_job_extras", False)) or os.getenv("MI_JOB_EXTRAS") == "1"):
            self._enable_job_extras(int(config.get("_extras_interval", 16)))

    def enable_job_extras(self, interval: int = 16):
        """public switch of the job-wide `extras` (see _enable_job_extras); every rank of the job must call it, and step(), in lockstep"""
        self._enable_job_extras(int(interval))
        return self._job_extras is not None

    def disable_job_extras(self):
        """stop issuing the collective (e.g. before a rank-0-only evaluation loop); the rank's own values return to `extras` with the next step"""
        if self._job_extras is not None:
            self._job_extras.rebase()      # waits for the reduction in flight
        self._job_extras = None

    def _enable_job_extras(self, interval):
        """Sharded run (`make(..., multi_gpu=True)` inside a torch.distributed job, with cfg["_job_extras"]): the task's `extras` statistics become JOB-wide -- SURVEY 8e's
        one collective, a SUM all-reduce of <= 16 numbers on a side stream every `interval` steps (parallel.TaskExtrasReducer); step() publishes
        each completed window (one window behind, never blocking the step stream) and keeps this rank's own values under `*_rank`.
        LOCKSTEP CONTRACT: while this is on, every rank must call step() the same number of times -- a rank that stops stepping (early exit, a
        rank-0-only evaluation) leaves the others waiting in the all-reduce until the process group's timeout; call disable_job_extras() on every
        rank first.  Ant / Humanoid / ... have no task extras of this kind: their episode statistics go through parallel.EpisodeStatsReducer."""
        import torch.distributed as dist
        if self.native_task not in ("AnymalTerrain", "ShadowHand", "AllegroHand") or not (dist.is_available() and dist.is_initialized()):
            return
        from ...parallel import TaskExtrasReducer
        self._job_extras = TaskExtrasReducer(self, interval=interval)
        self._job_window = None

    def _publish_job_extras(self):
        red = self._job_extras
        red.step()
        w = red.poll()
        if w is None or w is self._job_window:
            return
        self._job_window = w
        dev = self.obs_buf.device
        if self.native_task == "AnymalTerrain":
            if "episode_rank" not in self.extras:
                self.extras["episode_rank"] = self.extras.get("episode", {})
            # (a window in which nobody reset carries only the terrain level: the reward terms keep their last values, as the reference's dict does)
            job = dict(self.extras["episode"]) if self.extras.get("episode") is not self.extras["episode_rank"] else {}
            job.update({k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in w.items() if k != "num_resets"})
            self.extras["episode"] = job
        else:
            if "consecutive_successes_rank" not in self.extras:
                self.extras["consecutive_successes_rank"] = self.extras.get("consecutive_successes")
            self.extras["consecutive_successes"] = torch.tensor(w["consecutive_successes"], dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ sim params (vec_task.py:514-562)
    def _parse_sim_params(self, physics_engine: str, config_sim: Dict[str, Any]) -> native.MiSimParams:
        if physics_engine not in ("physx", "flex"):
            raise ValueError(f"Invalid physics engine backend: {physics_engine}")
        if config_sim["up_axis"] not in ("z", "y"):
            raise ValueError(f"Invalid physics up-axis: {config_sim['up_axis']}")
        if config_sim["up_axis"] != "z":
            raise ValueError("only up_axis 'z' is implemented (all five target tasks use z)")
        p = native.MiSimParams()
        p.dt = float(config_sim["dt"])
        p.substeps = int(config_sim.get("substeps", 2))
        g = config_sim.get("gravity", [0.0, 0.0, -9.81])
        for i in range(3):
            p.gravity[i] = float(g[i])
        px = config_sim.get("physx", {})
        p.iters = int(px.get("num_position_iterations", 4)) + int(px.get("num_velocity_iterations", 0))
        p.contact_offset = float(px.get("contact_offset", 0.02))
        p.rest_offset = float(px.get("rest_offset", 0.0))
        p.max_depen_vel = float(px.get("max_depenetration_velocity", 10.0))
        p.erp = float(config_sim.get("erp", 0.5))
        p.plane_mu = float(self.cfg["env"].get("plane", {}).get("staticFriction", 1.0))
        p.ground_z = 0.0
        p.cfm = float(config_sim.get("cfm", 1e-6))
        p.warm = float(config_sim.get("warm_start", 1.0))
        return p

    # ------------------------------------------------------------------ engine creation = create_sim + prepare_sim
    def create_sim(self):
        tp = self._task_params()
        self.engine = native.Engine(self.native_task, self.sim_params, tp, self.num_envs, self.device,
                                    seed=self.engine_seed, env_id_offset=self.rank * self.num_envs)
        self._task_params_struct = tp
        if np.isfinite(self.clip_obs):
            self.engine.set_option("clip_obs", self.clip_obs)
        self.engine.set_option("control_freq_inv", self.control_freq_inv)
        if self.device == "cpu":
            # sim.physx.num_threads (vec_task.py:541; cfg/config.yaml:30 default 4): worker threads of the CPU pipeline
            self.engine.set_option("num_threads", max(1, int(self.cfg["sim"].get("physx", {}).get("num_threads", 4))))
        else:
            self._select_multi_wave()
        self.sim = self.engine  # what the reference calls self.sim

    def _select_multi_wave(self):
        """Launch shape of the physics sub-step: `sim.multi_wave` "auto" (default, native.auto_multi_wave), 0, 16, 32 or 64 envs per workgroup;
        the env var MI_MULTI_WAVE overrides it for A/B runs and profiling."""
        mw = os.environ.get("MI_MULTI_WAVE") or self.cfg["sim"].get("multi_wave", "auto")
        native.select_multi_wave(self.engine, self.native_task, self.num_envs, mw)

    def _task_params(self):
        raise NotImplementedError

    def allocate_buffers(self):
        """vec_task.py:301-324 -- same names/dtypes; storage is the engine arena."""
        t = self.engine.tensors
        self.obs_buf = t["obs_buf"]
        self.states_buf = torch.zeros((self.num_envs, self.num_states), device=self.device, dtype=torch.float)
        self.rew_buf = t["rew_buf"]
        self.reset_buf = t["reset_buf"]
        self.timeout_buf = t["timeout_buf"].view(torch.bool)
        self.progress_buf = t["progress_buf"]
        self.randomize_buf = t["randomize_buf"]
        self._obs_out = t["obs_out"]
        self._obs_ring = (self._obs_out[0], self._obs_out[1])       # the two slots of the clamped-observation ring, as views made once
        self._rl_is_sim = torch.device(self.rl_device) == self.obs_buf.device      # then the `.to(rl_device)` of every output is the tensor itself
        self.extras = {}

    def get_state(self):
        return torch.clamp(self.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    # ------------------------------------------------------------------ the hot path (vec_task.py:360-408)
    def step(self, actions: torch.Tensor) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor, Dict[str, Any]]:
        if getattr(self, "randomize", False):
            # the reference calls this from reset_idx (ant.py:253-255), i.e. on steps where some env resets; resets are
            # in-kernel here, so the (frequency-gated) refresh is evaluated every step
            self.apply_randomizations(self.randomization_params)
        if "actions" in self._torch_noise:               # tasks without in-kernel noise
            actions = self._torch_noise["actions"](actions)
        if actions.device != self.obs_buf.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(device=self.obs_buf.device, dtype=torch.float32).contiguous()
        if actions.shape != (self.num_envs, self.num_actions):
            # the kernel reads num_envs * num_actions floats from a raw pointer: a short tensor would be read out of bounds
            raise ValueError(f"actions must have shape ({self.num_envs}, {self.num_actions}), got {tuple(actions.shape)}")
        # one fused launch: clamp -> pre_physics_step -> simulate x control_freq_inv -> post_physics_step -> timeouts
        self.engine.step(actions)
        self.control_steps += 1
        obs = self._obs_ring[self.engine.last_ring()]
        if "observations" in self._torch_noise:
            self.obs_buf[:] = self._torch_noise["observations"](self.obs_buf)
            obs = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs)
        self._post_step_extras()
        if self._job_extras is not None:
            self._publish_job_extras()
        if self._rl_is_sim:          # (vec_task.py:402-408 moves every output to rl_device: the same device here, so the tensors themselves)
            self.extras["time_outs"] = self.timeout_buf
            self.obs_dict["obs"] = obs
            if self.num_states > 0:
                self.obs_dict["states"] = self.get_state()
            return self.obs_dict, self.rew_buf, self.reset_buf, self.extras
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.obs_dict["obs"] = obs.to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def _post_step_extras(self):
        pass

    def zero_actions(self) -> torch.Tensor:
        return torch.zeros([self.num_envs, self.num_actions], dtype=torch.float32, device=self.rl_device)

    def reset_idx(self, env_ids):
        """reference ant.py:252-279 etc.: re-draw the start state of the given envs (int64 indices)."""
        env_ids = torch.as_tensor(env_ids, device=self.device, dtype=torch.int64).contiguous()
        self.engine.reset_idx(env_ids)

    def reset(self):
        """vec_task.py:426-438: returns the current (clamped) observations; resets happen inside step()."""
        self.obs_dict["obs"] = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict

    def reset_done(self):
        """vec_task.py:440-455"""
        done_env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(done_env_ids) > 0:
            self.reset_idx(done_env_ids)
        self.obs_dict["obs"] = torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict, done_env_ids

    # ------------------------------------------------------------------ domain randomisation (vec_task.py:610-840)
    #: tasks whose step kernels apply observation / action noise themselves (mi_engine_set_noise); the others get torch ops
    KERNEL_NOISE_TASKS = ("Cartpole", "Ant", "Humanoid", "ShadowHand", "AllegroHand")
    #: `actor_params` entries that have a per-env engine parameter (Ant, Humanoid): the `actor_scale` tensor carries one factor per BODY for
    #: rigid_body_properties.mass and one per DOF for each of dof_properties.damping / stiffness / armature (csrc/core/engine.hpp AS_*)
    ACTOR_SCALE_GROUPS = {("rigid_body_properties", "mass"): ("body", 0), ("dof_properties", "damping"): ("dof", 0), ("dof_properties", "stiffness"): ("dof", 1),
                          ("dof_properties", "armature"): ("dof", 2)}

    def _actor_scale_column(self, actor, group, attr):
        """What an `actor_params.<actor>.<group>.<attr>` entry drives in the engine's `actor_scale` tensor, or None:
        an int -- ONE column, one draw per env (the ShadowHand task's override);
        (first column, model values [width]) -- one column per body / dof, one draw per env and element, as the reference samples its
        property arrays (vec_task.py:783-828)."""
        hit = self.ACTOR_SCALE_GROUPS.get((group, attr))
        if hit is None:
            return None
        spec = self._dr_model()
        if hit[0] == "body":
            return 0, np.asarray(spec.mass, np.float64)
        og = np.asarray({0: spec.dof_damping, 1: spec.dof_stiffness, 2: spec.dof_armature}[hit[1]], np.float64)
        return spec.nb + hit[1] * spec.nd, og

    def apply_randomizations(self, dr_params):
        """What the reference's VecTask.apply_randomizations decides, when (vec_task.py:610-648: `frequency` in sim frames for the
        non-env parameters, per env once it has been reset after `frequency` of its own steps), is kept; what it then does is
        different: observation / action noise becomes a parameter block of the step kernels (no torch ops per step), gravity an
        engine option, and the per-actor property structs of PhysX become per-env tensors the sub-step reads
        (friction, mass / damping / stiffness / armature scales), sampled for all due envs at once."""
        from ...utils.dr_utils import Draw, apply_random_gravity
        rand_freq = dr_params.get("frequency", 1)
        self.last_step = int(self.control_steps * max(1, self.control_freq_inv))       # gym.get_frame_count
        due_envs = None                                                                # None: every env (first call)
        if self.first_randomization:
            refresh_global = True
        else:
            refresh_global = (self.last_step - self.last_rand_step) >= rand_freq
            due_envs = torch.logical_and(self.randomize_buf >= rand_freq, self.reset_buf.bool())
            self.randomize_buf[due_envs] = 0
        if refresh_global:
            self.last_rand_step = self.last_step
            # every refresh replaces the noise closures (reference vec_task.py:690, :716: the new dict has no 'corr'), i.e. the correlated
            # offset is re-drawn at the next call: the in-kernel generator gets a new epoch, the torch-op noise forgets its tensor
            self._noise_epoch = getattr(self, "_noise_epoch", -1) + 1
            for which, name in enumerate(("observations", "actions")):
                if name in dr_params:
                    self._configure_noise(which, name, dr_params[name], Draw)
            for attr, prm in dr_params.get("sim_params", {}).items():
                if attr != "gravity":
                    raise NotImplementedError(f"sim_params.{attr} randomisation is not supported by the engine")
                if self.first_randomization:
                    self.original_props = getattr(self, "original_props", {})
                    self.original_props["sim_params"] = {"gravity": [float(self.sim_params.gravity[i]) for i in range(3)]}
                g = apply_random_gravity([float(self.sim_params.gravity[i]) for i in range(3)],
                                         self.original_props["sim_params"]["gravity"], prm, self.last_step)
                for i, key in enumerate(("gravity_x", "gravity_y", "gravity_z")):
                    self.sim_params.gravity[i] = float(g[i])
                    self.engine.set_option(key, float(g[i]))
        if dr_params.get("actor_params"):
            self._apply_actor_params(dr_params["actor_params"], due_envs)
        self.first_randomization = False

    def _configure_noise(self, which, name, prm, Draw):
        """`observations` / `actions` block (vec_task.py:650-718) -> noise parameters.  The schedule blends the white and the
        correlated range exactly like any other randomised parameter (utils/dr_utils.py::Draw)."""
        dist, op = prm["distribution"], prm["operation"]
        if dist not in ("gaussian", "uniform"):
            raise ValueError(f"unsupported noise distribution {dist}")
        white = Draw(dict(prm), self.last_step)
        corr = Draw(dict(prm, range=prm.get("range_correlated", [0.0, 0.0])), self.last_step)
        spec = dict(dist=dist, op=op, a=float(white.a), b=float(white.b), a_corr=float(corr.a), b_corr=float(corr.b))
        epoch = int(getattr(self, "_noise_epoch", 0))
        self.dr_randomizations[name] = dict(spec, in_kernel=self.native_task in self.KERNEL_NOISE_TASKS, epoch=epoch)
        if self.native_task in self.KERNEL_NOISE_TASKS:
            self.engine.set_noise(which, epoch=epoch, **spec)
        else:
            self._torch_noise[name] = _TorchNoise(spec, None)

    def _apply_actor_params(self, actor_params, due_envs):
        """`actor_params` (vec_task.py:752-828).  The reference walks every env's PhysX property structs in Python; here each
        supported entry is one vectorised draw over the due envs into a per-env tensor the sub-step kernel reads:
          rigid_shape_properties.friction                     -> `friction` (Ant, Humanoid, Anymal; ShadowHand: mean of hand and object);
          rigid_body_properties.mass, dof_properties.damping / stiffness / armature -> `actor_scale` (Ant, Humanoid; Anymal: masses and the
          position drives' gains, which are its dofs' stiffness / damping, tasks/anymal.py): one factor per env
          and BODY resp. DOF, drawn per element like the reference draws its property arrays (`scaling`: the sample itself; `additive`:
          relative to the element's model value);
          ShadowHand (`_actor_scale_column` of the task): hand mass / dof damping / dof stiffness (the drives' kp) / tendon stiffness /
          tendon damping, object mass and `scale` (vec_task.py:760-775) -> columns of its `actor_scale`; dof_properties.lower / upper ->
          one shift per joint and env in `dof_limit_shift`.
        Entries without an engine parameter (restitution, colours, ...) are named once in a warning."""
        from ...utils.dr_utils import apply_random_samples_array
        t = self.engine.tensors
        fr = t.get("friction") if self.native_task in ("Ant", "Humanoid", "ShadowHand", "AllegroHand", "Anymal") else None
        scales = t.get("actor_scale")
        pair = self.native_task in ("ShadowHand", "AllegroHand")       # one contact coefficient per env = mean of the two actors' shape friction
        if pair and not hasattr(self, "_dr_actor_friction"):
            self._dr_actor_friction = {}
        ids = torch.arange(self.num_envs, device=self.device) if due_envs is None else torch.nonzero(due_envs, as_tuple=False).squeeze(-1)
        if scales is not None and not getattr(self, "_actor_tensors_on", False):
            self._enable_actor_tensors()
        base_mu = float(getattr(self, "model_shape_friction", 1.0))
        skipped = []
        for actor, groups in actor_params.items():
            for group, attrs in groups.items():
                if group == "color":
                    continue                                                             # no renderer
                if not isinstance(attrs, dict):
                    skipped.append(f"{actor}.{group}")
                    continue
                if group == "scale":                                                     # the entry is the parameter block itself (:760-775)
                    attrs = {"scale": attrs}
                for attr, prm in attrs.items():
                    col = self._actor_scale_column(actor, group, attr) if scales is not None else None
                    is_friction = group == "rigid_shape_properties" and attr == "friction" and fr is not None
                    shift = t.get("dof_limit_shift") if (group == "dof_properties" and attr in ("lower", "upper")) else None
                    if col is None and not is_friction and shift is None:
                        skipped.append(f"{actor}.{group}" if group == "scale" else f"{actor}.{group}.{attr}")
                        continue
                    if (prm.get("setup_only", False) and not self.first_randomization) or len(ids) == 0:
                        continue
                    if shift is not None:
                        # one draw per joint and env, like the reference's per-element sampling of the dof property array; the engine
                        # keeps the model's limits and adds the shift (`additive`) or the scaled-minus-original value (`scaling`)
                        nd = shift.shape[1] // 2
                        base = np.asarray(self._dr_model().dof_lower if attr == "lower" else self._dr_model().dof_upper, np.float64)
                        og = {"v": np.tile(base, (len(ids), 1))}
                        vals = np.asarray(apply_random_samples_array({"v": og["v"].copy()}, og, "v", prm, self.last_step), np.float64)
                        c0 = 0 if attr == "lower" else nd
                        shift[ids, c0:c0 + nd] = torch.as_tensor((vals - og["v"]).astype(np.float32), device=self.device)
                    elif is_friction:
                        og = {"v": np.full(len(ids), base_mu)}
                        vals = apply_random_samples_array({"v": og["v"].copy()}, og, "v", prm, self.last_step)
                        vals_t = torch.as_tensor(np.asarray(vals, np.float32), device=self.device)
                        if pair:
                            mine = self._dr_actor_friction.setdefault(actor, torch.full((self.num_envs,), base_mu, device=self.device))
                            mine[ids] = vals_t
                            others = [v for k, v in self._dr_actor_friction.items() if k != actor]
                            other = others[0] if others else torch.full_like(mine, base_mu)
                            fr[ids] = 0.5 * (mine[ids] + other[ids])
                        else:
                            fr[ids] = vals_t
                    elif isinstance(col, tuple):
                        # one draw per env and body / dof; the engine multiplies the model's own value, so an element whose model value is
                        # zero (an Ant joint has no stiffness) keeps factor 1: `scaling` leaves it zero anyway, `additive` cannot be expressed
                        dst = scales
                        if len(col) == 3:      # (tensor name, first column, model values): a tensor of its own (the ShadowHand's per-body link masses)
                            dst, col = t[col[0]], col[1:]
                            self._on_body_tensor_written()
                        c0, base = col
                        og = {"v": np.tile(base, (len(ids), 1))}
                        vals = np.asarray(apply_random_samples_array({"v": og["v"].copy()}, og, "v", prm, self.last_step), np.float64)
                        factor = np.where(base > 0, np.clip(vals / np.where(base > 0, base, 1.0), 0.05, 20.0), 1.0)
                        if prm.get("operation") == "additive" and (base <= 0).any() and self.first_randomization:
                            skipped.append(f"{actor}.{group}.{attr} (additive on model values of zero)")
                        dst[ids, c0:c0 + len(base)] = torch.as_tensor(factor.astype(np.float32), device=self.device)
                    else:
                        ref_val = self._actor_reference_value(group, attr, actor)
                        og = {"v": np.full(len(ids), ref_val)}
                        vals = np.asarray(apply_random_samples_array({"v": og["v"].copy()}, og, "v", prm, self.last_step), np.float64)
                        factor = np.clip(vals / ref_val, 0.05, 20.0) if ref_val > 0 else np.ones(len(ids))
                        scales[ids, col] = torch.as_tensor(factor.astype(np.float32), device=self.device)
        if skipped and self.first_randomization:
            import warnings
            warnings.warn("actor_params entries without an engine counterpart are skipped (reference vec_task.py:752-828): " + ", ".join(skipped))

    def _on_body_tensor_written(self):
        """hook of the tasks whose kernels read a per-body factor tensor only when told to (ShadowHand: option "hand_body_mass")"""

    def _enable_actor_tensors(self):
        """the sub-step kernels of Ant / Humanoid read `actor_scale` / `dof_limit_shift` only when told to (option "actor_tensors")"""
        set_option = getattr(self.engine, "set_option", None)
        if set_option is not None:
            set_option("actor_tensors", 1)
        self._actor_tensors_on = True

    def _dr_model(self):
        spec = getattr(self, "_dr_spec", None)
        if spec is None:
            from ...registry import load_model
            spec = self._dr_spec = load_model(getattr(self, "model_name", self.native_task.lower()))
        return spec

    def _actor_reference_value(self, group, attr, actor=None):
        """the model's own (mean) value a randomised actor property is measured against"""
        spec = self._dr_model()
        arr = {"mass": spec.mass, "damping": spec.dof_damping, "stiffness": spec.dof_stiffness, "armature": spec.dof_armature}[attr]
        return float(np.mean(arr)) if len(arr) else 0.0

    def _viewer_extras(self, env):
        """(centres [k, 3], radii [k], colours [k, 3]) of what render() draws besides the articulation's bodies, or None"""
        return None

    def render(self, mode="rgb_array"):
        """Reference vec_task.py:457-512 draws Isaac Gym's OpenGL viewer and, for `rgb_array`, grabs the virtual display.  Here the frame is
        rasterised on the host from the rigid-body state tensor of ONE env (`env.viewerEnv`, default 0) by utils/viewer.py: returned as a
        uint8 [H, W, 3] array for mode "rgb_array" (whenever `virtual_screen_capture` or `force_render` asked for frames, or the call is
        made explicitly with that mode), written to `record_frames_dir/frame_<control step>.png` while `record_frames` is set (the viewer's
        `record_frames` key event, :471-472,503-507).  Headless runs that never call it pay nothing: the body states are refreshed here."""
        if mode not in ("rgb_array", "human"):
            raise ValueError(f"render mode {mode!r}: 'human' or 'rgb_array'")
        if mode == "human" and not self.record_frames:
            return None                                   # no window to draw into
        from ...utils.viewer import SoftwareViewer, write_png
        if getattr(self, "_soft_viewer", None) is None:
            vc = self.cfg["env"].get("viewer", {}) or {}
            self._soft_viewer = SoftwareViewer(width=vc.get("width", 640), height=vc.get("height", 480))
            self._viewer_env = int(self.cfg["env"].get("viewerEnv", 0))
        spec = self._dr_model()
        self.engine.refresh_rigid_body_states()
        bs = self.engine.tensors["rigid_body_state"][self._viewer_env].detach().cpu().numpy().astype(np.float64)
        centres, radii, colours = self._soft_viewer.spheres_of(spec, bs)
        if spec.sph_body is None or len(spec.sph_body) == 0:
            # manipulators carry their collision samples against the OBJECT (models/*_extras.json), not against the ground: draw those
            from ...registry import load_extras
            try:
                ex = load_extras(getattr(self, "model_name", self.native_task.lower()))
                shown = type("S", (), dict(sph_body=ex["os_body"], sph_pos=ex["os_pos"], sph_rad=ex["os_rad"], parent=spec.parent))
                centres, radii, colours = self._soft_viewer.spheres_of(shown, bs)
            except (KeyError, FileNotFoundError, TypeError):
                pass
        extra = self._viewer_extras(self._viewer_env)         # free bodies that are not part of the articulation (a hand's object and goal)
        if extra is not None:
            centres, radii, colours = np.concatenate([centres, extra[0]]), np.concatenate([radii, extra[1]]), np.concatenate([colours, extra[2]])
        if hasattr(self, "env_origins") and "Terrain" in type(self).__name__:
            self._soft_viewer.ground_z = float(self.env_origins[self._viewer_env, 2])
        img = self._soft_viewer.draw(centres, radii, colours, focus=bs[0, :3])
        if self.record_frames:
            d = getattr(self, "record_frames_dir", os.path.join("recorded_frames", type(self).__name__))
            os.makedirs(d, exist_ok=True)
            write_png(os.path.join(d, f"frame_{self.control_steps}.png"), img)
        return img if mode == "rgb_array" else None

    # ------------------------------------------------------------------ physics-state checkpointing
    def get_env_state(self):
        """Full simulator + task state (the reference never checkpoints physics, vec_task.py:196-204): the arena, the step counters and
        the domain-randomisation state that lives outside the arena (gravity, noise parameters and epoch, whether the sub-step reads the
        actor tensors, the torch-op noise's correlated tensors, the randomisation schedule)."""
        g = [float(self.sim_params.gravity[i]) for i in range(3)]
        return {"arena": self.engine.arena.clone(), "abi_version": native.MI_ABI_VERSION, "arena_bytes": int(self.engine.arena.numel()),
                "task": self.native_task, "num_envs": int(self.num_envs),
                "control_steps": self.control_steps, "engine_steps": self.engine.get_option("steps"),
                "gravity": g, "noise": {k: dict(v) for k, v in self.dr_randomizations.items() if isinstance(v, dict) and "dist" in v},
                "noise_epoch": int(getattr(self, "_noise_epoch", -1)), "actor_tensors": bool(getattr(self, "_actor_tensors_on", False)),
                "hand_body_mass": bool(getattr(self, "_hand_body_mass_on", False)),
                "torch_noise_corr": {k: (None if n.corr is None else n.corr.clone()) for k, n in self._torch_noise.items()},
                "last_rand_step": int(getattr(self, "last_rand_step", -1)), "first_randomization": bool(getattr(self, "first_randomization", True)),
                "last_step": int(getattr(self, "last_step", -1))}

    def set_env_state(self, env_state):
        if env_state is None:
            return
        # the arena is raw engine memory: a checkpoint of another layout (ABI version), task or env count must not be copied over it
        abi = env_state.get("abi_version")
        if abi is not None and abi != native.MI_ABI_VERSION:
            raise RuntimeError(f"set_env_state: checkpoint of ABI version {abi}, this engine has {native.MI_ABI_VERSION} (the arena layout differs)")
        if tuple(env_state["arena"].shape) != tuple(self.engine.arena.shape) or env_state.get("task", self.native_task) != self.native_task:
            raise RuntimeError(f"set_env_state: checkpoint of {env_state.get('task', '?')} with {env_state['arena'].numel()} arena bytes does not fit "
                               f"{self.native_task} with {self.engine.arena.numel()}")
        self.engine.arena.copy_(env_state["arena"])
        if self._job_extras is not None:
            self._job_extras.rebase()        # the cumulative sums were just replaced: the next reduction is a new baseline, not a window
        self.control_steps = env_state.get("control_steps", 0)
        # the engine's own step counter drives the observation-ring parity, the AnymalTerrain push schedule and the noise counters
        self.engine.set_option("steps", env_state.get("engine_steps", self.control_steps))
        if "gravity" in env_state:
            for i, key in enumerate(("gravity_x", "gravity_y", "gravity_z")):
                self.sim_params.gravity[i] = float(env_state["gravity"][i])
                self.engine.set_option(key, float(env_state["gravity"][i]))
        self._noise_epoch = int(env_state.get("noise_epoch", getattr(self, "_noise_epoch", -1)))
        for which, name in enumerate(("observations", "actions")):
            spec = env_state.get("noise", {}).get(name)
            if spec is None:
                continue
            self.dr_randomizations[name] = dict(spec)
            keys = {k: spec[k] for k in ("dist", "op", "a", "b", "a_corr", "b_corr")}
            if spec.get("in_kernel"):
                self.engine.set_noise(which, epoch=int(spec.get("epoch", 0)), **keys)
            else:
                corr = env_state.get("torch_noise_corr", {}).get(name)
                self._torch_noise[name] = _TorchNoise(keys, None if corr is None else corr.clone())
        if env_state.get("actor_tensors") and not getattr(self, "_actor_tensors_on", False):
            self._enable_actor_tensors()
        if env_state.get("hand_body_mass"):
            self._on_body_tensor_written()
        for k in ("last_rand_step", "first_randomization", "last_step"):
            if k in env_state:
                setattr(self, k, env_state[k])

    def get_number_of_agents(self):
        return self.num_agents

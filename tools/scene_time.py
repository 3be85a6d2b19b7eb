#!/usr/bin/env python3
"""ms per step of the reference's unmodified franka_cube_stack.py on the HIP backend (its own torch code + gym.simulate of the Franka scene,
csrc/core/scene_engine.hpp) and per gym.simulate() alone: python tools/scene_time.py [num_envs ...]"""
import importlib
import os
import sys
import time
import types

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = os.environ.get("MI_REFERENCE_ROOT") or next((p for p in ("/root/reference", os.path.join(ROOT, "ab", "ref_stage"))
                                                    if os.path.isdir(os.path.join(p, "isaacgymenvs", "tasks"))), "/root/reference")
import isaacgymenvs_amd.shims as shims  # noqa: E402
from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict  # noqa: E402

shims.install(force=True)
for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"), ("isaacgymenvs.utils", "isaacgymenvs/utils"),
                  ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
    mod = types.ModuleType(name)
    mod.__path__ = [os.path.join(REF, rel)]
    sys.modules[name] = mod
task = importlib.import_module("isaacgymenvs.tasks.franka_cube_stack")
vt = importlib.import_module("isaacgymenvs.tasks.base.vec_task")
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
for n in [int(a) for a in sys.argv[1:]] or [4096]:
    vt.EXISTING_SIM = None
    cfg = omegaconf_to_dict(compose("config", overrides=["task=FrankaCubeStack"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
    cfg["env"]["numEnvs"], cfg["sim"]["use_gpu_pipeline"] = n, dev != "cpu"
    if os.environ.get("MI_SCENE_ITERS"):       # "pos,vel": how much of a sub-step its sweeps are (A/B of the solver iteration counts, timing only)
        cfg["sim"]["physx"]["num_position_iterations"], cfg["sim"]["physx"]["num_velocity_iterations"] = [int(x) for x in os.environ["MI_SCENE_ITERS"].split(",")]
    env = task.FrankaCubeStack(cfg, rl_device=dev, sim_device=dev, graphics_device_id=-1, headless=True, virtual_screen_capture=False, force_render=False)
    acts = [2 * torch.rand((n, 7), device=dev) - 1 for _ in range(8)]
    for i in range(30):
        env.step(acts[i % 8])
    if dev != "cpu":
        torch.cuda.synchronize()
    t = time.perf_counter()
    steps = 100
    for i in range(steps):
        env.step(acts[i % 8])
    if dev != "cpu":
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    t = time.perf_counter()
    for i in range(steps):
        env.gym.simulate(env.sim)
    if dev != "cpu":
        torch.cuda.synchronize()
    ds = (time.perf_counter() - t) / steps
    nc = env.sim.engine.tensors["scene_contacts"]
    print(f"FrankaCubeStack@{n} on {dev}: {dt * 1e3:.3f} ms per task step ({n / dt / 1e6:.3f} M env-steps/s), {ds * 1e3:.3f} ms per gym.simulate() (2 sub-steps); "
          f"contacts per env {float(nc[:, 0].float().mean()):.1f}, refused since reset {int(nc[:, 1].sum())}; mean reward {float(env.rew_buf.detach().mean()):.3f}, NaN envs {int(torch.isnan(env.obs_buf).any(dim=1).sum())}")

#!/usr/bin/env python3
"""Ant / Humanoid step time with the `actor_params` tensors off (plain kernels) and on (the Sim<Scaled<M>> kernels of kernels_scaled_*.hip,
all factors 1 -> the same trajectories), alternating in one process on one box.
Usage: tools/actor_tensors_ab.py [Task:num_envs ...]   (set MI_ENGINE_LIB to time another build of the library)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for spec in sys.argv[1:] or ["Ant:4096", "Humanoid:8192"]:
    task, n = spec.split(":")
    n = int(n)
    envs = {}
    for on in (0, 1):
        envs[on] = isaacgymenvs_amd.make(seed=42, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
        envs[on].engine.set_option("actor_tensors", on)
    na = envs[0].num_actions
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.rand((n, na), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for env in envs.values():
        for i in range(300):
            env.step(acts[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        for on, env in envs.items():
            k = 500
            t0 = time.perf_counter()
            for i in range(k):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            print(f"{task}@{n} rep{rep} actor_tensors={on}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s", flush=True)

#!/usr/bin/env python3
"""Step time with all physics sub-steps of a control step in ONE launch (option fused_sub = 1) against one launch per sub-step (0): one
process on one box, alternating.  Usage: tools/fused_sub_ab.py [Task:num_envs ...]   (default Ant:4096 AnymalTerrain:4096)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for spec in sys.argv[1:] or ["Ant:4096", "AnymalTerrain:4096"]:
    task, n = spec.split(":")
    n = int(n)
    envs = {}
    for fs in (0, 1):
        envs[fs] = isaacgymenvs_amd.make(seed=42, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
        envs[fs].engine.set_option("fused_sub", fs)
        if task == "Ant":
            envs[fs].engine.set_option("fused_post", 0)
    na = envs[0].num_actions
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.rand((n, na), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for fs, env in envs.items():
        for i in range(300):
            env.step(acts[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        for fs, env in envs.items():
            k = 2000 if task == "Ant" else 800
            t0 = time.perf_counter()
            for i in range(k):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            print(f"{task}@{n} rep{rep} multi_wave={int(env.engine.get_option('multi_wave'))} fused_sub={fs}: {dt * 1e3:.4f} ms/step, "
                  f"{n / dt / 1e6:.2f} M env-steps/s (pre-generated actions)", flush=True)

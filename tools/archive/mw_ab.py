#!/usr/bin/env python3
"""Step time of the single-wave sub-step against the multi-wave one (core/engine_mw.hpp), same process / same box (GPU).
Usage: tools/mw_ab.py [Ant:4096 AnymalTerrain:4096 Ant:16384 ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

specs = sys.argv[1:] or ["Ant:4096", "AnymalTerrain:4096", "Ant:16384", "Ant:65536", "AnymalTerrain:16384"]
for spec in specs:
    task, n = spec.split(":")
    n = int(n)
    env = isaacgymenvs_amd.make(seed=42, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    na = env.num_actions
    acts = [torch.rand((n, na), device="cuda:0") * 2 - 1 for _ in range(8)]
    modes = [0, 32] + ([16] if os.environ.get("MI_MW_HAS16") == "1" else [])
    for rep in range(2):
        for mw in modes:
            env.engine.set_option("multi_wave", mw)
            for i in range(100):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            k = 600 if n <= 16384 else 150
            t0 = time.perf_counter()
            for i in range(k):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            print(f"{task}@{n} rep{rep} multi_wave={mw:2d}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s, mean reward {env.rew_buf.mean().item():.3f}",
                  flush=True)
    del env

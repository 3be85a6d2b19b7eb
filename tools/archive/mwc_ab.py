#!/usr/bin/env python3
"""Humanoid step time: option multi_wave = 0 (one wave), 2 (round 2: main wave + self-collision helper), 32 (round 3: one limb per wave,
every wave sweeping its own rows, csrc/core/engine_mwc.hpp) in one process on one box, alternating.  Also the episode statistics of
each form (mean reward, reset rate): the solver order differs, the task-level behaviour must not.
Usage: tools/mwc_ab.py [num_envs ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [8192]:
    envs = {}
    for mw in (0, 2, 32):
        envs[mw] = isaacgymenvs_amd.make(seed=42, task="Humanoid", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
        envs[mw].engine.set_option("multi_wave", mw)
    na = 21
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.rand((n, na), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for mw, env in envs.items():
        for i in range(300):
            env.step(acts[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        for mw, env in envs.items():
            st0 = env.engine.tensors["episode_stats"].clone()
            k = 500
            t0 = time.perf_counter()
            for i in range(k):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            st = (env.engine.tensors["episode_stats"] - st0).cpu().tolist()
            print(f"Humanoid@{n} rep{rep} multi_wave={mw:2d}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s | mean reward {st[3] / max(st[4], 1):.4f}, "
                  f"resets per env-step {st[2] / max(st[4], 1):.5f}, mean episode length {st[1] / max(st[2], 1):.1f}", flush=True)

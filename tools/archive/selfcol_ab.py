#!/usr/bin/env python3
"""Humanoid@8192 step time with and without self-collision, one wave per workgroup and with the self-collision phase on a helper wave
(csrc/sc2_kernels.hpp), same process / same box (GPU)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

n = 8192
env = isaacgymenvs_amd.make(seed=42, task="Humanoid", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
acts = [torch.rand((n, 21), device="cuda:0") * 2 - 1 for _ in range(8)]
for rep in range(2):
  for mw in (32, 0):
    env.engine.set_option("multi_wave", mw)
    for on in (1, 0):
        if not on and mw:
            continue
        env.engine.set_option("self_collision", on)
        for i in range(100):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 400
        for i in range(k):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        act = (env.self_contact_impulse[:, :, 0] > 0).sum(1).float().mean().item() if on else 0.0
        print(f"rep{rep} self_collision={on} waves={2 if (mw and on) else 1}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s, loaded self contacts per env {act:.3f}", flush=True)

#!/usr/bin/env python3
"""Do the limb-per-wave and the single-wave Ant sub-step give the same statistics under one closed-loop-free but contact-rich policy
(a crude periodic gait)?  Mean reward / forward speed / resets over many envs and steps; run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402

n, steps = 4096, 800
for task in ("Ant",):
    for mw in (0, 16, 32):
        for amp in (1.0, 0.5):
            env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
            env.engine.set_option("multi_wave", mw)
            g = torch.Generator(device="cuda:0").manual_seed(1)
            phase = torch.rand((n, env.num_actions), device="cuda:0", generator=g) * 6.283
            freq = 0.15 + 0.1 * torch.rand((n, 1), device="cuda:0", generator=g)
            rsum = torch.zeros((), device="cuda:0"); vsum = torch.zeros((), device="cuda:0"); resets = 0
            for t in range(steps):
                a = amp * torch.sin(freq * t + phase)
                obs, rew, reset, _ = env.step(a)
                rsum += rew.mean(); vsum += env.engine.tensors["root_states"][:, 7].mean(); resets += int(reset.sum())
            print(f"{task} mw={mw:2d} amp={amp}: mean reward {float(rsum) / steps:.4f}  mean vx {float(vsum) / steps:.4f}  resets {resets}", flush=True)

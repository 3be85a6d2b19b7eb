#!/usr/bin/env python3
"""rocprofv3 helper: plain step() loop (optionally controlFrequencyInv = k) for per-launch duration analysis."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402

task, n, cfi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = isaacgymenvs_amd.make(seed=1, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
env.engine.set_option("control_freq_inv", cfi)
acts = [torch.rand((n, env.num_actions), device="cuda:0") * 2 - 1 for _ in range(16)]
for i in range(200):
    env.engine.step(acts[i % 16])
torch.cuda.synchronize()

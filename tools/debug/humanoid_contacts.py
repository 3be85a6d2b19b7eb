#!/usr/bin/env python3
"""How many ground contacts does a Humanoid env have (compact store: KMAX = 16 slots per env)?  Counts spheres whose normal
impulse is positive after a step (a lower bound of the slots taken: spheres inside contact_offset without load also take one)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402

n = 4096
env = isaacgymenvs_amd.make(seed=3, task="Humanoid", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
lam = env.engine.tensors["contact_impulse"]
g = torch.Generator(device="cuda:0").manual_seed(1)
hist = torch.zeros(36, device="cuda:0")
for step in range(600):
    env.step(torch.rand((n, 21), device="cuda:0", generator=g) * 2 - 1)
    if step >= 50:
        c = (lam[:, :, 0] > 0).sum(1)
        hist += torch.bincount(c.clamp(0, 35).long(), minlength=36).float()
h = (hist / hist.sum()).cpu().tolist()
print(" ".join(f"{k}:{p:.3f}" for k, p in enumerate(h[:20])), "P(>=12)=%.4f P(>=16)=%.5f" % (sum(h[12:]), sum(h[16:])))

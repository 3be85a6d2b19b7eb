#!/usr/bin/env python3
"""Time the ShadowHand step of the library named by MI_ENGINE_LIB (tools/hand_residency_ab.sh) at 16384 and 8192 envs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for n in (16384, 8192):
    env = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    acts = [torch.rand((n, 20), device="cuda:0") * 2 - 1 for _ in range(8)]
    for i in range(60):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{os.path.basename(os.environ.get('MI_ENGINE_LIB', 'default'))}: ShadowHand@{n} {dt * 1e3:.4f} ms/step", flush=True)
    del env

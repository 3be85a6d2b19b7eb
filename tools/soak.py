#!/usr/bin/env python3
"""Soak run: every task at its configured env count under a uniform random policy for thousands of steps; every observation, reward and
root state must stay finite and inside loose physical bounds.  (GPU.)  Usage: python tools/soak.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
TASKS = [("Cartpole", 512), ("Ant", 4096), ("Humanoid", 8192), ("Anymal", 4096), ("AnymalTerrain", 4096), ("ShadowHand", 16384), ("Quadcopter", 8192),
         ("Ingenuity", 4096), ("BallBalance", 4096)]
bad = 0
for task, n in TASKS:
    env = isaacgymenvs_amd.make(seed=123, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    g = torch.Generator(device="cuda:0").manual_seed(7)
    t0 = time.perf_counter()
    resets, worst = 0, 0.0
    for i in range(steps):
        a = torch.rand((n, env.num_actions), device="cuda:0", generator=g) * 2 - 1
        obs, rew, reset, _ = env.step(a)
        resets += int(reset.sum()) if i % 50 == 0 else 0
        if i % 100 == 99:
            ok = bool(torch.isfinite(obs["obs"]).all()) and bool(torch.isfinite(rew).all())
            t = env.engine.tensors
            ok = ok and bool(torch.isfinite(t["root_states"]).all()) and bool(torch.isfinite(t["dof_state"]).all())
            worst = max(worst, float(t["root_states"][:, 7:13].abs().max()), float(t["dof_state"][..., 1].abs().max()))
            if not ok:
                print(f"{task}: NON-FINITE at step {i}", flush=True)
                bad += 1
                break
    torch.cuda.synchronize()
    print(f"{task}@{n}: {steps} steps in {time.perf_counter() - t0:.1f} s, resets sampled {resets}, largest speed seen {worst:.1f}", flush=True)
    del env
print("SOAK", "FAILED" if bad else "OK")

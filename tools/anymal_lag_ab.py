#!/usr/bin/env python3
"""What AnymalTerrain's lagging dof-state tensor costs (round 5; option dof_state_lag: the PD law's first evaluation, the joint observations and the
reward's joint terms read the tensor of the task's last refresh, anymal_terrain.py:441-455): the SAME engine stepped in alternating blocks with the
option off and on, random-action rollout.  Usage: tools/anymal_lag_ab.py [envs] [steps]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import isaacgymenvs_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = isaacgymenvs_amd.make(seed=42, task="AnymalTerrain", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
g = torch.Generator(device="cuda:0").manual_seed(0)
acts = [2.0 * torch.rand((n, 12), device="cuda:0", generator=g) - 1.0 for _ in range(8)]
for i in range(200):
    env.step(acts[i % 8])
torch.cuda.synchronize()
print(f"AnymalTerrain@{n} multi_wave={int(env.engine.get_option('multi_wave'))} fused_sub={int(env.engine.get_option('fused_sub'))}, {steps} steps per block")
for rep in range(3):
    for lag in (0, 1):
        env.engine.set_option("dof_state_lag", lag)
        for i in range(50):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        d = (env.dof_state_refreshed - env.engine.tensors["dof_state"]).abs().max().item() if lag else 0.0
        print(f"rep {rep} dof_state_lag {lag}: {1e3 * dt / steps:.4f} ms/step  {n * steps / dt / 1e6:.2f} M env-steps/s   max |refreshed - physics| {d:.3f}")

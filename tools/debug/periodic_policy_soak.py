#!/usr/bin/env python3
"""Every task under a periodic full-amplitude policy (per-env random phases and frequencies): is everything still finite, how fast does
it get?  (This kind of policy found the missing velocity clamp that the uniform-random soak never reached.)  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402

DEV = "cuda:0"
for task, n, steps in [("Cartpole", 1024, 600), ("Ant", 4096, 800), ("Humanoid", 4096, 600), ("Anymal", 2048, 600), ("AnymalTerrain", 2048, 600),
                       ("ShadowHand", 4096, 500), ("Quadcopter", 2048, 600), ("Ingenuity", 2048, 800), ("BallBalance", 2048, 600)]:
    env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    g = torch.Generator(device=DEV).manual_seed(1)
    phase = torch.rand((n, env.num_actions), device=DEV, generator=g) * 6.283
    freq = 0.1 + 0.3 * torch.rand((n, 1), device=DEV, generator=g)
    t = env.engine.tensors
    bad_at, vmax, wmax = None, 0.0, 0.0
    for k in range(steps):
        obs, rew, reset, _ = env.step(torch.sin(freq * k + phase))
        if k % 20 == 19:
            ok = torch.isfinite(rew).all() and torch.isfinite(t["root_states"]).all() and torch.isfinite(t["dof_state"]).all() and torch.isfinite(obs["obs"]).all()
            if not bool(ok) and bad_at is None:
                bad_at = k
            vmax = max(vmax, float(torch.nan_to_num(t["dof_state"][..., 1]).abs().max()))
            wmax = max(wmax, float(torch.nan_to_num(t["root_states"][:, 10:13]).norm(dim=1).max()))
    print(f"{task:14s} n={n} steps={steps}: {'NOT FINITE from step ' + str(bad_at) if bad_at is not None else 'finite'}  max |joint speed| {vmax:.1f}  max |root omega| {wmax:.1f}", flush=True)
    del env

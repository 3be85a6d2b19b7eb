#!/usr/bin/env python3
"""Ant step time with post_physics_step fused into the last limb-per-wave sub-step launch (option fused_post = 1, default) against the
separate loco_post_kernel (0), one process on one box, alternating.  Usage: tools/ant_fused_post_ab.py [num_envs ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [4096]:
    envs = {}
    for fp in (0, 1):
        envs[fp] = isaacgymenvs_amd.make(seed=42, task="Ant", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
        envs[fp].engine.set_option("fused_post", fp)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.rand((n, 8), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for fp, env in envs.items():
        for i in range(300):
            env.step(acts[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        for fp, env in envs.items():
            k = 2000
            t0 = time.perf_counter()
            for i in range(k):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            print(f"Ant@{n} rep{rep} multi_wave={int(env.engine.get_option('multi_wave'))} fused_post={fp}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s (pre-generated actions)", flush=True)

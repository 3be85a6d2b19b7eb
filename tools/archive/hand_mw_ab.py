#!/usr/bin/env python3
"""ShadowHand step time: option multi_wave = 0 (one wave per 32 envs, core/hand_engine.hpp) against 32 and 64 (one finger per wave,
core/hand_engine_mw.hpp, 32 / 64 envs per workgroup) in one process on one box, alternating.  Also the episode statistics of each form (mean reward, reset rate,
consecutive successes, contact counts): the solver order differs, the task-level behaviour must not.
Usage: tools/hand_mw_ab.py [num_envs ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [16384]:
    envs = {}
    for mw in (0, 32, 64):
        envs[mw] = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
        envs[mw].engine.set_option("multi_wave", mw)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.rand((n, 20), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for mw, env in envs.items():
        for i in range(200):
            env.step(acts[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        for mw, env in envs.items():
            k = 300
            rew = torch.zeros((), device="cuda:0"); rs = torch.zeros((), device="cuda:0"); nc = torch.zeros((), device="cuda:0")
            t0 = time.perf_counter()
            for i in range(k):
                env.step(acts[i % 8])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k
            for i in range(100):        # statistics outside the timed loop
                _, r, d, _ = env.step(acts[i % 8])
                rew += r.mean(); rs += d.float().mean(); nc += env.engine.tensors["object_contact_count"].float().mean()
            print(f"ShadowHand@{n} rep{rep} multi_wave={mw:2d}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s | mean reward {rew.item() / 100:.4f}, "
                  f"resets per env-step {rs.item() / 100:.5f}, contacts per env {nc.item() / 100:.2f}, dropped {int(env.engine.tensors['object_contact_dropped'].sum())}", flush=True)

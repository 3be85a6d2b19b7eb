#!/usr/bin/env python3
"""ms per VecTask.step() of a task with make()'s default engine options, pre-generated actions (a pool of 8 batches), three repetitions.
Usage: [MI_OPTS=key=value,...] tools/step_time.py Task:num_envs[:steps] ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

for spec in sys.argv[1:] or ["Ant:4096", "Humanoid:8192"]:
    parts = spec.split(":")
    task, n = parts[0], int(parts[1])
    k = int(parts[2]) if len(parts) > 2 else 1000
    env = isaacgymenvs_amd.make(seed=42, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    for kv in os.environ.get("MI_OPTS", "").split(","):      # e.g. MI_OPTS=fused_sub=0,fused_post=1
        if kv:
            env.engine.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.rand((n, env.num_actions), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for i in range(300):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(k):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        print(f"{task}@{n} rep{rep}: {dt * 1e3:.4f} ms/step, {n / dt / 1e6:.2f} M env-steps/s (pre-generated actions)", flush=True)
    del env

#!/usr/bin/env python3
"""Is the README-protocol loop of Ant@4096 GPU-bound or host-bound?  Host time to ENQUEUE K steps (actions drawn by torch.rand inside the loop, as
bench.py does) against the time until the GPU has finished them."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import isaacgymenvs_amd  # noqa: E402

n = 4096
env = isaacgymenvs_amd.make(seed=42, task="Ant", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
g = torch.Generator(device="cuda:0").manual_seed(1)
for K in (20, 200, 2000):
    for rep in range(3):
        for _ in range(100):
            env.step(2.0 * torch.rand((n, 8), device="cuda:0", generator=g) - 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            env.step(2.0 * torch.rand((n, 8), device="cuda:0", generator=g) - 1.0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"K={K}: host enqueue {1e6 * (t1 - t0) / K:.1f} us/step, until the GPU is done {1e6 * (t2 - t0) / K:.1f} us/step", flush=True)
# the pieces of the host side
a = 2.0 * torch.rand((n, 8), device="cuda:0", generator=g) - 1.0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    a = 2.0 * torch.rand((n, 8), device="cuda:0", generator=g) - 1.0
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"actions only: host {1e6 * (t1 - t0) / 2000:.1f} us/step")
t0 = time.perf_counter()
for _ in range(2000):
    env.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"env.step only: host {1e6 * (t1 - t0) / 2000:.1f} us/step, GPU done {1e6 * (t2 - t0) / 2000:.1f} us/step")

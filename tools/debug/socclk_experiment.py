#!/usr/bin/env python3
"""On a box of the slow kind (socclk asleep under load, profiles/r4z_box_kinds.txt): does anything the process can do wake the SOC clock?
Times ShadowHand@16384 stepping (a) as it is, (b) with a side stream that keeps the copy engines busy (device-to-device and pinned
host-to-device copies), printing rocm-smi's socclk next to each."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import isaacgymenvs_amd  # noqa: E402


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    return " | ".join(ln.split(":", 1)[-1].strip() for ln in out.splitlines() if ("socclk" in ln or "sclk" in ln or "Power (W)" in ln) and "GPU[" in ln)


n = 16384
env = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
g = torch.Generator(device="cuda:0").manual_seed(1)
acts = [torch.rand((n, env.num_actions), device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]


def run(k, label, side=None):
    for i in range(300):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    res = {}
    th = threading.Thread(target=lambda: (time.sleep(1.0), res.setdefault("smi", smi())))
    th.start()
    t0 = time.perf_counter()
    for i in range(k):
        env.step(acts[i % 8])
        if side is not None and i % 4 == 0:
            side()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    th.join()
    print(f"{label}: {dt * 1e3:.4f} ms/step   [{res.get('smi')}]", flush=True)


run(3000, "plain (short)")
if "--always" not in sys.argv:
    t0 = time.perf_counter()
    for i in range(2000):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    if (time.perf_counter() - t0) / 2000 < 0.25e-3:
        print("a box of the fast kind: nothing to learn here")
        sys.exit(0)
run(12000, "plain")
s2 = torch.cuda.Stream()
a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
b = torch.empty_like(a)
h = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()


def d2d():
    with torch.cuda.stream(s2):
        b.copy_(a, non_blocking=True)


def h2d():
    with torch.cuda.stream(s2):
        a[:h.numel()].copy_(h, non_blocking=True)


run(12000, "with device-to-device copies on a side stream", d2d)
run(12000, "with pinned host-to-device copies on a side stream", h2d)
run(12000, "plain again")

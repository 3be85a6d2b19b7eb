#!/usr/bin/env python3
"""(python tools/debug/phase_timing_live.py hand: the same for the ShadowHand sub-step, see the end of the file.)
Per-phase time of the Humanoid sub-step on a LIVE rollout (random actions, self-collision on / off), from s_memtime stamps.
Needs a library whose kernels_humanoid.hip was built with -DMI_TIMING (tools/debug/build_timing_variant.sh -> ab/lib_timing.so):
    MI_ENGINE_LIB=$PWD/ab/lib_timing.so python tools/debug/phase_timing_live.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd import native  # noqa: E402

HAND = len(sys.argv) > 1 and sys.argv[1] == "hand"
if HAND:
    n = int(os.environ.get("N", 16384))
    env = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    L = native.lib()
    wg = (n + 31) // 32
    buf = torch.zeros((wg + 8) * 16, dtype=torch.int64, device="cuda:0")
    L.mi_debug_set_tstamp_hand.argtypes = [C.c_void_p]
    assert L.mi_debug_set_tstamp_hand(C.c_void_p(buf.data_ptr())) == 0
    names = ["tree pass", "rhs + factor + w (as far as the compiler leaves it between the stamps)", "limit rows", "narrow phase + contact rows",
             "warm start (limit rows)", "sweeps", "back-substitution + outputs", "integrate"]
    for i in range(150):
        env.step(torch.rand((n, 20), device="cuda:0") * 2 - 1)
    acc = torch.zeros(8, dtype=torch.float64)
    reps = 40
    for i in range(reps):
        env.step(torch.rand((n, 20), device="cuda:0") * 2 - 1)
        torch.cuda.synchronize()
        st = buf.view(wg + 8, 16)[:wg].cpu().double()
        acc += torch.stack([st[:, k + 1] - st[:, k] for k in range(8)], 1).mean(0)
    acc /= reps
    tot = float(acc.sum())
    print("ShadowHand@%d sub-step phases: " % n + ", ".join(f"{nm} {100 * float(acc[k]) / tot:.1f} %" for k, nm in enumerate(names)), flush=True)
    sys.exit(0)
os.environ["MI_MULTI_WAVE"] = "0"          # the stamps are in the one-wave kernel
n = int(os.environ.get("N", 8192))
env = isaacgymenvs_amd.make(seed=42, task="Humanoid", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
L = native.lib()
wg = (n + 31) // 32 + 8
buf = torch.zeros(wg * 16, dtype=torch.int64, device="cuda:0")
L.mi_debug_set_tstamp.argtypes = [C.c_void_p]
assert L.mi_debug_set_tstamp(C.c_void_p(buf.data_ptr())) == 0
names = ["stage warm", "tree pass", "factor + w", "limit rows", "contact rows (ground + self)", "warm apply", "PGS", "finish"]
for on in (1, 0):
    env.engine.set_option("self_collision", on)
    for i in range(150):
        env.step(torch.rand((n, 21), device="cuda:0") * 2 - 1)
    torch.cuda.synchronize()
    acc = torch.zeros(10, dtype=torch.float64)
    reps = 40
    for i in range(reps):
        env.step(torch.rand((n, 21), device="cuda:0") * 2 - 1)
        torch.cuda.synchronize()
        st = buf.view(wg, 16)[: (n + 31) // 32].cpu().double()
        d = torch.stack([st[:, k + 1] - st[:, k] for k in range(8)] + [st[:, 9] - st[:, 4], st[:, 5] - st[:, 9]], 1)   # ticks of 10 ns
        acc += d.mean(0)
    acc /= reps
    print(f"self_collision={on}: " + ", ".join(f"{nm} {acc[k] / 100:.1f} us" for k, nm in enumerate(names)) +
          f" | ground rows {acc[8] / 100:.1f} us, self-collision phase {acc[9] / 100:.1f} us | sum {acc[:8].sum() / 100:.1f} us", flush=True)

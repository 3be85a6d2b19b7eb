#!/usr/bin/env python3
"""Per-role phase times of the Humanoid limb-per-wave sub-step (csrc/core/engine_mwc.hpp) on a live random-action rollout, from
s_memtime stamps.  Needs ab/lib_timing_mwc.so = the library with kernels_humanoid_mwc.hip rebuilt with -DMI_TIMING:
    tools/debug/build_timing_mwc.sh && MI_ENGINE_LIB=$PWD/ab/lib_timing_mwc.so python tools/debug/mwc_phases.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd import native  # noqa: E402

n = int(os.environ.get("N", 8192))
env = isaacgymenvs_amd.make(seed=42, task="Humanoid", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
assert int(env.engine.get_option("multi_wave")) == 32
L = native.lib()
wg = ((n + 63) // 64 + 7) // 8 * 8 * 2
buf = torch.zeros(wg * 4 * 32, dtype=torch.int64, device="cuda:0")
L.mi_debug_set_tstamp_mwc.argtypes = [C.c_void_p]
assert L.mi_debug_set_tstamp_mwc(C.c_void_p(buf.data_ptr())) == 0
names = ["P1 tree+limb factor", "wait B1", "P2 trunk + P3 own rows", "(pair role) FK + narrow phase", "wait B2", "S1 side a", "wait B3", "S2 side b", "wait B4",
         "pair warm start", "wait B5", "P4 sweeps (incl. 4 barriers)", "P5 outputs"]
for on in (1, 0):
    env.engine.set_option("self_collision", on)
    for i in range(200):
        env.step(torch.rand((n, 21), device="cuda:0") * 2 - 1)
    torch.cuda.synchronize()
    acc = torch.zeros((4, 13), dtype=torch.float64)
    acc2 = torch.zeros((4, 6), dtype=torch.float64)
    tot = torch.zeros(4, dtype=torch.float64)
    reps = 30
    nwg = (n + 31) // 32
    for i in range(reps):
        env.step(torch.rand((n, 21), device="cuda:0") * 2 - 1)
        torch.cuda.synchronize()
        st = buf.view(wg, 4, 32)[:nwg].cpu().double()
        if not on:      # without self-collision stamps 6 .. 9 are not taken: carry 5 forward
            for k in (6, 7, 8, 9):
                st[:, :, k] = st[:, :, 5]
        acc += torch.stack([st[:, :, k + 1] - st[:, :, k] for k in range(13)], 2).mean(0)
        tot += (st[:, :, 13] - st[:, :, 0]).mean(0)
        acc2 += torch.stack([st[:, :, k + 1] - st[:, :, k] for k in range(15, 21)], 2).mean(0)     # the LAST sweep
    acc /= reps; tot /= reps; acc2 /= reps
    print(f"Humanoid@{n} self_collision={on}: sub-step phases per role, us (100 MHz s_memtime ticks / 100)")
    print("%-36s" % "phase" + "".join(f"   role {r}" for r in range(4)))
    for k, nm in enumerate(names):
        print("%-36s" % nm + "".join(f" {acc[r, k] / 100:8.2f}" for r in range(4)))
    print("%-36s" % "total" + "".join(f" {tot[r] / 100:8.2f}" for r in range(4)), flush=True)
    for k, nm in enumerate(["  last sweep: weights + views", "  last sweep: own limit rows", "  last sweep: own ground slots", "  last sweep: publish + pair block", "  last sweep: wait barrier", "  last sweep: sum exchange"]):
        print("%-36s" % nm + "".join(f" {acc2[r, k] / 100:8.2f}" for r in range(4)))

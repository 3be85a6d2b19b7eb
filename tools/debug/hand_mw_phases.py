#!/usr/bin/env python3
"""Per-role phase times of the ShadowHand finger-per-wave sub-step (csrc/core/hand_engine_mw.hpp) on a live random-action rollout, from
s_memtime stamps.  Needs ab/lib_timing_hmw.so = the library with kernels_shadow_hand_mw.hip rebuilt with -DMI_TIMING:
    tools/debug/build_timing_hmw.sh && MI_ENGINE_LIB=$PWD/ab/lib_timing_hmw.so python tools/debug/hand_mw_phases.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd import native  # noqa: E402

n = int(os.environ.get("N", 16384))
env = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
if os.environ.get("MW"):
    env.engine.set_option("multi_wave", int(os.environ["MW"]))
E = int(env.engine.get_option("multi_wave"))
assert E in (32, 64)
L = native.lib()
wg = ((n + 63) // 64 + 7) // 8 * 8 * 2
buf = torch.zeros(wg * 4 * 16, dtype=torch.int64, device="cuda:0")
L.mi_debug_set_tstamp_hmw.argtypes = [C.c_void_p]
assert L.mi_debug_set_tstamp_hmw(C.c_void_p(buf.data_ptr())) == 0
names = ["P1 tree pass + limb factor", "wait B1", "P2 trunk up + wrist factor + object", "wait B1b", "P3 own limit rows", "P3 own contacts (narrow + rows)",
         "wait B2", "P4 sweeps (incl. their barriers)", "P5 outputs + integration"]
for i in range(300):
    env.step(torch.rand((n, 20), device="cuda:0") * 2 - 1)
torch.cuda.synchronize()
acc = torch.zeros((4, 9), dtype=torch.float64)
tot = torch.zeros(4, dtype=torch.float64)
reps = 30
nwg = (n + E - 1) // E
for i in range(reps):
    env.step(torch.rand((n, 20), device="cuda:0") * 2 - 1)
    torch.cuda.synchronize()
    st = buf.view(wg, 4, 16)[:nwg].cpu().double()
    acc += torch.stack([st[:, :, k + 1] - st[:, :, k] for k in range(9)], 2).mean(0)
    tot += (st[:, :, 9] - st[:, :, 0]).mean(0)
acc /= reps; tot /= reps
print(f"ShadowHand@{n}, {E} envs per workgroup: finger-per-wave sub-step, phases per role, us (100 MHz s_memtime ticks / 100); roles: 0 little finger, 1 thumb, 2 first + ring finger, 3 palm + middle finger")
print("%-40s" % "phase" + "".join(f"   role {r}" for r in range(4)))
for k, nm in enumerate(names):
    print("%-40s" % nm + "".join(f" {acc[r, k] / 100:8.2f}" for r in range(4)))
print("%-40s" % "total" + "".join(f" {tot[r] / 100:8.2f}" for r in range(4)), flush=True)

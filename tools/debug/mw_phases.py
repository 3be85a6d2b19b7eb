#!/usr/bin/env python3
"""Per-role phase times of the Ant's one-launch control step (csrc/mw_kernels.hpp substep_mw_fused_post_kernel: two sub-steps of csrc/core/engine_mw.hpp
substep_role + post_physics_step on the role waves) on a live random-action rollout, from s_memtime stamps (100 MHz).  Needs ab/lib_timing_mw.so = the
library with kernels_mw_ant.hip rebuilt with -DMI_TIMING:
    tools/debug/build_timing_mw.sh && MI_ENGINE_LIB=$PWD/ab/lib_timing_mw.so python tools/debug/mw_phases.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd import native  # noqa: E402

PHASES = [("load state, warm start -> LDS", 30, 0), ("s0 P1 tree + limb factor", 0, 1), ("s0 wait B1", 1, 2), ("s0 P2 trunk + P3 own rows", 2, 3), ("s0 wait B2", 3, 4),
          ("s0 P4 sweeps (incl. their barriers)", 4, 5), ("s0 P5 outputs, integration", 5, 6), ("root exchange + efforts of s1", 6, 8),
          ("s1 P1 tree + limb factor", 8, 9), ("s1 wait B1", 9, 10), ("s1 P2 trunk + P3 own rows", 10, 11), ("s1 wait B2", 11, 12),
          ("s1 P4 sweeps (incl. their barriers)", 12, 13), ("s1 P5 outputs, integration", 13, 14), ("post_physics_step on the role waves", 14, 31)]
for n in [int(x) for x in os.environ.get("N", "4096,1024").split(",")]:
    env = isaacgymenvs_amd.make(seed=42, task="Ant", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    assert int(env.engine.get_option("fused_sub")) == 1 and int(env.engine.get_option("fused_post")) == 1
    per_wg = int(env.engine.get_option("multi_wave"))
    L = native.lib()
    nwg = (n + per_wg - 1) // per_wg
    wg = (nwg + 7) // 8 * 8 + 64
    buf = torch.zeros(wg * 4 * 32, dtype=torch.int64, device="cuda:0")
    L.mi_debug_set_tstamp_mw.argtypes = [C.c_void_p]
    assert L.mi_debug_set_tstamp_mw(C.c_void_p(buf.data_ptr())) == 0
    for i in range(300):
        env.step(torch.rand((n, 8), device="cuda:0") * 2 - 1)
    torch.cuda.synchronize()
    reps = 40
    acc = torch.zeros((4, len(PHASES)), dtype=torch.float64)
    tot = torch.zeros(4, dtype=torch.float64)
    for i in range(reps):
        env.step(torch.rand((n, 8), device="cuda:0") * 2 - 1)
        torch.cuda.synchronize()
        st = buf.view(wg, 4, 32).cpu().double()
        live = st[:, 0, 30] > 0                                  # the workgroups that ran (the block -> env mapping is XCD-aware: not the first nwg)
        st = st[live]
        for k, (_, a, b) in enumerate(PHASES):
            acc[:, k] += (st[:, :, b] - st[:, :, a]).mean(0) / 100.0        # us
        tot += (st[:, :, 31] - st[:, :, 30]).mean(0) / 100.0
        buf.zero_()
    acc /= reps; tot /= reps
    print(f"Ant@{n} ({per_wg} envs per workgroup, {int(live.sum())} workgroups): one control step in one launch, phases per role, us (mean over workgroups and {reps} steps)")
    print(f"{'phase':42s}" + "".join(f"{'role ' + str(r):>9s}" for r in range(4)))
    for k, (name, _, _) in enumerate(PHASES):
        print(f"{name:42s}" + "".join(f"{acc[r, k]:9.2f}" for r in range(4)))
    print(f"{'role entered -> post step done':42s}" + "".join(f"{tot[r]:9.2f}" for r in range(4)))
    print()
    del env

#!/usr/bin/env python3
"""us per gym.simulate() of the Articulation task's stock robot (mjcf/amp_humanoid.xml, 28 position drives) on the HIP backend:
python tools/articulation_time.py [num_envs ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isaacgymenvs_amd import native  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [4096]:
    sp = native.MiSimParams(dt=1 / 60.0, substeps=2, iters=4, contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0, erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)
    sp.gravity[2] = -9.81
    tp = native.MiArticulationParams()
    for d in range(28):
        tp.kp[d], tp.kd[d] = 300.0, 30.0
    tp.init_root[2], tp.init_root[6] = 0.89, 1.0
    eng = native.Engine("Articulation", sp, tp, n, "cuda:0", seed=1)
    tg = [(torch.rand((n, 28), device="cuda:0") - 0.5) for _ in range(8)]
    for rep in range(3):
        for i in range(20):
            eng.tensors["dof_position_targets"].copy_(tg[i % 8])
            eng.simulate()
        torch.cuda.synchronize()
        t = time.perf_counter()
        steps = 200
        for i in range(steps):
            eng.tensors["dof_position_targets"].copy_(tg[i % 8])
            eng.simulate()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        print(f"Articulation(amp_humanoid)@{n} rep{rep}: {dt * 1e3:.4f} ms per simulate() (2 sub-steps), {n / dt / 1e6:.2f} M env-steps/s")

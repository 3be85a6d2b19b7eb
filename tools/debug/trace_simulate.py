#!/usr/bin/env python3
"""rocprofv3 helper: run gym.simulate()-only launches (stored efforts, no pre/post kernels) so that the kernel trace shows whether
the first sub-step launch of a control step is slower because of the action path / the neighbouring post kernel or by itself.
    rocprofv3 --kernel-trace --output-format csv -d out -o sim -- python tools/debug/trace_simulate.py Humanoid 8192"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd  # noqa: E402

task, n = sys.argv[1], int(sys.argv[2])
env = isaacgymenvs_amd.make(seed=1, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
a = torch.rand((n, env.num_actions), device="cuda:0") * 2 - 1
for _ in range(50):
    env.step(a)
torch.cuda.synchronize()
for _ in range(100):
    env.engine.simulate()
torch.cuda.synchronize()

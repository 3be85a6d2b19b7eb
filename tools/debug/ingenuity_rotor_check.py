#!/usr/bin/env python3
"""Ingenuity kernels against the oracle env under FULL-range uniform actions (the parity test uses near-hover thrusts): first step at which
they part, and which quantity."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd.registry import load_model, sensor_bodies  # noqa: E402
from oracle.tasks import OracleIngenuityEnv  # noqa: E402
from test_gpu_parity import _sim_dict  # noqa: E402

n, seed = 128, 29
env = isaacgymenvs_amd.make(seed=seed, task="Ingenuity", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
orc = OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, precision="f64")
g = torch.Generator(device="cpu").manual_seed(4)
for step in range(400):
    a = torch.rand((n, 6), generator=g) * 2 - 1
    env.step(a.to("cuda:0"))
    orc.step(a.numpy())
    t = env.engine.tensors
    dq = np.abs(t["dof_state"][..., 0].cpu().numpy() - orc.eng.q).max(0)
    dv = np.abs(t["dof_state"][..., 1].cpu().numpy() - orc.eng.qd).max(0)
    dr = np.abs(t["root_states"].cpu().numpy() - orc.eng.root).max(0)
    if step < 12 or step % 40 == 0 or dv.max() > 1.0:
        print(step, "dq", dq.round(5), "dv", dv.round(4), "droot pos %.1e quat %.1e lin %.1e ang %.1e" % (dr[0:3].max(), dr[3:7].max(), dr[7:10].max(), dr[10:13].max()),
              "| gpu max |w| %.2f" % float(t["root_states"][:, 10:13].abs().max()), "orc %.2f" % np.abs(orc.eng.root[:, 10:13]).max(), flush=True)
    if dv.max() > 50:
        k = int(np.abs(t["dof_state"][..., 1].cpu().numpy() - orc.eng.qd).max(1).argmax())
        print("env", k, "gpu dof", t["dof_state"][k].cpu().numpy().round(3).tolist(), "orc q", orc.eng.q[k].round(4), "qd", orc.eng.qd[k].round(3),
              "\n gpu root", t["root_states"][k].cpu().numpy().round(3), "\n orc root", orc.eng.root[k].round(3), "progress", int(env.progress_buf[k]), int(orc.progress_buf[k]),
              "limit_impulse", t["limit_impulse"][k].cpu().numpy(), "orc", orc.eng.lam[k])
        break

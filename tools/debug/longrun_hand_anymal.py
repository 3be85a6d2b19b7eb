#!/usr/bin/env python3
"""One-off long-rollout statistics (as tests/test_gpu_longrun.py) for the two tasks whose oracles are too slow for the test suite:
AnymalTerrain (height field) and ShadowHand (numpy oracle, a few dozen envs).  GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd.registry import load_extras, load_model, sensor_bodies  # noqa: E402
from oracle.tasks import OracleAnymalTerrainEnv, OracleShadowHandEnv  # noqa: E402
from test_gpu_parity import _hand_order, _sim_dict  # noqa: E402

DEV = "cuda:0"
from isaacgymenvs_amd.utils.config import compose  # noqa: E402
CASES = (("AnymalTerrain", 128, 400, 13), ("ShadowHand", 32, 250, 13), ("ShadowHand:egg", 32, 200, 13), ("ShadowHand:pen", 32, 200, 13))
if len(sys.argv) > 1 and sys.argv[1] == "pen":      # is a gpu / oracle difference of the pen noise or bias?  other seeds, more envs
    CASES = (("ShadowHand:pen", 48, 200, 29), ("ShadowHand:pen", 48, 200, 71), ("ShadowHand:egg", 48, 200, 29))
for task, n, steps, seed in CASES:
    obj = task.split(":")[1] if ":" in task else None
    task = task.split(":")[0]
    t0 = time.time()
    cfg = None
    if obj:
        cfg = compose(overrides=["task=ShadowHand"])
        cfg["task"]["env"]["numEnvs"] = n
        cfg["task"]["env"]["objectType"] = obj
    env = isaacgymenvs_amd.make(seed=seed, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, **({"cfg": cfg} if cfg else {}))
    if task == "AnymalTerrain":
        orc = OracleAnymalTerrainEnv(load_model("anymal"), _sim_dict(env.sim_params), env._task_params_struct, env.terrain, n, seed=seed, precision="f64")
    else:
        orc = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"), _sim_dict(env.sim_params),
                                  env._task_params_struct, n, seed=seed, **_hand_order(env))
    g = torch.Generator(device="cpu").manual_seed(3)
    G, O = dict(rew=0.0, resets=0, nc=0), dict(rew=0.0, resets=0, nc=0)
    for i in range(steps):
        a = torch.rand((n, env.num_actions), generator=g) * 2 - 1
        _, rew, reset, _ = env.step(a.to(DEV))
        out = orc.step(a.numpy())
        G["rew"] += float(rew.mean()); G["resets"] += int(reset.sum())
        O["rew"] += float(np.mean(out[1])); O["resets"] += int(np.sum(out[2]))
        if task == "ShadowHand":
            G["nc"] += int(env.engine.tensors["object_contact_count"].sum()); O["nc"] += int(orc.eng.ncontacts.sum())
    print(f"{task}{':' + obj if obj else ''}@{n} x {steps}: mean step reward gpu {G['rew'] / steps:.4f} / oracle {O['rew'] / steps:.4f}; resets gpu {G['resets']} / oracle {O['resets']}; "
          f"contacts gpu {G['nc']} / oracle {O['nc']}  ({time.time() - t0:.0f} s)", flush=True)
    del env

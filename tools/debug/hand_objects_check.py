import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, isaacgymenvs_amd
from isaacgymenvs_amd.utils.config import compose
for ot in ("block", "egg", "pen"):
    n = 512
    cfg = compose(overrides=["task=ShadowHand"]); cfg["task"]["env"]["numEnvs"] = n; cfg["task"]["env"]["objectType"] = ot
    env = isaacgymenvs_amd.make(seed=3, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True, cfg=cfg)
    a = torch.zeros((n, 20), device="cuda:0")
    zs, nc, rs = [], [], 0
    for step in range(120):
        o, r, d, e = env.step(a)
        rs += int(d.sum())
        if step % 20 == 19:
            zs.append(float(env.object_pos[:, 2].median())); nc.append(float(env.engine.tensors["object_contact_count"].float().mean()))
    print(ot, "median z", np.round(zs, 3), "mean contacts", np.round(nc, 2), "resets", rs, "finite", bool(torch.isfinite(o["obs"]).all()))

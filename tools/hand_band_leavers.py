#!/usr/bin/env python3
"""Why some ShadowHand envs leave the parity band ("report which contact branch, once").

The engine (fp32; here its CPU build, one Gauss-Seidel sequence per env -- the same per-env functions the one-wave HIP kernels run) and the oracle
(oracle/hand.c, fp64, same solver order and contact-slot rule) step the task from the same seed with the same actions.  For every env that is
outside 5e-3 (1 + step) on a kinematic observation column after 3 steps, the script looks for the FIRST sub-step at which the two runs made a
different discrete choice: a contact sphere inside the contact offset in one run and outside in the other (contact set), a joint-limit row on
the other side or with an impulse in one run only (limit set), a drive at its force limit in one run only, a reset flag.

    python tools/hand_band_leavers.py [num_envs [steps [device]]] > profiles/r4_hand_band_leavers.txt
(device "cuda:0": the HIP kernels; the oracle is then told the finger-per-wave form's block order, as the GPU tests do)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import isaacgymenvs_amd  # noqa: E402
from isaacgymenvs_amd.registry import load_extras, load_model, sensor_bodies  # noqa: E402
from oracle.tasks import OracleShadowHandEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
DEV = sys.argv[3] if len(sys.argv) > 3 else "cpu"
seed = 13
env = isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
sp = env.sim_params
sim = dict(dt=sp.dt, substeps=sp.substeps, iters=sp.iters, gravity=tuple(sp.gravity), contact_offset=sp.contact_offset, rest_offset=sp.rest_offset,
           max_depen_vel=sp.max_depen_vel, erp=sp.erp, plane_mu=sp.plane_mu, ground_z=sp.ground_z, cfm=sp.cfm, warm=sp.warm)
order = dict(solver="gs")
if DEV != "cpu" and int(env.engine.get_option("multi_wave")) != 0:
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    order = dict(solver="blocks", blocks=hand_solver_blocks(load_model("shadow_hand")))
print(f"ShadowHand@{n} on {DEV}, oracle order {order['solver']}, {int(sys.argv[2]) if len(sys.argv) > 2 else 3} control steps")
orc = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"), sim, env._task_params_struct, n,
                          seed=seed, **order)
ex = load_extras("shadow_hand")
fmax = np.array(ex["dof_force_limit"], float)
g = torch.Generator(device="cpu").manual_seed(7)
force_cols = np.r_[48:72, 161:191]
kin_cols = np.setdiff1d(np.arange(211), force_cols)
T = env.engine.tensors
first = {}           # env -> (step, what)
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def note(mask, step, what):
    for e in np.nonzero(mask)[0]:
        first.setdefault(int(e), (step, what))


for step in range(STEPS):
    a = torch.rand((n, 20), generator=g) * 2 - 1
    env.step(a.to(DEV))
    o_obs, o_rew, o_reset = orc.step(a.numpy())
    obs = env.obs_buf.cpu().numpy()
    # discrete choices of the LAST sub-step of this control step, as far as both sides expose them
    nc_e, nc_o = T["object_contact_count"].cpu().numpy(), orc.eng.ncontacts
    note(nc_e != nc_o, step, "contact set (another number of contacts kept)")
    ll_e, ll_o = T["limit_impulse"].cpu().numpy(), orc.eng.laml
    note(((ll_e != 0) != (ll_o != 0)).any(axis=1), step, "limit set (a joint-limit impulse in one run only)")
    note((np.sign(ll_e) * np.sign(ll_o) < 0).any(axis=1), step, "limit set (the other limit of a joint)")
    df_e, df_o = T["dof_force"].cpu().numpy(), orc.eng.dof_force
    sat_e = (np.abs(df_e) >= fmax * (1 - 1e-4)) & (fmax > 0) & (ll_e == 0)
    sat_o = (np.abs(df_o) >= fmax * (1 - 1e-4)) & (fmax > 0) & (ll_o == 0)
    note((sat_e != sat_o).any(axis=1), step, "drive force limit (saturated in one run only)")
    note(env.reset_buf.cpu().numpy() != o_reset, step, "reset flag")
    d = np.abs(obs - o_obs)
    out = d[:, kin_cols].max(axis=1) >= 5e-3 * (1 + step)
    print(f"step {step}: {out.sum()} of {n} envs ({100.0 * out.mean():.2f} %) outside 5e-3 x (1 + step) on a kinematic column; largest difference {d[:, kin_cols].max():.3g}; "
          f"median difference of the envs inside {np.median(d[~out][:, kin_cols].max(axis=1)):.2e}")
leavers = np.nonzero(out)[0]
why = {}
for e in leavers:
    w = first.get(int(e), (None, "no differing discrete choice seen at the ends of the control steps (a difference inside a step's first sub-step, or rounding amplified by a stiff contact)"))[1]
    why[w] = why.get(w, 0) + 1
print(f"\nof the {len(leavers)} envs outside the band after {STEPS} steps, first differing discrete choice (observed at the end of a control step):")
for w, c in sorted(why.items(), key=lambda kv: -kv[1]):
    print(f"  {c:5d}  ({100.0 * c / max(len(leavers), 1):5.1f} %)  {w}")
inside_with_diff = len([e for e in first if e not in set(leavers.tolist())])
print(f"\nenvs INSIDE the band although a discrete choice differed somewhere: {inside_with_diff} of {n - len(leavers)}")
print(f"envs with any differing discrete choice: {len(first)} of {n} ({100.0 * len(first) / n:.2f} %)")

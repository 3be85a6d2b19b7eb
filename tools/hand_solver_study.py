#!/usr/bin/env python
"""Shadow Hand: contacts per limb and convergence of the two solver orders on states of a random-policy rollout of the task
(oracle/hand.c, fp64, CPU; test infrastructure: uses oracle/).

  gs      one Gauss-Seidel sequence over all rows, <= 12 contacts per env              (hand_substep_kernel, one wave per 32 envs)
  blocks  Gauss-Seidel inside a wavefront's rows, Jacobi with mass splitting across
          the wavefronts on the wrist and object coordinates, contacts kept per limb    (hand_substep_mw_kernel, finger per wave)

The rollout runs on the `gs` order with 8 sweeps (the task's num_position_iterations, ShadowHand.yaml:181).  Every 10th control step
ONE sub-step is solved from the rollout's state with 4 / 8 / 16 sweeps of each order and with 400 Gauss-Seidel sweeps (the converged
solution of the same complementarity problem with the same contact set); reported: distance of the resulting velocities (hand dofs
rad/s, object m/s and rad/s) from the converged ones, and how many contacts each limb holds.
Usage: python tools/hand_solver_study.py [num_envs] [steps] [policy: random|zero]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isaacgymenvs_amd.assets.model import hand_solver_blocks, limb_paths  # noqa: E402
from isaacgymenvs_amd.registry import load_extras, load_model, sensor_bodies  # noqa: E402
from isaacgymenvs_amd.tasks.shadow_hand import hand_params_from_cfg  # noqa: E402
from isaacgymenvs_amd.utils.config import compose  # noqa: E402
from oracle.hand import OracleHandEngine  # noqa: E402
from oracle.tasks import OracleShadowHandEnv  # noqa: E402


def main(n=256, steps=300, policy="random"):
    spec, ex, sens = load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand")
    cfg = compose(overrides=["task=ShadowHand"])["task"]
    p = hand_params_from_cfg(cfg)
    ph = cfg["sim"]["physx"]
    sim = dict(dt=cfg["sim"]["dt"], substeps=cfg["sim"]["substeps"], iters=ph["num_position_iterations"], gravity=tuple(cfg["sim"]["gravity"]),
               contact_offset=ph["contact_offset"], rest_offset=ph["rest_offset"], max_depen_vel=ph["max_depenetration_velocity"],
               plane_mu=1.0, ground_z=0.0)
    # the engine's solver constants as VecTask._parse_sim_params reads them
    sim.update(iters=int(ph.get("num_position_iterations", 4)) + int(ph.get("num_velocity_iterations", 0)), erp=float(cfg["sim"].get("erp", 0.5)),
               cfm=float(cfg["sim"].get("cfm", 1e-6)), warm=float(cfg["sim"].get("warm_start", 1.0)))
    blocks = hand_solver_blocks(spec)
    limb, limbs = limb_paths(spec)
    env = OracleShadowHandEnv(spec, ex, sens, sim, p, n, seed=3)
    rng = np.random.default_rng(0)

    def make(solver, iters):
        return OracleHandEngine(spec, ex, n, dict(sim, iters=iters, substeps=1, dt=sim["dt"] / sim["substeps"]), sens, solver=solver,
                                blocks=blocks if solver == "blocks" else None)
    cfgs = [("gs", 4), ("gs", 8), ("gs", 16), ("blocks", 4), ("blocks", 8), ("blocks", 16)]
    engs = {c: make(*c) for c in cfgs}
    ref = {s: make(s, 400) for s in ("gs", "blocks")}       # same contact set as the order it is the limit of
    errs = {c: dict(hand=[], lin=[], ang=[]) for c in cfgs}
    ncon = {s: [] for s in ("gs", "blocks")}
    free = OracleHandEngine(spec, ex, n, dict(sim, iters=1, substeps=1, dt=sim["dt"] / sim["substeps"]), sens, solver="blocks",
                            blocks=dict(blocks, limb_cap=[24] * len(limbs)))     # per-limb counts with the limb caps out of the way
    lcount = []

    def load(e):
        src = env.eng
        e.eng.state[:] = src.eng.state
        e.obj[:] = src.obj; e.targets[:] = src.targets; e.obj_force[:] = src.obj_force; e.scale[:] = src.scale; e.limit_shift[:] = src.limit_shift

    for s in range(steps):
        a = rng.uniform(-1, 1, (n, 20)) if policy == "random" else np.zeros((n, 20))
        env.step(a.astype(np.float32))
        if s % 10 == 9:
            for sv, r in ref.items():
                load(r); r.step()
                ncon[sv].append(r.ncontacts.copy())
            load(free); free.step(); lcount.append(free.limb_counts.copy())
            for c, e in engs.items():
                load(e); e.step()
                r = ref[c[0]]
                errs[c]["hand"].append(np.abs(e.qd - r.qd).max(1)); errs[c]["lin"].append(np.abs(e.obj[:, 7:10] - r.obj[:, 7:10]).max(1))
                errs[c]["ang"].append(np.abs(e.obj[:, 10:13] - r.obj[:, 10:13]).max(1))
    print(f"shadow_hand: {n} envs, {steps} control steps ({policy} policy), one sub-step of h = {sim['dt'] / sim['substeps']:.5f} s solved from "
          f"{len(ncon['gs'])} sampled states per env")
    for sv in ("gs", "blocks"):
        nc = np.concatenate(ncon[sv])
        print(f"  contacts per env, {sv:6s} contact set: mean {nc.mean():.2f}  median {np.median(nc):.0f}  99 % {np.percentile(nc, 99):.0f}  max {nc.max()}")
    lc = np.concatenate(lcount)
    print(f"  contacts per limb without the limb caps (<= 4 per body); caps of the finger-per-wave form {blocks['limb_cap']}:")
    for l in range(len(limbs)):
        hist = np.bincount(lc[:, l], minlength=9)[:9] / len(lc)
        print(f"    limb {l} (bodies {limbs[l][0]}..{limbs[l][-1]}): mean {lc[:, l].mean():.2f}  P(k) k=0..8: " + " ".join(f"{x:.4f}" for x in hist)
              + f"  P(> cap) = {(lc[:, l] > blocks['limb_cap'][l]).mean():.4f}")
    print("order   sweeps  |v - v_converged|_inf:  hand dofs [rad/s]  mean / 95 % / 99 % / max        object lin [m/s]  mean / 99 % / max       object ang [rad/s]  mean / 99 % / max")
    for c in cfgs:
        h, l, a = (np.concatenate(errs[c][k]) for k in ("hand", "lin", "ang"))
        print(f"{c[0]:7s} {c[1]:4d}   {'':22s}{h.mean():9.2e} {np.percentile(h, 95):9.2e} {np.percentile(h, 99):9.2e} {h.max():9.2e}      "
              f"{l.mean():9.2e} {np.percentile(l, 99):9.2e} {l.max():9.2e}      {a.mean():9.2e} {np.percentile(a, 99):9.2e} {a.max():9.2e}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 256, int(a[1]) if len(a) > 1 else 300, a[2] if len(a) > 2 else "random")

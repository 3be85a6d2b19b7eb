#!/usr/bin/env python
"""Convergence of the two solver orders of the engine on states of a random-policy rollout (oracle/physics.c, fp64, CPU):

  gs      one Gauss-Seidel sequence over all rows                      (single-wave kernels)
  blocks  Gauss-Seidel inside a wavefront's block of rows, Jacobi with
          mass splitting across the blocks                            (limb-per-wave kernels)

For every sampled state one sub-step is solved with 4 / 8 / 16 sweeps of each order and with 400 Gauss-Seidel sweeps (the
converged solution of the same LCP); reported is the distance of the resulting generalised velocity from the converged one
(max over the dofs, statistics over envs x samples) and the largest remaining constraint violation.
Usage: python tools/solver_convergence.py [ant|humanoid] [num_envs] [steps]     (test infrastructure: uses oracle/)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isaacgymenvs_amd.assets.model import solver_blocks  # noqa: E402
from isaacgymenvs_amd.registry import load_model, load_selfcol, sensor_bodies  # noqa: E402
from oracle.engine import OracleEngine  # noqa: E402

TASK = dict(ant=dict(z0=0.44, term=0.31, gear=15.0, iters=4), humanoid=dict(z0=1.34, term=0.8, gear=60.0, iters=4))


def main(name="humanoid", n=256, steps=200):
    t = TASK[name]
    spec = load_model(name)
    sc = load_selfcol(name)
    kw = dict(selfcol=sc, kmax=12, kpair=3, warm_slots=9) if sc else {}
    sim = dict(dt=1 / 60, substeps=2, iters=t["iters"], max_depen_vel=10.0, contact_offset=0.02)
    blocks = solver_blocks(spec, self_collision=bool(sc))
    rng = np.random.default_rng(0)

    def make(solver, iters, substeps=1):
        return OracleEngine(spec, n, params=dict(sim, iters=iters, substeps=substeps, dt=sim["dt"] / 2 * substeps),
                            sensor_bodies=sensor_bodies(name), precision="f64", solver=solver,
                            blocks=blocks if solver == "blocks" else None, **kw)
    roll = make("gs", t["iters"], substeps=2)

    def reset(ids):
        roll.state[ids] = 0
        roll.state[ids, 6] = 1
        roll.state[ids, 2] = t["z0"]
        roll.q[ids] = rng.uniform(-0.2, 0.2, (len(ids), spec.nd))
        lo, up = np.asarray(spec.dof_lower), np.asarray(spec.dof_upper)
        lim = np.asarray(spec.dof_limited) > 0
        roll.q[ids] = np.where(lim, np.clip(roll.q[ids], lo, up), roll.q[ids])
    reset(np.arange(n))
    cfgs = [("gs", 4), ("gs", 8), ("gs", 16), ("blocks", 4), ("blocks", 8), ("blocks", 16)]
    engs = {c: make(*c) for c in cfgs}
    ref = make("gs", 400)
    errs = {c: [] for c in cfgs}
    ncon = []
    for s in range(steps):
        tau = rng.uniform(-1, 1, (n, spec.nd)) * t["gear"]
        if s % 10 == 9:
            ref.state[:] = roll.state
            ref.step(tau)
            ncon.append(((np.abs(np.nan_to_num(ref.state)[:, 13 + 2 * spec.nd:13 + 2 * spec.nd + 3 * len(spec.sph_body):3]) > 0).sum(1)).mean())
            for c, e in engs.items():
                e.state[:] = roll.state
                e.step(tau)
                dv = np.abs(np.concatenate([e.root[:, 7:], e.qd], 1) - np.concatenate([ref.root[:, 7:], ref.qd], 1)).max(1)
                errs[c].append(dv[np.isfinite(dv)])
        roll.step(tau)
        bad = np.nonzero((roll.root[:, 2] < t["term"]) | ~np.isfinite(roll.state).all(1))[0]
        if len(bad):
            reset(bad)
    print(f"{name}: {n} envs, {steps} steps, one sub-step of h = 1/120 s solved from {len(ncon)} sampled states per env; "
          f"mean active ground contacts per env {np.mean(ncon):.2f}")
    print("order   sweeps   |v - v_converged|_inf  mean      median    95 %      99 %      max")
    for c in cfgs:
        e = np.concatenate(errs[c])
        print(f"{c[0]:7s} {c[1]:4d}   {'':22s}{e.mean():9.2e} {np.median(e):9.2e} {np.percentile(e, 95):9.2e} {np.percentile(e, 99):9.2e} {e.max():9.2e}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "humanoid", int(a[1]) if len(a) > 1 else 256, int(a[2]) if len(a) > 2 else 200)

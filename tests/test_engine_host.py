"""The specialised engine core (csrc/core/engine.hpp + generated model tables), built for the HOST with g++, against the
independent CPU oracle (oracle/physics.c, fp64).  Same source the HIP kernels compile; different formulation from the
oracle (branch-sparse L^T L factor, whitened PGS, depth-first tree pass vs dense CRBA + Cholesky + generalized-velocity
PGS), so agreement here checks the algebra, and the -m gpu tests then check the device build against the same oracle."""
import numpy as np
import pytest

import hostsim  # tests/hostbuild (path added by conftest.py)
from isaacgymenvs_amd.registry import load_model, sensor_bodies
from oracle.engine import OracleEngine

SIM = dict(dt=0.0166, substeps=2, iters=4, gravity=(0.0, 0.0, -9.81), contact_offset=0.02, rest_offset=0.0,
           max_depen_vel=10.0, erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)


def _random_state(spec, n, rng, z_lo, z_hi):
    nd = spec.nd
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    lo, up = np.maximum(lo, -3.0), np.minimum(up, 3.0)
    root = np.zeros((n, 13))
    root[:, 0:2] = rng.normal(size=(n, 2))
    root[:, 2] = rng.uniform(z_lo, z_hi, n)
    q = rng.normal(size=(n, 4)); q[:, 3] += 3; q /= np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 3:7] = q
    root[:, 7:13] = rng.normal(size=(n, 6))
    return root, rng.uniform(lo, up, (n, nd)), rng.normal(size=(n, nd)) * 2


@pytest.mark.parametrize("task,z_lo,z_hi,gear,full", [("cartpole", 2.0, 2.0, 100.0, False), ("ant", 0.3, 0.6, 15.0, False),
                                                      ("humanoid", 0.9, 1.4, 60.0, True)])
def test_host_build_of_engine_core_matches_oracle(task, z_lo, z_hi, gear, full):
    spec = load_model(task)
    sb = sensor_bodies(task)
    n = 128
    lib = hostsim.build(humanoid=full)
    rng = np.random.default_rng(0)
    root, q, qd = _random_state(spec, n, rng, z_lo, z_hi)
    if spec.fixed_base:
        root[:, :] = 0; root[:, 2] = z_lo; root[:, 6] = 1
    tau = rng.uniform(-gear, gear, (n, spec.nd))
    orc = OracleEngine(spec, n, params=SIM, sensor_bodies=sb, precision="f64")
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    nd, nsph = spec.nd, len(spec.sph_body)
    st = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd), np.float32)
    st[:, :13] = root; st[:, 13:13 + nd] = q; st[:, 13 + nd:13 + 2 * nd] = qd
    out = np.zeros((n, 6 * len(sb) + nd + 3 * nsph), np.float32)
    p = hostsim.make_params(SIM)
    tau32 = np.ascontiguousarray(tau, np.float32)
    for it in range(3):
        hostsim.step(lib, task, p, st, tau32, out)
        orc.step(tau)
        scale = max(1.0, np.abs(orc.qd).max())
        e = max(np.abs(st[:, :13] - orc.root).max(), np.abs(st[:, 13:13 + nd] - orc.q).max(),
                np.abs(st[:, 13 + nd:13 + 2 * nd] - orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (task, it, e)
        # warm-start impulses, force sensors and dof forces agree too
        assert np.abs(st[:, 13 + 2 * nd:] - orc.lam).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        if len(sb):
            assert np.abs(out[:, :6 * len(sb)] - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
        assert np.abs(out[:, 6 * len(sb):6 * len(sb) + nd] - orc.dof_force).max() < 2e-3 * max(1.0, np.abs(orc.dof_force).max())


def test_host_build_on_heightfield_matches_oracle():
    """ANYmal on a random rough height field: general contact normals, per-env friction, per-body net contact forces."""
    spec = load_model("anymal")
    n = 128
    lib = hostsim.build()
    rng = np.random.default_rng(3)
    rows, cols, hscale, vscale, border = 120, 140, 0.1, 0.005, 2.0
    hs = (rng.integers(-30, 30, (rows // 4 + 1, cols // 4 + 1)).repeat(4, 0).repeat(4, 1)[:rows, :cols]
          + rng.integers(-6, 6, (rows, cols))).astype(np.int16)
    sim = dict(SIM, dt=0.005, substeps=1, iters=5, max_depen_vel=100.0)
    root, q, qd = _random_state(spec, n, rng, 0.25, 0.7)
    root[:, 0] = rng.uniform(1, 8, n); root[:, 1] = rng.uniform(1, 10, n)
    q = rng.uniform(-1.0, 1.0, (n, spec.nd))
    tau = rng.uniform(-80, 80, (n, spec.nd))
    mu = rng.uniform(0.5, 1.25, n).astype(np.float32)
    orc = OracleEngine(spec, n, params=sim, precision="f64")
    orc.set_ground(hs, hscale, vscale, border)
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    nd, nsph = spec.nd, len(spec.sph_body)
    st = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd), np.float32)
    st[:, :13] = root; st[:, 13:13 + nd] = q; st[:, 13 + nd:13 + 2 * nd] = qd
    out = np.zeros((n, nd + 3 * nsph), np.float32)
    netf = np.zeros((n, spec.nb, 3), np.float32)
    p = hostsim.make_params(sim)
    tau32 = np.ascontiguousarray(tau, np.float32)
    contacts = 0
    for it in range(4):
        hostsim.step_terrain(lib, "anymal", p, st, tau32, out, hs, hscale, vscale, border, mu, netf)
        orc.step(tau, env_mu=mu)
        scale = max(1.0, np.abs(orc.qd).max())
        e = max(np.abs(st[:, :13] - orc.root).max(), np.abs(st[:, 13:13 + nd] - orc.q).max(),
                np.abs(st[:, 13 + nd:13 + 2 * nd] - orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        assert np.abs(netf - orc.netf).max() < 2e-3 * max(1.0, np.abs(orc.netf).max())
        contacts += int((np.abs(orc.netf).sum(-1) > 0).sum())
    assert contacts > 50   # the scenario does exercise contacts

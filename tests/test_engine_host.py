"""The specialised engine core (csrc/core/engine.hpp + generated model tables), built for the HOST with g++, against the
independent CPU oracle (oracle/physics.c, fp64).  Same source the HIP kernels compile; different formulation from the
oracle (branch-sparse L^T L factor, whitened PGS, depth-first tree pass vs dense CRBA + Cholesky + generalized-velocity
PGS), so agreement here checks the algebra, and the -m gpu tests then check the device build against the same oracle."""
import numpy as np
import pytest

import hostsim  # tests/hostbuild (path added by conftest.py)
from isaacgymenvs_amd.registry import load_model, sensor_bodies
from isaacgymenvs_amd.assets.model import solver_blocks
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


@pytest.mark.parametrize("slope_threshold", [0.0, 0.5])
def test_host_build_on_heightfield_matches_oracle(slope_threshold):
    """ANYmal on a random rough height field: general contact normals, per-env friction, per-body net contact forces.
    slope_threshold 0.5 (terrain.slopeTreshold of AnymalTerrain.yaml): edges steeper than that are levelled to their lower end, as the
    reference's mesh generator turns them into vertical walls -- most block edges of this field (up to 60 raw units, threshold 10)."""
    import ctypes as C
    spec = load_model("anymal")
    n = 128
    lib = hostsim.build()
    lib.hs_set_slope_threshold.argtypes = [C.c_float]
    lib.hs_set_slope_threshold.restype = None
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
    orc.set_ground(hs, hscale, vscale, border, slope_threshold=slope_threshold)
    lib.hs_set_slope_threshold(slope_threshold * hscale / vscale)
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
    lib.hs_set_slope_threshold(0.0)
    assert contacts > 50   # the scenario does exercise contacts


@pytest.mark.parametrize("scaled", [False, True])
def test_host_build_of_hand_engine_matches_oracle(scaled):
    """csrc/core/hand_engine.hpp compiled for the host (fp32) against oracle/hand.py (fp64 restatement): hand + cube sub-steps
    with finger/cube contacts, tendon rows, implicit PD drives; contact counts identical, states agree to fp32 rounding.
    scaled: with per-env `actor_params` factors (hand link masses, joint damping, drive stiffness, tendon stiffness / damping, object
    mass and size; reference ShadowHand.yaml:104-159) -- and those factors do change the motion."""
    import ctypes as C
    from oracle.hand import OracleHandEngine
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    lib = hostsim.build_hand()
    spec, ex = load_model("shadow_hand"), load_extras("shadow_hand")
    sim = dict(dt=1.0 / 60.0, substeps=2, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.002, rest_offset=0.0,
               max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-4, warm=0.9)
    N, nd = 6, spec.nd
    rng = np.random.default_rng(5)
    orc = OracleHandEngine(spec, ex, N, sim, sensor_bodies("shadow_hand"))
    lo, up = orc.lo, orc.up
    orc.q[:] = lo + (up - lo) * rng.uniform(0.2, 0.5, (N, nd))
    orc.qd[:] = rng.normal(0, 0.5, (N, nd))
    orc.targets[:] = lo + (up - lo) * rng.uniform(0.1, 0.9, (N, nd))
    # cube a little above the palm, dropping onto the fingers
    tips = orc.fingertip_states()
    orc.obj[:, 0:3] = tips[:, :, 0:3].mean(1) + rng.normal(0, 0.01, (N, 3)) + np.array([0.0, 0.0, 0.02])
    qn = rng.normal(size=(N, 4)); orc.obj[:, 3:7] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    orc.obj[:, 7:10] = rng.normal(0, 0.1, (N, 3))
    from oracle.hand import CUBE_HALF as half, CUBE_MASS as mass, CUBE_INERTIA as inertia
    mu = 1.0
    ss = 4 * nd + 13
    state = np.zeros((N, ss), np.float32)
    state[:, 0:nd] = orc.q; state[:, nd:2 * nd] = orc.qd; state[:, 3 * nd:4 * nd] = orc.targets; state[:, 4 * nd:] = orc.obj
    root13 = np.zeros(13, np.float32); root13[:7] = orc.eng.root[0, :7]
    ns = len(orc.sens)
    out = np.zeros((N, 6 * ns + nd + 1), np.float32)
    P = hostsim.make_params(sim)
    tipsh = np.zeros((N, ns, 13), np.float32)
    lib.hs_hand_fingertips(N, state.ctypes.data_as(C.c_void_p), root13.ctypes.data_as(C.c_void_p), tipsh.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(tipsh, tips, atol=2e-5)
    total = 0
    scale = None
    if scaled:
        scale = np.ones((N, 8), np.float32)
        scale[:, 0] = rng.uniform(0.5, 1.5, N); scale[:, 1] = rng.uniform(0.3, 3.0, N); scale[:, 2] = rng.uniform(0.75, 1.5, N)
        scale[:, 3] = rng.uniform(0.75, 1.5, N); scale[:, 4] = rng.uniform(0.3, 3.0, N); scale[:, 5] = rng.uniform(0.5, 1.5, N)
        scale[:, 6] = rng.uniform(0.95, 1.05, N)
        orc.scale[:] = scale
        lsh = rng.normal(0, 0.03, (N, 2 * nd)).astype(np.float32)
        orc.limit_shift[:] = lsh
        plain = OracleHandEngine(spec, ex, N, sim, sensor_bodies("shadow_hand"))
        plain.q[:] = orc.q; plain.qd[:] = orc.qd; plain.targets[:] = orc.targets; plain.obj[:] = orc.obj
    for it in range(12):
        orc.step()
        rc = lib.hs_step_hand(C.byref(P), N, state.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                              root13.ctypes.data_as(C.c_void_p), C.c_float(half), C.c_float(mass), C.c_float(inertia), C.c_float(mu),
                              scale.ctypes.data_as(C.c_void_p) if scaled else None, lsh.ctypes.data_as(C.c_void_p) if scaled else None)
        assert rc == 0
        np.testing.assert_allclose(out[:, 6 * ns:6 * ns + nd], orc.dof_force, atol=2e-2 * max(1.0, np.abs(orc.dof_force).max()))
        if scaled and it < 3:
            plain.step()
            if it == 2:
                assert np.abs(plain.q - orc.q).max() > 5e-3          # the factors matter
        np.testing.assert_array_equal(out[:, -1].astype(int), orc.ncontacts)
        total += int(orc.ncontacts.sum())
        np.testing.assert_allclose(state[:, 0:nd], orc.q, atol=2e-3)
        np.testing.assert_allclose(state[:, 4 * nd:4 * nd + 7], orc.obj[:, 0:7], atol=5e-3)
    assert total > 0, "scenario never produced finger/cube contacts"
    assert np.isfinite(state).all()


@pytest.mark.parametrize("shape", ["block", "egg", "pen"])
def test_host_build_of_finger_per_wave_hand_engine_matches_block_order_oracle(shape):
    """csrc/core/hand_engine_mw.hpp compiled for the host (fp32; the four role waves of an env run as four threads that meet at a pthread
    barrier) against oracle/hand.c with the block solver order (fp64): wrist Schur complement / carries exchanged between the roles,
    joint-limit rows in registers, contacts kept per limb in the fixed row shape [limb | wrist], block sweeps with mass splitting on the
    wrist and the object coordinates.  Contact counts identical, states agree to fp32 rounding -- with `actor_params` factors and
    joint-limit shifts in place, for the cube, the egg and the pen."""
    import ctypes as C
    from oracle.hand import OracleHandEngine, CUBE_HALF as half, CUBE_MASS as mass, CUBE_INERTIA as inertia
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    lib = hostsim.build_hand()
    spec, ex = load_model("shadow_hand"), load_extras("shadow_hand")
    sim = dict(dt=1.0 / 60.0, substeps=2, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.002, rest_offset=0.0,
               max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-4, warm=0.9)
    N, nd = 8, spec.nd
    rng = np.random.default_rng(11)
    objp = {"block": None, "egg": dict(shape="egg", dims=[0.03, 0.03, 0.04], mass=0.151, inertia=[7.5e-5, 7.5e-5, 5.4e-5]),
            "pen": dict(shape="pen", dims=[0.008, 0.1], mass=0.042, inertia=[1.5e-4, 1.5e-4, 1.5e-6])}[shape]
    orc = OracleHandEngine(spec, ex, N, sim, sensor_bodies("shadow_hand"), obj=objp, solver="blocks", blocks=hand_solver_blocks(spec))
    lo, up = orc.lo, orc.up
    orc.q[:] = lo + (up - lo) * rng.uniform(0.2, 0.5, (N, nd))
    orc.qd[:] = rng.normal(0, 0.5, (N, nd))
    orc.targets[:] = lo + (up - lo) * rng.uniform(0.1, 0.9, (N, nd))
    tips = orc.fingertip_states()
    orc.obj[:, 0:3] = tips[:, :, 0:3].mean(1) + rng.normal(0, 0.01, (N, 3)) + np.array([0.0, 0.0, 0.02])
    qn = rng.normal(size=(N, 4)); orc.obj[:, 3:7] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    orc.obj[:, 7:10] = rng.normal(0, 0.1, (N, 3))
    scale = np.ones((N, 8), np.float32)
    scale[:, 0] = rng.uniform(0.5, 1.5, N); scale[:, 1] = rng.uniform(0.3, 3.0, N); scale[:, 2] = rng.uniform(0.75, 1.5, N)
    scale[:, 3] = rng.uniform(0.75, 1.5, N); scale[:, 4] = rng.uniform(0.3, 3.0, N); scale[:, 5] = rng.uniform(0.5, 1.5, N)
    scale[:, 6] = rng.uniform(0.95, 1.05, N)
    lsh = rng.normal(0, 0.03, (N, 2 * nd)).astype(np.float32)
    orc.scale[:] = scale; orc.limit_shift[:] = lsh
    ss = 4 * nd + 13
    state = np.zeros((N, ss), np.float32)
    state[:, 0:nd] = orc.q; state[:, nd:2 * nd] = orc.qd; state[:, 3 * nd:4 * nd] = orc.targets; state[:, 4 * nd:] = orc.obj
    root13 = np.zeros(13, np.float32); root13[:7] = orc.eng.root[0, :7]
    ns = len(orc.sens)
    out = np.zeros((N, 6 * ns + nd + 1), np.float32)
    P = hostsim.make_params(sim)
    shp = {"block": 0, "pen": 1, "egg": 2}[shape]
    dims = np.zeros(3, np.float32); in3 = np.zeros(3, np.float32)
    om, oi = mass, inertia
    if objp is not None:
        dims[:len(objp["dims"])] = objp["dims"]; in3[:] = objp["inertia"]; om, oi = objp["mass"], 1.0
    total, fingers = 0, 0
    for it in range(10):
        orc.step()
        rc = lib.hs_step_hand_mw(C.byref(P), N, state.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), root13.ctypes.data_as(C.c_void_p),
                                 C.c_float(half), C.c_float(om), C.c_float(oi), C.c_float(1.0), scale.ctypes.data_as(C.c_void_p),
                                 lsh.ctypes.data_as(C.c_void_p), shp, dims.ctypes.data_as(C.c_void_p), in3.ctypes.data_as(C.c_void_p))
        assert rc == 0
        np.testing.assert_array_equal(out[:, -1].astype(int), orc.ncontacts)
        total += int(orc.ncontacts.sum()); fingers += int(orc.limb_counts[:, 1:].sum())
        np.testing.assert_allclose(state[:, 0:nd], orc.q, atol=2e-4)
        np.testing.assert_allclose(state[:, nd:2 * nd], orc.qd, atol=2e-2)
        np.testing.assert_allclose(state[:, 4 * nd:4 * nd + 7], orc.obj[:, 0:7], atol=5e-4)
        # (force-limited drives: the largest joint force is now the wrist's limit force, not an unclamped drive's -- hence the relative part)
        np.testing.assert_allclose(out[:, 6 * ns:6 * ns + nd], orc.dof_force, atol=2e-3 * max(1.0, np.abs(orc.dof_force).max()), rtol=5e-3)
        # (3e-3 of the largest force present since round 5: with the thumb where the asset puts it the pen is pinched between thumb and fingers, and one
        #  of 240 sensor elements sat at 2.2e-3 -- fp32 against fp64 through the friction disc of a sliding contact)
        np.testing.assert_allclose(out[:, :6 * ns], orc.sensor, atol=3e-3 * max(1.0, np.abs(orc.sensor).max()))
    assert total > 100 and fingers > 30, "scenario must exercise palm and finger contacts"
    assert np.isfinite(state).all()


@pytest.mark.parametrize("form", ["one_wave", "finger_per_wave"])
def test_host_build_of_hand_engine_limits_the_drive_forces(form):
    """Effort-limited position drives (shared.xml:250-269 forcerange; core/hand_engine.hpp drive_clamp_update) in both forms of the hand
    engine: every driven dof's target far beyond its force range (kp dq = 2..4 x fmax, both directions), the hand at rest, the cube away.
    The joint forces the engine reports are +- fmax while a joint moves freely, the motion is the oracle's (oracle/physics.c OrDriveClamp;
    its closed form is pinned by tests/test_oracle_physics.py::test_hand_drive_force_limit_known_answer), and joints that reach a limit
    stop there."""
    import ctypes as C
    from oracle.hand import OracleHandEngine, CUBE_HALF as half, CUBE_MASS as mass, CUBE_INERTIA as inertia
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    lib = hostsim.build_hand()
    spec, ex = load_model("shadow_hand"), load_extras("shadow_hand")
    sim = dict(dt=1.0 / 60.0, substeps=2, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.002, rest_offset=0.0,
               max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-4, warm=0.9)
    N, nd = 8, spec.nd
    rng = np.random.default_rng(2)
    kw = dict(solver="blocks", blocks=hand_solver_blocks(spec)) if form == "finger_per_wave" else {}
    orc = OracleHandEngine(spec, ex, N, sim, sensor_bodies("shadow_hand"), **kw)
    kp, fmax = np.array(ex["dof_kp"], float), np.array(ex["dof_force_limit"], float)
    driven = kp > 0
    orc.q[:] = 0.5 * (orc.lo + orc.up); orc.qd[:] = 0.0
    orc.targets[:] = orc.q
    orc.targets[:, driven] += rng.choice([-1.0, 1.0], (N, int(driven.sum()))) * rng.uniform(2.0, 4.0, (N, int(driven.sum()))) * (fmax / np.where(driven, kp, 1.0))[driven]
    orc.obj[:, 0:3] = [0.0, 0.0, 5.0]
    ss = 4 * nd + 13
    state = np.zeros((N, ss), np.float32)
    state[:, 0:nd] = orc.q; state[:, 3 * nd:4 * nd] = orc.targets; state[:, 4 * nd:] = orc.obj
    root13 = np.zeros(13, np.float32); root13[:7] = orc.eng.root[0, :7]
    ns = len(orc.sens)
    out = np.zeros((N, 6 * ns + nd + 1), np.float32)
    P = hostsim.make_params(sim)
    dims = np.zeros(3, np.float32)
    saturated = 0
    for it in range(6):
        orc.step()
        if form == "one_wave":
            rc = lib.hs_step_hand(C.byref(P), N, state.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), root13.ctypes.data_as(C.c_void_p),
                                  C.c_float(half), C.c_float(mass), C.c_float(inertia), C.c_float(1.0), None, None)
        else:
            rc = lib.hs_step_hand_mw(C.byref(P), N, state.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), root13.ctypes.data_as(C.c_void_p),
                                     C.c_float(half), C.c_float(mass), C.c_float(inertia), C.c_float(1.0), None, None, 0,
                                     dims.ctypes.data_as(C.c_void_p), dims.ctypes.data_as(C.c_void_p))
        assert rc == 0
        np.testing.assert_allclose(state[:, 0:nd], orc.q, atol=5e-4)
        np.testing.assert_allclose(state[:, nd:2 * nd], orc.qd, atol=5e-2)
        f = out[:, 6 * ns:6 * ns + nd]
        np.testing.assert_allclose(f, orc.dof_force, atol=5e-3, rtol=5e-3)
        # away from the joint limits (no limit force in the sum) the delivered force never exceeds the force range -- it is the range itself until
        # the joint moves fast enough for the drive's damping to take the force below it -- and points towards the target
        free = driven[None, :] & (orc.laml == 0.0)
        fm = np.broadcast_to(fmax, f.shape)
        assert np.all(np.abs(f[free]) <= fm[free] * (1 + 2e-3))
        at_range = free & (np.abs(f) >= fm * (1 - 2e-3))
        saturated += int(at_range.sum())
        assert np.all(np.sign(f[at_range]) == np.sign((orc.targets - orc.q)[at_range]))
        assert np.abs(kp * (orc.targets - orc.q))[free].max() > 1.5 * fmax.max() or it > 0       # the unclamped drive would push harder
    assert saturated > 40
    assert np.all(state[:, 0:nd] < orc.up + 0.02) and np.all(state[:, 0:nd] > orc.lo - 0.02)


def test_host_build_with_position_drives_and_body_forces_matches_oracle():
    """Quadcopter model: implicit PD position drives (kp 1000) on the rotor joints + thrust forces on the rotor bodies in
    their local frames (engine.hpp Drive) against oracle/physics.c or_step_drive; free flight, no contacts."""
    import ctypes as C
    lib = hostsim.build()
    spec = load_model("quadcopter")
    sens = sensor_bodies("quadcopter")
    sim = dict(dt=0.01, substeps=2, iters=4, gravity=(0.0, 0.0, -9.81), contact_offset=0.02, rest_offset=0.001,
               max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=0.9)
    N, nd = 16, spec.nd
    rng = np.random.default_rng(11)
    orc = OracleEngine(spec, N, params=sim, sensor_bodies=sens, precision="f64")
    orc.root[:, 2] = 1.0 + rng.uniform(-0.2, 0.5, N)
    qn = rng.normal(size=(N, 4)); qn[:, 3] += 4; orc.root[:, 3:7] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    orc.root[:, 7:13] = rng.normal(0, 0.5, (N, 6))
    orc.q[:] = rng.uniform(-0.2, 0.2, (N, nd))
    state = np.ascontiguousarray(orc.state, np.float32)
    out = np.zeros((N, orc.os), np.float32)
    P = hostsim.make_params(sim)
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    assert np.allclose(up, np.deg2rad(30)) and np.allclose(lo, -np.deg2rad(30))
    for it in range(40):
        target = rng.uniform(lo, up, (N, nd))
        thrust = rng.uniform(0.0, 2.0, (N, 4))
        fext = np.zeros((N, spec.nb, 3)); fext[:, sens, 2] = thrust
        fs = np.zeros((N, 4, 3), np.float32); fs[:, :, 2] = thrust
        tg = np.ascontiguousarray(target, np.float32)
        orc.step_drive(np.zeros((N, nd)), 1000.0, 0.0, target, fext)
        rc = lib.hs_step_drive(b"quadcopter", C.byref(P), N, state.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                               C.c_float(1000.0), C.c_float(0.0), tg.ctypes.data_as(C.c_void_p), fs.ctypes.data_as(C.c_void_p))
        assert rc == 0
        tol = 2e-4 * (1 + it)
        np.testing.assert_allclose(state[:, :13], orc.root, atol=tol)
        np.testing.assert_allclose(state[:, 13:13 + nd], orc.q, atol=tol)
    # hover check: total thrust = weight along a level craft => no vertical acceleration
    orc.root[:] = 0; orc.root[:, 2] = 1; orc.root[:, 6] = 1; orc.q[:] = 0; orc.qd[:] = 0
    w = spec.total_mass() * 9.81 / 4
    fext = np.zeros((N, spec.nb, 3)); fext[:, sens, 2] = w
    for _ in range(50):
        orc.step_drive(np.zeros((N, nd)), 1000.0, 0.0, np.zeros((N, nd)), fext)
    assert np.abs(orc.root[:, 2] - 1.0).max() < 1e-6 and np.abs(orc.root[:, 7:13]).max() < 1e-6


def test_multi_wave_substep_matches_oracle_and_single_wave():
    """core/engine_mw.hpp (one leg per wave, trunk recomputed by every wave, Schur complements / right-hand-side carries exchanged
    through the row store, every wave sweeping its own rows) run as four host threads per env that meet at a barrier: same state,
    impulses, sensors and joint forces as the fp64 oracle IN THE BLOCK SOLVER ORDER within the stated tolerance; the single-wave form
    (one Gauss-Seidel sequence) solves the same problem in another order."""
    spec, sb = load_model("ant"), sensor_bodies("ant")
    n = 96
    lib = hostsim.build()
    rng = np.random.default_rng(5)
    root, q, qd = _random_state(spec, n, rng, 0.3, 0.6)
    tau = rng.uniform(-15, 15, (n, spec.nd))
    orc = OracleEngine(spec, n, params=SIM, sensor_bodies=sb, precision="f64", solver="blocks", blocks=solver_blocks(spec))
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    nd, nsph = spec.nd, len(spec.sph_body)

    def fresh():
        st = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd), np.float32)
        st[:, :13] = root; st[:, 13:13 + nd] = q; st[:, 13 + nd:13 + 2 * nd] = qd
        return st, np.zeros((n, 6 * len(sb) + nd + 3 * nsph), np.float32)
    (st, out), (st1, out1) = fresh(), fresh()
    p = hostsim.make_params(SIM)
    tau32 = np.ascontiguousarray(tau, np.float32)
    for it in range(3):
        hostsim.step_mw_ant(lib, p, st, tau32, out)
        hostsim.step(lib, "ant", p, st1, tau32, out1)
        orc.step(tau)
        scale = max(1.0, np.abs(orc.qd).max())
        e = max(np.abs(st[:, :13] - orc.root).max(), np.abs(st[:, 13:13 + nd] - orc.q).max(), np.abs(st[:, 13 + nd:13 + 2 * nd] - orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        assert np.abs(st[:, 13 + 2 * nd:] - orc.lam).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        assert np.abs(out[:, :6 * len(sb)] - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
        assert np.abs(out[:, 6 * len(sb):6 * len(sb) + nd] - orc.dof_force).max() < 2e-3 * max(1.0, np.abs(orc.dof_force).max())
        # (the single-wave form sweeps the same rows in ONE Gauss-Seidel sequence: from these random, deeply penetrating states 4 sweeps
        #  of either order are far from converged and the two differ by O(1) rad/s; tools/solver_convergence.py compares the orders
        #  on rollout states.  Here: root position / orientation, which one step barely moves, must still agree.)
        assert np.abs(st[:, :7] - st1[:, :7]).max() < 5e-2 * (it + 1)
    assert np.abs(orc.sensor).max() > 10.0                                  # the ants do stand on the ground


def test_multi_wave_substep_on_heightfield_matches_oracle():
    """ANYmal on a rough height field through the multi-wave sub-step: general contact normals, per-env friction, per-body net contact
    forces written by the wave that owns the body."""
    spec = load_model("anymal")
    n = 64
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
    orc = OracleEngine(spec, n, params=sim, precision="f64", solver="blocks", blocks=solver_blocks(spec))
    orc.set_ground(hs, hscale, vscale, border)
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    nd, nsph = spec.nd, len(spec.sph_body)
    st = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd), np.float32)
    st[:, :13] = root; st[:, 13:13 + nd] = q; st[:, 13 + nd:13 + 2 * nd] = qd
    out = np.zeros((n, nd + 3 * nsph), np.float32)
    netf = np.zeros((n, spec.nb, 3), np.float32)
    p = hostsim.make_params(sim)
    tau32 = np.ascontiguousarray(tau, np.float32)
    for it in range(4):
        hostsim.step_mw_terrain(lib, p, st, tau32, out, hs, hscale, vscale, border, mu, netf)
        orc.step(tau, env_mu=mu)
        scale = max(1.0, np.abs(orc.qd).max())
        e = max(np.abs(st[:, :13] - orc.root).max(), np.abs(st[:, 13:13 + nd] - orc.q).max(), np.abs(st[:, 13 + nd:13 + 2 * nd] - orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        assert np.abs(netf - orc.netf).max() < 2e-3 * max(1.0, np.abs(orc.netf).max())
        assert np.abs(st[:, 13 + 2 * nd:] - orc.lam).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
    assert np.abs(orc.netf).max() > 50.0

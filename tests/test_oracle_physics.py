"""First-principles known-answer tests that pin the CPU physics oracle (oracle/physics.c).

The reference's physics lives in the closed isaacgym/PhysX binary (reference vec_task.py:382) and the reference has no
simulation tests or recorded trajectories: parity against PhysX is UNPINNED.  These tests pin what can
be pinned: conservation laws, closed-form motions, static equilibrium and the model constants of SURVEY.md Appendix A.
"""
import numpy as np
import pytest

from isaacgymenvs_amd.registry import load_model, sensor_bodies
from oracle.engine import OracleEngine

G = 9.81


def _eng(name, n=1, **kw):
    p = dict(dt=1.0 / 60.0, substeps=2, iters=4, gravity=(0.0, 0.0, -G), contact_offset=0.02, rest_offset=0.0,
             max_depen_vel=10.0, erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)
    p.update(kw)
    return OracleEngine(load_model(name), n, params=p, sensor_bodies=sensor_bodies(name), precision="f64")


def test_model_constants_match_asset_files():
    """SURVEY.md Appendix A (derived from the reference's MJCF/URDF files)."""
    ant, hum, cp = load_model("ant"), load_model("humanoid"), load_model("cartpole")
    assert (ant.nd, hum.nd, cp.nd) == (8, 21, 2)
    assert abs(ant.total_mass() - 0.9109) < 2e-3          # nv_ant.xml, density 5
    assert abs(hum.total_mass() - 40.84) < 0.05           # nv_humanoid.xml, density 1000
    assert not ant.fixed_base and not hum.fixed_base and cp.fixed_base
    # ant hips +-40 deg (nv_ant.xml:48-75)
    lo, up = np.minimum(ant.dof_lower, ant.dof_upper), np.maximum(ant.dof_lower, ant.dof_upper)
    np.testing.assert_allclose(lo[0::2], -np.deg2rad(40), atol=1e-6)
    np.testing.assert_allclose(up[0::2], np.deg2rad(40), atol=1e-6)


def test_free_fall_matches_closed_form():
    e = _eng("ant", ground_z=-1000.0)           # ground far away
    e.root[:, 2] = 5.0
    q0 = e.q.copy()
    e.q[:] = np.array(load_model("ant").dof_lower) * 0 + np.where(np.minimum(load_model("ant").dof_lower, load_model("ant").dof_upper) > 0,
                                                                  np.minimum(load_model("ant").dof_lower, load_model("ant").dof_upper),
                                                                  np.where(np.maximum(load_model("ant").dof_lower, load_model("ant").dof_upper) < 0,
                                                                           np.maximum(load_model("ant").dof_lower, load_model("ant").dof_upper), 0.0))
    tau = np.zeros((1, 8))
    n = 30
    for _ in range(n):
        e.step(tau)
    t = n / 60.0
    h = 1.0 / 120.0
    # semi-implicit Euler: v_k = -g k h ; z_k = z0 - g h^2 k (k+1)/2
    k = 2 * n
    assert abs(e.root[0, 9] + G * t) < 1e-6
    # the root body is not the COM of the whole ant, so compare the COM-independent quantity: root z error is small
    assert abs(e.root[0, 2] - (5.0 - G * h * h * k * (k + 1) / 2)) < 5e-3
    # no rotation is induced by gravity alone on a symmetric, unactuated body (legs at rest angles move slightly)
    assert np.abs(e.root[0, 10:13]).max() < 0.5


def test_energy_is_conserved_without_dissipation():
    """Cart-pole with zero damping/armature dissipation: KE + PE drifts only by the integrator's O(h) error."""
    e = _eng("cartpole", substeps=8)
    e.root[:, 2] = 2.0
    e.q[0] = [0.0, 0.4]
    e.qd[0] = [0.3, 0.0]
    ke0, pe0 = e.energy(0)
    tau = np.zeros((1, 2))
    es = []
    for _ in range(120):
        e.step(tau)
        ke, pe = e.energy(0)
        es.append(ke + pe)
    drift = np.abs(np.array(es) - (ke0 + pe0)).max() / abs(pe0 - min(es) + 1e-9 + abs(ke0 + pe0))
    assert drift < 2e-2, drift


def test_cartpole_matches_planar_ode():
    """planar cart-pole, m_c = 1, m_p = 1, l_com = 0.47, I_com = (0.06^2 + 1^2)/12, force on the cart."""
    spec = load_model("cartpole")
    e = _eng("cartpole", substeps=16)
    e.root[:, 2] = 2.0
    th0 = 0.3
    e.q[0] = [0.0, th0]
    F = 3.0
    tau = np.array([[F, 0.0]])
    mc, mp = float(spec.mass[1]) if spec.nb > 2 else 1.0, 1.0
    mc, mp, l = 1.0, 1.0, 0.47
    Ic = (0.06 ** 2 + 1.0 ** 2) / 12.0

    # reference ODE integrated with a much smaller step (RK4).  Pole angle theta is measured from upright about +x, the
    # cart slides along +y; sign conventions are fitted once from the first step (see below).
    def f(s, sgn):
        x, th, xd, thd = s
        # M(q) qdd = rhs with generalized coords (x, th); pole COM at x + sgn*l*sin(th) horizontally, l*cos(th) up
        a11, a12, a22 = mc + mp, sgn * mp * l * np.cos(th), Ic + mp * l * l
        r1 = F + sgn * mp * l * np.sin(th) * thd * thd
        r2 = mp * G * l * np.sin(th)
        det = a11 * a22 - a12 * a12
        return np.array([xd, thd, (r1 * a22 - a12 * r2) / det, (a11 * r2 - a12 * r1) / det])

    def rk4(s, dt, sgn):
        k1 = f(s, sgn); k2 = f(s + 0.5 * dt * k1, sgn); k3 = f(s + 0.5 * dt * k2, sgn); k4 = f(s + dt * k3, sgn)
        return s + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)

    n = 30
    for _ in range(n):
        e.step(tau)
    best = None
    for sgn in (+1.0, -1.0):
        s = np.array([0.0, th0, 0.0, 0.0])
        for _ in range(n * 64):
            s = rk4(s, (1.0 / 60.0) / 64, sgn)
        err = max(abs(s[0] - e.q[0, 0]), abs(s[1] - e.q[0, 1]))
        best = err if best is None else min(best, err)
    # first-order integrator with h = 1/960 over 0.5 s of a falling pole: a few 1e-3
    assert best < 1e-2, best


def test_ant_rests_on_the_ground_with_its_weight():
    e = _eng("ant", n=1)
    spec = load_model("ant")
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    e.q[:] = np.where(lo > 0, lo, np.where(up < 0, up, 0.0))
    e.root[:, 2] = 0.44
    tau = np.zeros((1, 8))
    fz = []
    for i in range(300):
        e.step(tau)
        if i >= 240:
            fz.append(e.sph_force[0, :, 2].sum())  # world force (x, y, z) per sphere
    assert abs(np.mean(fz) - spec.total_mass() * G) < 0.02 * spec.total_mass() * G
    assert e.root[0, 2] > 0.31                       # above terminationHeight (reference Ant.yaml:29)
    assert np.abs(e.root[0, 7:10]).max() < 0.02      # at rest ...
    assert np.abs(e.root[0, 10:13]).max() < 0.15     # ... up to the slow yaw creep 4 PGS sweeps leave in the friction rows
    # joint limits respected
    assert (e.q[0] >= lo - 0.02).all() and (e.q[0] <= up + 0.02).all()


def test_momentum_conservation_in_free_flight():
    """No external force but gravity: the horizontal linear momentum of the whole ant stays zero under joint torques,
    up to the first-order error of the semi-implicit integrator (error ~ h: 4x more sub-steps => ~4x less drift)."""
    errs = []
    for ss in (2, 8, 32):
        e = _eng("ant", ground_z=-1000.0, substeps=ss)
        e.root[:, 2] = 3.0
        rng = np.random.default_rng(0)
        px = []
        for _ in range(20):
            tau = rng.uniform(-5, 5, (1, 8))
            e.step(tau)
            M, _ = e.dynamics(0)
            v = np.concatenate([e.root[0, 7:13], e.qd[0]])
            px.append((M @ v)[:2])      # rows 0..2 of the generalized momentum = total linear momentum
        errs.append(np.abs(np.array(px)).max())
    assert errs[0] < 0.1 and errs[1] < errs[0] / 2.5 and errs[2] < errs[1] / 2.5, errs


@pytest.mark.parametrize("name", ["ant", "humanoid"])
def test_f32_build_tracks_f64(name):
    spec = load_model(name)
    n = 32
    rng = np.random.default_rng(1)
    p = dict(dt=1.0 / 60.0, substeps=2, iters=4)
    a = OracleEngine(spec, n, params=p, sensor_bodies=sensor_bodies(name), precision="f64")
    b = OracleEngine(spec, n, params=p, sensor_bodies=sensor_bodies(name), precision="f32")
    z = 0.6 if name == "ant" else 1.4
    for e in (a, b):
        e.root[:, 2] = z
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    q = rng.uniform(lo, up, (n, spec.nd))
    a.q[:] = q; b.q[:] = q
    for _ in range(3):
        tau = rng.uniform(-10, 10, (n, spec.nd))
        a.step(tau); b.step(tau)
    assert np.abs(a.q - b.q).max() < 2e-4 and np.abs(a.qd - b.qd).max() < 5e-3


def test_friction_cone_on_a_tilted_plane():
    """Coulomb cone: a robot resting on the plane with gravity tilted by theta starts to slide iff tan(theta) > mu (combined
    coefficient = mean of the geom / per-env value and the plane's).  Gentle slopes (mu = 0.1) so that nothing topples or rolls."""
    mu = 0.1
    spec = load_model("ant")
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    out = {}
    for name, factor in (("stick", 0.6), ("slide", 1.5)):
        th = np.arctan(mu * factor)
        e = _eng("ant", plane_mu=mu, erp=0.2)
        e.q[:] = 0.5 * (lo + up)
        e.root[:, 2] = 0.30
        for _ in range(90):                                   # settle on level ground first
            e.step(np.zeros((1, spec.nd)), env_mu=np.array([mu]))
        e.set_params(**dict(e.params_dict, gravity=(G * np.sin(th), 0.0, -G * np.cos(th))))
        x0 = e.root[0, 0]
        for _ in range(120):                                  # 2 s on the slope
            e.step(np.zeros((1, spec.nd)), env_mu=np.array([mu]))
        out[name] = (e.root[0, 0] - x0, e.root[0, 7], th)
    assert abs(out["stick"][1]) < 1e-2 and abs(out["stick"][0]) < 0.02, out           # static friction holds
    th = out["slide"][2]
    a = G * (np.sin(th) - mu * np.cos(th))                    # net acceleration along the plane of a rigidly translating body
    # The passive legs shuffle and the body yaws a little while it slides, so the per-foot friction vectors are not all exactly
    # uphill: the acceleration lies between the rigid-translation value and frictionless sliding
    assert 0.95 * a * 2.0 < out["slide"][1] < G * np.sin(th) * 2.0, (out, a)
    # what must hold exactly: every touching sphere sits ON the friction cone, and the normal forces carry the weight
    f = e.sph_force[0]
    touching = f[:, 2] > 1e-6
    assert touching.sum() >= 3
    np.testing.assert_allclose(np.hypot(f[touching, 0], f[touching, 1]), mu * f[touching, 2], rtol=1e-6)
    np.testing.assert_allclose(f[:, 2].sum(), spec.total_mass() * G * np.cos(th), rtol=2e-3)


@pytest.mark.parametrize("robot,solver", [("ant", "gs"), ("ant", "blocks"), ("humanoid", "gs"), ("humanoid", "blocks"), ("humanoid", "blocks+caps")])
def test_friction_opposes_the_sliding_velocity(robot, solver):
    """Coulomb's law on a tilted plane and on level ground, for the Ant on its foot spheres and the Humanoid lying on its capsules' end spheres
    (tests/friction_util.py): static below the friction angle; a = g (sin theta - mu cos theta) along the fall line -- a DIAGONAL
    of the tangent axes -- and nothing across it; stop distance v0^2 / (2 mu g) on the line of the push.  With the rows' own step sizes (before
    round 6) the Humanoid was braked 16 degrees off its sliding direction: lateral acceleration 0.27 m/s^2, stop 2.2 cm beside the line."""
    import friction_util as F
    from isaacgymenvs_amd.assets.model import solver_blocks
    spec = load_model(robot)
    kw = dict(solver="blocks", blocks=solver_blocks(spec)) if solver == "blocks" else {}
    if solver == "blocks+caps":        # as the Humanoid's limb-wave kernels run it: self-collision on, ground contacts capped per wave (4 / 4 / 3)
        from isaacgymenvs_amd.registry import load_selfcol
        kw = dict(solver="blocks", blocks=solver_blocks(spec, self_collision=True, wave_caps=True), selfcol=load_selfcol(robot), kpair=3)
    p = dict(dt=1.0 / 60.0, substeps=2, iters=4, gravity=(0.0, 0.0, -G), contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0, erp=0.5,
             plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)
    e = OracleEngine(spec, 1, params=p, sensor_bodies=sensor_bodies(robot), precision="f64", **kw)
    mu = 0.5
    out = F.friction_known_answers(F.OracleRig(e, env_mu=np.array([2.0 * mu - 1.0])), spec, robot, mu, 1.0 / 60.0)
    ae, th, de = out["slide_acc_expected"], out["slide_theta"], out["stop_dist_expected"]
    strict = robot == "humanoid" or solver == "blocks"      # (the PD-held Ant creeps at cm/s in the one-sequence order at 4 iterations: tests/test_friction.py)
    assert out["stick_speed"] < (1e-3 if strict else 0.1) and out["stick_shift"] < (0.02 if strict else 0.25), out
    assert abs(out["slide_acc"][0] - ae) < 0.03 * ae and abs(out["slide_acc_lateral"][0]) < 0.015 * G * np.sin(th), out
    assert 0.80 * de < out["stop_dist"][0] < 1.05 * de and abs(out["stop_lateral"][0]) < 0.01 * de and out["stop_speed"] < 5e-3, out


def test_angular_momentum_of_a_tumbling_body_converges_first_order():
    """Torque-free flight of the articulated Humanoid with all joints moving.  Joint springs / dampers are internal forces, so
    the total angular momentum about the centre of mass (from the oracle's own body velocities) is conserved by the equations of
    motion; the semi-implicit Euler integrator conserves it to first order: the drift over a fixed time halves with the step."""
    import ctypes as C
    from oracle.engine import _ptr
    spec = load_model("humanoid")
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)

    def drift(dt, T=0.5):
        e = _eng("humanoid", gravity=(0.0, 0.0, 0.0), ground_z=-1000.0, dt=dt, substeps=1)
        rng = np.random.default_rng(2)
        e.root[:, 2] = 2.0
        e.root[:, 10:13] = rng.normal(0, 1.5, 3)
        e.qd[:] = rng.normal(0, 1.0, spec.nd)
        e.q[:] = 0.5 * (lo + up)

        def momentum():
            _, _, bp = e.energy(0, poses=True)
            s = np.ascontiguousarray(e.state[0])
            v6 = np.zeros(6)
            items, M, com = [], 0.0, np.zeros(3)
            for b in range(spec.nb):
                e.lib.or_body_vel(C.byref(e.model), _ptr(s), b, _ptr(v6))       # [omega; v of the body point at O]
                R = bp[b, 3:12].reshape(3, 3)
                c = bp[b, 0:3] + R @ np.asarray(spec.com[b])
                om = v6[0:3].copy()
                vc = v6[3:6] + np.cross(om, c - e.root[0, :3])
                I6 = np.asarray(spec.inertia[b])
                Il = np.array([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]])
                items.append((spec.mass[b], c, vc, R @ Il @ R.T @ om))
                M += spec.mass[b]; com += spec.mass[b] * c
            com /= M
            L, P = np.zeros(3), np.zeros(3)
            for m, c, vc, Lb in items:
                L += Lb + m * np.cross(c - com, vc); P += m * vc
            return L, P
        L0, P0 = momentum()
        for _ in range(int(round(T / dt))):
            e.step(np.zeros((1, spec.nd)))
        L1, P1 = momentum()
        assert np.linalg.norm(L0) > 1.0
        return np.linalg.norm(L1 - L0) / np.linalg.norm(L0), np.linalg.norm(P1 - P0) / (np.linalg.norm(P0) + 1e-9)
    d = [drift(dt) for dt in (1 / 240, 1 / 480, 1 / 960)]
    assert d[2][0] < 0.012, d
    assert 0.45 < d[1][0] / d[0][0] < 0.55 and 0.45 < d[2][0] / d[1][0] < 0.55, d      # first order
    assert max(x[1] for x in d) < 0.05, d                                              # linear momentum too


def test_joint_limits_hold_against_a_constant_torque():
    """A hinge pushed into its limit by a constant effort stops there: the limit row's impulse balances the effort
    (dof_force = applied + limit force ~ 0 at rest) and the overshoot stays within the ERP band."""
    spec = load_model("ant")
    e = _eng("ant", gravity=(0.0, 0.0, 0.0), ground_z=-1000.0)
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    e.root[:, 2] = 3.0
    e.q[:] = 0.5 * (lo + up)
    tau = np.zeros((1, spec.nd)); tau[0, 0] = 2.0; tau[0, 1] = -2.0
    for _ in range(300):
        e.step(tau)
    assert e.q[0, 0] <= up[0] + 0.02 and e.q[0, 0] >= up[0] - 0.02, (e.q[0, 0], up[0])
    assert e.q[0, 1] >= lo[1] - 0.02 and e.q[0, 1] <= lo[1] + 0.02, (e.q[0, 1], lo[1])
    assert abs(e.qd[0, 0]) < 5e-2 and abs(e.qd[0, 1]) < 5e-2


def test_ellipsoid_distance_approximation_used_for_the_egg():
    """oracle/hand.py::sphere_ellipsoid (twin of HandSim::sphere_ellipsoid): exact along the axes and on the surface, within 4 % of the true
    distance for points up to 5 mm off the surface of the 3 x 3 x 4 cm egg, normal = surface normal at the closest point within 3 deg."""
    from oracle.hand import sphere_ellipsoid
    a = np.array([0.03, 0.03, 0.04])
    for k in range(3):                                          # along an axis the distance is exact
        c = np.zeros(3); c[k] = a[k] + 0.007
        d, n = sphere_ellipsoid(c, 0.002, a)
        assert abs(d - 0.005) < 1e-12 and abs(n[k] - 1) < 1e-12
    rng = np.random.default_rng(0)
    worst_d, worst_ang = 0.0, 0.0
    for _ in range(300):
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        ps = a * u                                              # a surface point and its outward normal
        ns = ps / a ** 2; ns /= np.linalg.norm(ns)
        off = rng.uniform(-0.002, 0.005)
        d, n = sphere_ellipsoid(ps + off * ns, 0.0, a)
        worst_d = max(worst_d, abs(d - off))
        worst_ang = max(worst_ang, np.degrees(np.arccos(np.clip(n @ ns, -1, 1))))
    assert worst_d < 0.04 * 0.005 + 1e-5, worst_d
    assert worst_ang < 3.0, worst_ang
    d, n = sphere_ellipsoid(np.zeros(3), 0.001, a)              # centre: deepest, some unit normal
    assert d < -0.03 and abs(np.linalg.norm(n) - 1) < 1e-12


# ------------------------------------------------------------------ solver orders (physics.c OrModel.solver)
SIM = dict(dt=1.0 / 60.0, substeps=2, iters=4, gravity=(0.0, 0.0, -G), contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0,
           erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)


def _rollout_states(name, n, steps, seed=0):
    """states of a random-torque rollout (resets when the robot falls): contacts, joint limits, self contacts all occur"""
    from isaacgymenvs_amd.registry import load_selfcol
    spec, sc = load_model(name), load_selfcol(name)
    kw = dict(selfcol=sc, kmax=12, kpair=3, warm_slots=9) if sc else {}
    rng = np.random.default_rng(seed)
    z0, term, gear = dict(ant=(0.44, 0.31, 15.0), humanoid=(1.34, 0.8, 60.0))[name]
    e = OracleEngine(spec, n, params=dict(SIM), sensor_bodies=sensor_bodies(name), precision="f64", **kw)
    e.root[:, 2] = z0
    e.q[:] = np.clip(rng.uniform(-0.2, 0.2, (n, spec.nd)), spec.dof_lower, spec.dof_upper)
    for _ in range(steps):
        e.step(rng.uniform(-1, 1, (n, spec.nd)) * gear)
        bad = np.nonzero(e.root[:, 2] < term)[0]
        e.state[bad] = 0
        e.state[bad, 6] = 1
        e.state[bad, 2] = z0
    return spec, sc, kw, e.state.copy(), rng.uniform(-1, 1, (n, spec.nd)) * gear


@pytest.mark.parametrize("name", ["ant", "humanoid"])
def test_block_solver_order_converges_to_the_gauss_seidel_solution(name):
    """The limb-per-wave kernels sweep their rows block by block (Gauss-Seidel inside a block, weighted Jacobi across blocks, physics.c
    solve_blocks).  Both orders solve the same complementarity problem: with many sweeps they must agree, and after the task's 4
    sweeps the block order must be within a small factor of plain Gauss-Seidel's distance to that solution."""
    from isaacgymenvs_amd.assets.model import solver_blocks
    n = 64
    spec, sc, kw, state, tau = _rollout_states(name, n, 60)
    blocks = solver_blocks(spec, self_collision=bool(sc))

    def solve(solver, iters):
        e = OracleEngine(spec, n, params=dict(SIM, iters=iters, substeps=1, dt=SIM["dt"] / 2), sensor_bodies=sensor_bodies(name),
                         precision="f64", solver=solver, blocks=blocks if solver == "blocks" else None, **kw)
        e.state[:] = state
        e.step(tau)
        return np.concatenate([e.root[:, 7:], e.qd], 1)
    ref = solve("gs", 3000)
    far = solve("blocks", 3000)
    # frictional contact has no unique solution in general (the cone depends on the normal impulse): most envs agree exactly
    d = np.abs(far - ref).max(1)
    assert np.median(d) < 1e-9 and (d < 1e-3).mean() > 0.75, (np.median(d), (d < 1e-3).mean())
    e_gs = np.abs(solve("gs", 4) - ref).max(1).mean()
    e_bl = np.abs(solve("blocks", 4) - ref).max(1).mean()
    e_bl8 = np.abs(solve("blocks", 8) - ref).max(1).mean()
    assert e_bl < 4.0 * e_gs + 1e-3 and e_bl8 < e_bl, (e_gs, e_bl, e_bl8)


def test_block_solver_keeps_the_known_answers():
    """static weight and free fall do not depend on the order the rows are swept in"""
    from isaacgymenvs_amd.assets.model import solver_blocks
    spec = load_model("ant")
    e = OracleEngine(spec, 1, params=dict(SIM), sensor_bodies=sensor_bodies("ant"), precision="f64", solver="blocks",
                     blocks=solver_blocks(spec))
    e.root[:, 2] = 0.3
    e.q[:] = [0, 0.9, 0, -0.9, 0, -0.9, 0, 0.9]
    for _ in range(400):
        e.step(np.zeros((1, spec.nd)))
    fz = e.sph_force[0, :, 2].sum()
    assert abs(fz - float(np.sum(spec.mass)) * 9.81) < 2e-3 * float(np.sum(spec.mass)) * 9.81, fz
    assert np.abs(e.qd).max() < 1e-3 and np.abs(e.root[0, 7:]).max() < 1e-3


@pytest.mark.parametrize("shape", ["block", "egg", "pen"])
def test_hand_oracle_in_c_equals_its_numpy_twin(shape):
    """oracle/hand.c (OpenMP, every env of ShadowHand@16384 in seconds) restates oracle/hand.py (numpy): same contacts, states equal to
    1e-9 over 12 steps with contacts, tendon rows, `actor_params` factors, joint-limit shifts and an external force on the object."""
    from oracle.hand import OracleHandEngine
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    spec, ex = load_model("shadow_hand"), load_extras("shadow_hand")
    sim = dict(dt=1.0 / 60.0, substeps=2, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.002, rest_offset=0.0,
               max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-4, warm=0.9)
    N, nd = 6, spec.nd
    objp = {"block": None, "egg": dict(shape="egg", dims=[0.03, 0.03, 0.04], mass=0.151, inertia=[7.5e-5, 7.5e-5, 5.4e-5]),
            "pen": dict(shape="pen", dims=[0.008, 0.1], mass=0.042, inertia=[1.5e-4, 1.5e-4, 1.5e-6])}[shape]
    rng = np.random.default_rng(5)
    a = OracleHandEngine(spec, ex, N, sim, sensor_bodies("shadow_hand"), obj=objp, backend="numpy")
    b = OracleHandEngine(spec, ex, N, sim, sensor_bodies("shadow_hand"), obj=objp, backend="c")
    lo, up = a.lo, a.up
    a.q[:] = lo + (up - lo) * rng.uniform(0.2, 0.5, (N, nd)); a.qd[:] = rng.normal(0, 0.5, (N, nd))
    a.targets[:] = lo + (up - lo) * rng.uniform(0.1, 0.9, (N, nd))
    tips = a.fingertip_states()
    a.obj[:, 0:3] = tips[:, :, 0:3].mean(1) + rng.normal(0, 0.01, (N, 3)) + np.array([0.0, 0.0, 0.02])
    qn = rng.normal(size=(N, 4)); a.obj[:, 3:7] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    a.obj[:, 7:10] = rng.normal(0, 0.1, (N, 3))
    for col, (x0, x1) in enumerate([(0.5, 1.5), (0.3, 3.0), (0.75, 1.5), (0.75, 1.5), (0.3, 3.0), (0.5, 1.5), (0.95, 1.05)]):
        a.scale[:, col] = rng.uniform(x0, x1, N)
    a.limit_shift[:] = rng.normal(0, 0.03, (N, 2 * nd)); a.obj_force[:] = rng.normal(0, 0.2, (N, 3))
    for k in ("targets", "obj", "scale", "limit_shift", "obj_force"):
        getattr(b, k)[:] = getattr(a, k)
    b.q[:] = a.q; b.qd[:] = a.qd
    tot = 0
    for it in range(12):
        a.step(); b.step()
        tot += int(a.ncontacts.sum())
        np.testing.assert_array_equal(a.ncontacts, b.ncontacts)
        np.testing.assert_allclose(b.q, a.q, atol=1e-9); np.testing.assert_allclose(b.obj, a.obj, atol=1e-9)
        np.testing.assert_allclose(b.laml, a.laml, atol=1e-9)
        np.testing.assert_allclose(b.sensor, a.sensor, atol=1e-7 * max(1.0, np.abs(a.sensor).max()))
        np.testing.assert_allclose(b.dof_force, a.dof_force, atol=1e-7 * max(1.0, np.abs(a.dof_force).max()))
    assert tot > 100


@pytest.mark.parametrize("solver", ["gs", "blocks"])
def test_hand_drive_force_limit_known_answer(solver):
    """Effort-limited position drives (shared.xml:250-269 forcerange; allegro_hand.py:264): a finger joint whose target is far away
    (kp dq = 3x its force range), everything else at its target, hand at rest, nothing in contact.  The actuator then delivers exactly
    fmax, so the sub-step's velocities solve (M + armature + h (D + h kp) on the OTHER dofs) v = h fmax e_d -- the saturated drive's own
    implicit terms are cancelled by the clamp impulse.  Inside the force range, and with a limit nobody reaches, nothing changes."""
    from oracle.hand import OracleHandEngine
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    spec, ex = load_model("shadow_hand"), load_extras("shadow_hand")
    sim = dict(dt=1.0 / 60.0, substeps=1, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.002, rest_offset=0.0,
               max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=0.9)
    h, nd = sim["dt"], spec.nd
    kw = dict(solver="blocks", blocks=hand_solver_blocks(spec)) if solver == "blocks" else {}
    names = list(spec.dof_names)
    for jn in ("robot0:FFJ2", "robot0:THJ3", "robot0:WRJ0"):
        d = names.index(jn)
        fmax, kp = ex["dof_force_limit"][d], ex["dof_kp"][d]
        for sign in (1.0, -1.0):
            o = OracleHandEngine(spec, ex, 1, sim, sensor_bodies("shadow_hand"), backend="c", **kw)
            o.pair_k = 0.0          # the closed form is a hand whose links touch nothing: the hand-to-hand pairs off (at mid-range angles fingers overlap)
            o.q[:] = 0.5 * (o.lo + o.up); o.qd[:] = 0.0
            o.targets[:] = o.q
            o.targets[0, d] = o.q[0, d] + sign * 3.0 * fmax / kp           # (targets are not clamped to the joint range here)
            o.obj[:, 0:3] = [0.0, 0.0, 5.0]                                  # the cube far away: no contact
            M, bias = o.eng.dynamics(0)
            assert np.abs(bias).max() < 1e-12
            c = np.array(spec.dof_damping, float) + h * np.array(ex["dof_kp"], float)
            A = M + np.diag(np.array(spec.dof_armature, float) + h * c)
            A[d, d] -= h * c[d]
            for t in ex["tendons"]:                                          # the coupling tendons' damping (inside their range: no stiffness)
                (d0, d1), (c0, c1) = t["dof"], t["coef"]
                cv = np.zeros(nd); cv[d0], cv[d1] = c0, c1
                A += h * float(ex["tendon_damping"]) * np.outer(cv, cv)
            rhs = np.zeros(nd); rhs[d] = h * sign * fmax
            v_expect = np.linalg.solve(A, rhs)
            o.step()
            np.testing.assert_allclose(o.qd[0], v_expect, atol=1e-9 * max(1.0, np.abs(v_expect).max()), rtol=1e-7)
            assert abs(o.dof_force[0, d] - sign * fmax) < 1e-9                # what the actuator delivered
            # unclamped, the same drive pushes harder (not three times: the light finger is already limited by the implicit damping h kp)
            u = OracleHandEngine(spec, ex, 1, sim, sensor_bodies("shadow_hand"), backend="c", **kw)
            u.pair_k = 0.0          # the closed form is a hand whose links touch nothing: the hand-to-hand pairs off (at mid-range angles fingers overlap)
            u.force_limit = None
            u.q[:] = 0.5 * (u.lo + u.up); u.qd[:] = 0.0; u.targets[:] = o.targets; u.obj[:] = 0.0; u.obj[:, 2] = 5.0; u.obj[:, 6] = 1.0
            u.step()
            assert abs(u.qd[0, d]) > 1.05 * abs(o.qd[0, d])
    # inside the force range the clamp is inert: bit-equal to the unclamped drive
    rng = np.random.default_rng(0)
    a = OracleHandEngine(spec, ex, 4, sim, sensor_bodies("shadow_hand"), backend="c", **kw)
    a.pair_k = 0.0          # the closed form is a hand whose links touch nothing: the hand-to-hand pairs off (at mid-range angles fingers overlap)
    b = OracleHandEngine(spec, ex, 4, sim, sensor_bodies("shadow_hand"), backend="c", **kw)
    b.pair_k = 0.0          # the closed form is a hand whose links touch nothing: the hand-to-hand pairs off (at mid-range angles fingers overlap)
    b.force_limit = None
    a.q[:] = a.lo + (a.up - a.lo) * rng.uniform(0.3, 0.7, (4, nd))
    a.targets[:] = a.q + rng.uniform(-0.05, 0.05, (4, nd))
    b.q[:] = a.q; b.targets[:] = a.targets
    for x in (a, b):
        x.obj[:, 0:3] = [0.0, 0.0, 5.0]
    for _ in range(5):
        a.step(); b.step()
    np.testing.assert_array_equal(a.q, b.q)


def test_hand_block_order_and_gauss_seidel_order_solve_the_same_problem():
    """The block order of the finger-per-wave kernel (oracle/hand.c solver 1: Gauss-Seidel inside a block, Jacobi with mass splitting on the
    wrist and object coordinates across blocks) and the one Gauss-Seidel sequence solve the same complementarity problem: on states of a
    random-policy rollout of the task both end, with many sweeps, at the same velocities; after the task's 8 sweeps the block order is
    within a small factor of the sequence's distance from that solution (tools/hand_solver_study.py reports the statistics)."""
    from oracle.hand import OracleHandEngine
    from oracle.tasks import OracleShadowHandEnv
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    from isaacgymenvs_amd.tasks.shadow_hand import hand_params_from_cfg
    from isaacgymenvs_amd.utils.config import compose
    spec, ex, sens = load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand")
    cfg = compose(overrides=["task=ShadowHand"])["task"]
    ph = cfg["sim"]["physx"]
    sim = dict(dt=cfg["sim"]["dt"], substeps=2, iters=8, gravity=tuple(cfg["sim"]["gravity"]), contact_offset=ph["contact_offset"],
               rest_offset=ph["rest_offset"], max_depen_vel=ph["max_depenetration_velocity"], plane_mu=1.0, ground_z=0.0, erp=0.5, cfm=1e-6, warm=1.0)
    n = 96
    blocks = hand_solver_blocks(spec)
    free = dict(blocks, limb_cap=[40] * len(blocks["limb_cap"]))       # caps out of the way: both orders keep every contact
    env = OracleShadowHandEnv(spec, ex, sens, sim, hand_params_from_cfg(cfg), n, seed=3)

    def make(solver, iters):
        x = OracleHandEngine(spec, ex, n, dict(sim, iters=iters, substeps=1, dt=sim["dt"] / 2), sens, solver=solver, blocks=free if solver == "blocks" else None)
        x.kmax = 40
        return x
    e = {k: make(*k) for k in [("gs", 8), ("blocks", 8), ("gs", 3000), ("blocks", 3000)]}
    rng = np.random.default_rng(0)
    for s in range(20):
        env.step(rng.uniform(-1, 1, (n, 20)).astype(np.float32))
    for x in e.values():
        x.eng.state[:] = env.eng.eng.state; x.obj[:] = env.eng.obj; x.targets[:] = env.eng.targets; x.obj_force[:] = env.eng.obj_force
        x.step()
    ref, refb = e[("gs", 3000)], e[("blocks", 3000)]
    assert int(ref.ncontacts.sum()) > n
    np.testing.assert_array_equal(ref.ncontacts, refb.ncontacts)
    vel = lambda x: np.concatenate([x.qd, x.obj[:, 7:13]], 1)         # noqa: E731
    dconv = np.abs(vel(ref) - vel(refb)).max(1)
    # (a friction LCP has more than one solution in rare configurations, and a few envs have not converged after 3000 sweeps)
    assert np.median(dconv) < 1e-9 and np.percentile(dconv, 80) < 1e-3, (np.median(dconv), np.percentile(dconv, 80))
    d_gs = np.abs(vel(e[("gs", 8)]) - vel(ref)).max(1); d_bl = np.abs(vel(e[("blocks", 8)]) - vel(ref)).max(1)
    assert d_bl.mean() < 3.0 * d_gs.mean() + 1e-3, (d_bl.mean(), d_gs.mean())

"""Self-collision of the Humanoid (reference humanoid.py:194 creates the actor with collision filter 0; BASELINE config 3 names
"self-contact").  PhysX's own contact generation cannot be pinned (closed binary), so these tests pin the engine's stated rule
-- one contact per limb pair, the deepest capsule pair, equal and opposite impulses at the contact point -- by first principles
on the CPU oracle, and then the specialised engine core (host build of csrc/core/engine.hpp) against the oracle."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from isaacgymenvs_amd.assets.model import (body_frames, collision_capsules, limb_paths, segment_distance,
                                           self_collision_groups, self_collision_pairs)
from isaacgymenvs_amd.codegen import selfcol_layout, topology
from isaacgymenvs_amd.registry import load_model, load_selfcol, sensor_bodies
from oracle.engine import OracleEngine, _ptr

SIM = dict(dt=0.0166, substeps=2, iters=4, gravity=(0.0, 0.0, -9.81), contact_offset=0.02, rest_offset=0.0,
           max_depen_vel=10.0, erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)


def test_segment_distance_known_answers_and_brute_force():
    z = np.zeros((1, 3))
    # crossed segments one above the other: distance = the vertical offset, closest points at the crossing
    ca, cb = segment_distance(np.array([[-1.0, 0, 0]]), np.array([[1.0, 0, 0]]), np.array([[0, -1.0, 0.3]]), np.array([[0, 1.0, 0.3]]))
    np.testing.assert_allclose(ca, [[0, 0, 0]], atol=1e-12); np.testing.assert_allclose(cb, [[0, 0, 0.3]], atol=1e-12)
    # collinear, apart: end to end
    ca, cb = segment_distance(z, np.array([[1.0, 0, 0]]), np.array([[1.5, 0, 0]]), np.array([[3.0, 0, 0]]))
    np.testing.assert_allclose(np.linalg.norm(ca - cb), 0.5, atol=1e-12)
    # point (zero-length segment) on either side against a segment: foot of the perpendicular, clamped to the ends
    ca, cb = segment_distance(z, np.array([[2.0, 0, 0]]), np.array([[0.7, 0.4, 0]]), np.array([[0.7, 0.4, 0]]))
    np.testing.assert_allclose(ca, [[0.7, 0, 0]], atol=1e-12)
    ca, cb = segment_distance(np.array([[3.0, 0.4, 0]]), np.array([[3.0, 0.4, 0]]), z, np.array([[2.0, 0, 0]]))
    np.testing.assert_allclose(cb, [[2.0, 0, 0]], atol=1e-12)
    # random segments (some degenerate) against a dense parameter scan, and the C oracle against the same function
    rng = np.random.default_rng(0)
    n = 3000
    a0 = rng.normal(size=(n, 3)); a1 = a0 + rng.normal(size=(n, 3)) * 0.3
    b0 = a0 + rng.normal(size=(n, 3)) * 0.3; b1 = b0 + rng.normal(size=(n, 3)) * 0.3
    a1[:200] = a0[:200]; b1[100:300] = b0[100:300]
    ca, cb = segment_distance(a0, a1, b0, b1)
    s = np.linspace(0, 1, 61)
    A = a0[:, None, None, :] + (a1 - a0)[:, None, None, :] * s[None, :, None, None]
    B = b0[:, None, None, :] + (b1 - b0)[:, None, None, :] * s[None, None, :, None]
    brute = np.linalg.norm(A - B, axis=3).reshape(n, -1).min(1)
    d = np.linalg.norm(ca - cb, axis=1)
    assert (d <= brute + 1e-9).all() and (brute - d).max() < 5e-3          # never worse than the scan, and the scan is close
    lib = OracleEngine(load_model("ant"), 1).lib
    oa, ob = np.zeros((n, 3)), np.zeros((n, 3))
    for i in range(n):
        lib.or_seg_seg_closest(_ptr(a0[i]), _ptr(a1[i]), _ptr(b0[i]), _ptr(b1[i]), _ptr(oa[i]), _ptr(ob[i]))
    np.testing.assert_allclose(oa, ca, atol=1e-12); np.testing.assert_allclose(ob, cb, atol=1e-12)


def test_pair_tables_of_the_humanoid():
    spec, sc = load_model("humanoid"), load_selfcol("humanoid")
    limb, limbs = limb_paths(spec)
    names = [[spec.body_names[b] for b in L] for L in limbs]
    assert names == [["torso", "lower_waist", "pelvis"], ["right_thigh", "right_shin", "right_foot"],
                     ["left_thigh", "left_shin", "left_foot"], ["right_upper_arm", "right_lower_arm"], ["left_upper_arm", "left_lower_arm"]]
    cb = sc["cap_body"]
    assert len(cb) == 19 and len(sc["groups"]) == 13                   # 19 collision geoms; 10 limb pairs + 3 intra-limb
    seen = set()
    for g in sc["groups"]:
        for i, j in g["pairs"]:
            a, b = cb[i], cb[j]
            assert a != b and spec.parent[a] != b and spec.parent[b] != a      # never the same body, never two jointed bodies
            assert limb[a] == limb[g["tip_a"]] and limb[b] == limb[g["tip_b"]]
            seen.add((min(i, j), max(i, j)))
    assert len(seen) == sum(len(g["pairs"]) for g in sc["groups"]) == 137
    # the committed tables come from the reachability analysis with 40 000 sampled poses (tools/compile_models.py): whatever a
    # smaller sample finds within reach must be in them
    for g in self_collision_groups(spec, self_collision_pairs(spec, n_samples=3000, seed=7)):
        assert set(g["pairs"]) <= seen | {(j, i) for i, j in seen}
    # union chains: descending generalised indices, containing both tips' chains
    t = topology(spec)
    sl = selfcol_layout(spec, sc, t)
    for g, ch in zip(sc["groups"], sl["chains"]):
        assert ch == sorted(ch, reverse=True) and set(t["body_chain"][g["tip_a"]]) | set(t["body_chain"][g["tip_b"]]) == set(ch)
    assert sl["pchain"] == 21                                          # leg vs leg: 6 + 6 leg dofs, 3 abdomen dofs, 6 root dofs


def _fixed_base_humanoid(n, **kw):
    spec = dataclasses.replace(load_model("humanoid"), fixed_base=True)
    p = dict(SIM, gravity=(0.0, 0.0, 0.0), ground_z=-100.0)
    p.update(kw)
    e = OracleEngine(spec, n, params=p, sensor_bodies=sensor_bodies("humanoid"), precision="f64", selfcol=load_selfcol("humanoid"))
    return spec, e


def test_arm_pressed_against_the_body_is_held_by_the_contact_force():
    """Known answer: torso fixed in space, no gravity.  A constant shoulder torque swings the right arm into the body; at rest
    every joint is in balance: (net joint force reported as dof_force) + J^T f = 0 with f the self-contact force the oracle
    reports at its contact point, +f on one body and -f on the other.  Interpenetration stays within the solver's slop."""
    spec, e = _fixed_base_humanoid(1)
    sc = load_selfcol("humanoid")
    d0 = spec.dof_names.index("right_shoulder1")
    tau = np.zeros((1, spec.nd)); tau[0, d0] = 12.0; tau[0, d0 + 1] = -12.0
    for it in range(1200):                                              # the arm swings in, bounces on the joint springs, settles
        e.step(tau)
    pi = e.pair_info[0]
    act = np.nonzero(pi[:, 3] >= 0)[0]
    assert len(act) >= 1 and np.abs(e.qd).max() < 2e-3                # at rest, touching
    assert pi[act, 4].min() > -4e-3                                    # penetration < 4 mm
    nv = spec.nd
    gen = np.zeros(nv)
    fsum = 0.0
    for g in act:
        i, j = e.pair_list[int(pi[g, 3])]
        f, x = pi[g, :3], pi[g, 6:9]
        for body, sgn in ((sc["cap_body"][i], 1.0), (sc["cap_body"][j], -1.0)):
            J3 = np.zeros((3, nv))
            e.lib.or_point_jac(C.byref(e.model), _ptr(np.ascontiguousarray(e.state[0])), int(body), _ptr(np.ascontiguousarray(x)), _ptr(J3))
            gen += sgn * (J3.T @ f)
        fsum += np.linalg.norm(f)
    assert fsum > 5.0                                                  # the contact really carries load
    resid = e.dof_force[0] + gen
    assert np.abs(resid).max() < 0.02 * np.abs(e.dof_force[0]).max(), (resid, e.dof_force[0])
    # without self-collision the same torque drives the arm through the body until a joint limit stops it
    e2 = OracleEngine(spec, 1, params=e.params_dict, sensor_bodies=sensor_bodies("humanoid"), precision="f64")
    for it in range(400):
        e2.step(tau)
    assert np.abs(e2.q[0] - e.q[0]).max() > 0.2


def test_self_contacts_resolve_interpenetration_and_keep_momentum():
    """Random joint configurations interpenetrate deeply.  A free-floating Humanoid in zero gravity pushes itself apart.  The
    contact impulses are internal (+f and -f at one point): the total linear momentum, zero at the start, stays zero up to the
    first-order error of the integrator -- the typical drift over a fixed time halves with the step, which it would not if the
    two sides of a contact row did not cancel."""
    spec, sc = load_model("humanoid"), load_selfcol("humanoid")
    n = 32
    rng = np.random.default_rng(4)
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    q = rng.uniform(lo, up, (n, spec.nd))
    drift = []
    for dt in (1 / 120, 1 / 480):
        e = OracleEngine(spec, n, params=dict(SIM, gravity=(0.0, 0.0, 0.0), ground_z=-100.0, dt=dt, substeps=1),
                         sensor_bodies=sensor_bodies("humanoid"), precision="f64", selfcol=sc)
        e.root[:, 2] = 2.0; e.q[:] = q
        e.step(np.zeros((n, spec.nd)))
        pi = e.pair_info
        d0 = np.where(pi[:, :, 3] >= 0, pi[:, :, 4], 0.0).min()
        assert d0 < -0.03 and (pi[:, :, 3] >= 0).any(1).mean() > 0.35       # deep overlaps, many envs touch themselves
        for _ in range(int(round(0.3 / dt)) - 1):
            e.step(np.zeros((n, spec.nd)))
        pi = e.pair_info
        assert np.where(pi[:, :, 3] >= 0, pi[:, :, 4], 0.0).min() > -6e-3   # pushed apart
        assert np.isfinite(e.state).all()
        P = []
        for k in range(n):
            M, _ = e.dynamics(k)
            P.append(np.abs((M @ np.concatenate([e.root[k, 7:13], e.qd[k]]))[:3]).max())
        drift.append(np.median(P))
    assert drift[0] < 0.6 and drift[1] < 0.45 * drift[0], drift


def test_contact_caps_match_the_engine_store():
    """kmax / kpair: first come first served in sphere / group order, the rest is dropped and counted."""
    spec, sc = load_model("humanoid"), load_selfcol("humanoid")
    n = 256
    rng = np.random.default_rng(1)
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    q = rng.uniform(lo, up, (n, spec.nd))
    out = []
    for kp in (0, 1):
        e = OracleEngine(spec, n, params=dict(SIM, substeps=1), sensor_bodies=sensor_bodies("humanoid"), precision="f64", selfcol=sc, kpair=kp)
        e.root[:, 2] = 1.5; e.root[:, 6] = 1.0; e.q[:] = q
        e.step(np.zeros((n, spec.nd)))
        out.append(e.pair_info.copy())
    free, capped = out
    nact = (free[:, :, 3] >= 0).sum(1)
    assert nact.max() >= 3
    np.testing.assert_array_equal((capped[:, :, 3] >= 0).sum(1), np.minimum(nact, 1))
    np.testing.assert_array_equal(capped[:, 0, 5], np.maximum(nact - 1, 0))           # dropped count
    first = np.argmax(free[:, :, 3] >= 0, axis=1)
    has = nact > 0
    np.testing.assert_array_equal(np.argmax(capped[:, :, 3] >= 0, axis=1)[has], first[has])   # the first group in order keeps its slot


def _random_state(spec, n, rng, z_lo, z_hi):
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    root = np.zeros((n, 13))
    root[:, 0:2] = rng.normal(size=(n, 2))
    root[:, 2] = rng.uniform(z_lo, z_hi, n)
    q4 = rng.normal(size=(n, 4)); q4[:, 3] += 3; q4 /= np.linalg.norm(q4, axis=1, keepdims=True)
    root[:, 3:7] = q4
    root[:, 7:13] = rng.normal(size=(n, 6))
    return root, rng.uniform(lo, up, (n, spec.nd)), rng.normal(size=(n, spec.nd)) * 2


def test_host_build_of_engine_core_matches_oracle_with_self_collision():
    """The code the HIP kernel compiles (union-chain rows, lane-varying chain masks, compact slots 12 ground + 3 self contacts)
    against the dense fp64 oracle, from random states in which ~60 % of the envs touch themselves: state within the stated
    5e-4 * scale after one step, impulses, sensor and contact forces per element."""
    import hostsim
    spec, sb, sc = load_model("humanoid"), sensor_bodies("humanoid"), load_selfcol("humanoid")
    lib = hostsim.build(humanoid=True)
    n, nd, nsph, npg = 384, spec.nd, len(spec.sph_body), len(sc["groups"])
    rng = np.random.default_rng(0)
    root, q, qd = _random_state(spec, n, rng, 0.9, 1.6)
    tau = rng.uniform(-60, 60, (n, nd))
    orc = OracleEngine(spec, n, params=SIM, sensor_bodies=sb, precision="f64", selfcol=sc, kmax=12, kpair=3, warm_slots=9)
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    st = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd + 3 * npg), np.float32)
    st[:, :13] = root; st[:, 13:13 + nd] = q; st[:, 13 + nd:13 + 2 * nd] = qd
    out = np.zeros((n, 6 * len(sb) + nd + 3 * nsph + 9 * npg), np.float32)
    p = hostsim.make_params(SIM)
    tau32 = np.ascontiguousarray(tau, np.float32)
    touched = 0
    for it in range(3):
        hostsim.step_selfcol(lib, p, st, tau32, out)
        orc.step(tau)
        scale = max(1.0, np.abs(orc.qd).max())
        e = max(np.abs(st[:, :13] - orc.root).max(), np.abs(st[:, 13:13 + nd] - orc.q).max(),
                np.abs(st[:, 13 + nd:13 + 2 * nd] - orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        pi = orc.pair_info
        touched += (pi[:, :, 3] >= 0).any(1).sum()
        lamp = st[:, 13 + 3 * nd + 3 * nsph:].reshape(n, npg, 3)
        assert np.abs(lamp - orc.lam_pair).max() < 2e-3 * max(1.0, np.abs(orc.lam_pair).max())
        np.testing.assert_array_equal(np.abs(lamp).sum(2) > 0, np.abs(orc.lam_pair).sum(2) > 0)     # the same groups carry load
        pf = out[:, 6 * len(sb) + nd + 3 * nsph:].reshape(n, npg, 9)[:, :, :3]
        assert np.abs(pf - pi[:, :, :3]).max() < 2e-3 * max(1.0, np.abs(pi[:, :, :3]).max())
        assert np.abs(st[:, 13 + 2 * nd:13 + 3 * nd + 3 * nsph] - orc.lam).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        assert np.abs(out[:, :12] - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
    assert touched > 0.3 * 3 * n


@pytest.mark.parametrize("selfcol", [True, False])
def test_limb_wave_sub_step_matches_oracle_in_the_block_order(selfcol):
    """core/engine_mwc.hpp -- the Humanoid on three limb waves + the pair wave: per-wave contact slots, dense self-contact rows that the
    two bodies' waves add their halves to, every wave sweeping its own block, the self contacts a block of their own -- as four host
    threads per env that share one row store and meet at a barrier where the GPU waves meet at s_barrier.  Against the fp64 oracle in
    the block solver order with the same per-wave caps: state within the stated 5e-4 * scale per step, impulses, sensors, self-contact
    impulses per element, the same groups carrying load.  Run twice: the threads' timing must not matter (no race in the exchange)."""
    import hostsim
    from isaacgymenvs_amd.assets.model import solver_blocks
    spec, sb, sc = load_model("humanoid"), sensor_bodies("humanoid"), load_selfcol("humanoid")
    lib = hostsim.build(humanoid=True)
    n, nd, nsph, npg = 192, spec.nd, len(spec.sph_body), len(sc["groups"])
    rng = np.random.default_rng(0)
    root, q, qd = _random_state(spec, n, rng, 0.9, 1.6)
    tau = rng.uniform(-60, 60, (n, nd))
    blocks = solver_blocks(spec, self_collision=selfcol, wave_caps=True)
    assert blocks["nblk"] == 4 and blocks["kmax_blk"] == [4, 4, 3, 0] and sorted(set(blocks["body_block"])) == [0, 1, 2]
    kw = dict(selfcol=sc, kpair=3) if selfcol else {}
    orc = OracleEngine(spec, n, params=SIM, sensor_bodies=sb, precision="f64", solver="blocks", blocks=blocks, **kw)
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    p = hostsim.make_params(SIM)
    tau32 = np.ascontiguousarray(tau, np.float32)
    runs = []
    for rep in range(2):
        st = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd + 3 * npg), np.float32)
        st[:, :13] = root; st[:, 13:13 + nd] = q; st[:, 13 + nd:13 + 2 * nd] = qd
        out = np.zeros((n, 6 * len(sb) + nd + 3 * nsph + 9 * npg), np.float32)
        hist = []
        for it in range(3):
            hostsim.step_mwc(lib, p, st, tau32, out, selfcol=selfcol)
            hist.append((st.copy(), out.copy()))
        runs.append(hist)
    touched = 0
    for it in range(3):
        orc.step(tau)
        st, out = runs[0][it]
        np.testing.assert_array_equal(st, runs[1][it][0])
        np.testing.assert_array_equal(out, runs[1][it][1])
        scale = max(1.0, np.abs(orc.qd).max())
        e = max(np.abs(st[:, :13] - orc.root).max(), np.abs(st[:, 13:13 + nd] - orc.q).max(), np.abs(st[:, 13 + nd:13 + 2 * nd] - orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        assert np.abs(st[:, 13 + 2 * nd:13 + 3 * nd + 3 * nsph] - orc.lam).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        assert np.abs(out[:, :12] - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
        assert np.abs(out[:, 12:12 + nd] - orc.dof_force).max() < 2e-3 * max(1.0, np.abs(orc.dof_force).max())
        if selfcol:
            lamp = st[:, 13 + 3 * nd + 3 * nsph:].reshape(n, npg, 3)
            assert np.abs(lamp - orc.lam_pair).max() < 2e-3 * max(1.0, np.abs(orc.lam_pair).max())
            np.testing.assert_array_equal(np.abs(lamp).sum(2) > 0, np.abs(orc.lam_pair).sum(2) > 0)
            pf = out[:, 6 * len(sb) + nd + 3 * nsph:].reshape(n, npg, 9)[:, :, :3]
            assert np.abs(pf - orc.pair_info[:, :, :3]).max() < 2e-3 * max(1.0, np.abs(orc.pair_info[:, :, :3]).max())
            touched += int((orc.pair_info[:, :, 3] >= 0).any(1).sum())
    assert (not selfcol) or touched > 0.3 * 3 * n


@pytest.mark.parametrize("selfcol,iters", [(True, 4), (True, 3), (False, 4)])
def test_fused_sub_steps_of_the_limb_waves_are_bit_identical_on_the_host(selfcol, iters):
    """SimMWC::substeps_fused (round 4: both sub-steps of a Humanoid control step inside ONE launch) against one call per sub-step: every role
    keeps its state between the sub-steps, the limb roles integrate the trunk and the root redundantly, the pair role gets the new pose through
    the half of the trunk exchange area the last sweep left dead (its parity depends on the sweep count: 3 and 4 sweeps here).  Same arithmetic
    on the same values: every bit of state, impulses, sensors, joint forces and pair forces is the same over several control steps."""
    import hostsim
    spec, sb, sc = load_model("humanoid"), sensor_bodies("humanoid"), load_selfcol("humanoid")
    lib = hostsim.build(humanoid=True)
    n, nd, nsph, npg = 128, spec.nd, len(spec.sph_body), len(sc["groups"])
    rng = np.random.default_rng(11)
    root, q, qd = _random_state(spec, n, rng, 0.9, 1.6)
    tau32 = np.ascontiguousarray(rng.uniform(-60, 60, (n, nd)), np.float32)
    p = hostsim.make_params(dict(SIM, iters=iters))
    st1 = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd + 3 * npg), np.float32)
    st1[:, :13] = root; st1[:, 13:13 + nd] = q; st1[:, 13 + nd:13 + 2 * nd] = qd
    st2 = st1.copy()
    out1 = np.zeros((n, 6 * len(sb) + nd + 3 * nsph + 9 * npg), np.float32)
    out2 = out1.copy()
    d1, d2 = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)
    touched = 0
    for it in range(4):
        hostsim.step_mwc(lib, p, st1, tau32, out1, selfcol=selfcol, dropped=d1)
        hostsim.step_mwc_fused(lib, p, st2, tau32, out2, selfcol=selfcol, dropped=d2)
        assert np.isfinite(st1).all()
        np.testing.assert_array_equal(st1, st2)
        np.testing.assert_array_equal(out1, out2)
        np.testing.assert_array_equal(d1, d2)
        touched += int((np.abs(st1[:, 13 + 3 * nd + 3 * nsph:]).reshape(n, npg, 3).sum(2) > 0).any(1).sum())
    assert (not selfcol) or touched > 0.2 * 4 * n           # the pair role did work with the handed-over poses


@pytest.mark.parametrize("waves", [2, 3])
def test_two_wave_split_of_the_sub_step_is_bit_identical_on_the_host(waves):
    """Sim::substep with role 0 (everything but the self-collision phase) and role 1 (tree pass, factor, self-collision phase) on two
    threads that share the row store and meet at one barrier -- what the GPU runs as two waves of a workgroup -- against the same
    sub-step on one thread: the same arithmetic on the same values, so every output bit is the same."""
    import hostsim
    spec, sb, sc = load_model("humanoid"), sensor_bodies("humanoid"), load_selfcol("humanoid")
    lib = hostsim.build(humanoid=True)
    n, nd, nsph, npg = 96, spec.nd, len(spec.sph_body), len(sc["groups"])
    rng = np.random.default_rng(5)
    root, q, qd = _random_state(spec, n, rng, 0.9, 1.6)
    tau32 = np.ascontiguousarray(rng.uniform(-60, 60, (n, nd)), np.float32)
    st1 = np.zeros((n, 13 + 2 * nd + 3 * nsph + nd + 3 * npg), np.float32)
    st1[:, :13] = root; st1[:, 13:13 + nd] = q; st1[:, 13 + nd:13 + 2 * nd] = qd
    st2 = st1.copy()
    out1 = np.zeros((n, 6 * len(sb) + nd + 3 * nsph + 9 * npg), np.float32)
    out2 = out1.copy()
    p = hostsim.make_params(SIM)
    touched = 0
    for it in range(3):
        hostsim.step_selfcol(lib, p, st1, tau32, out1)
        hostsim.step_selfcol2(lib, p, st2, tau32, out2, waves=waves)
        np.testing.assert_array_equal(st1, st2)
        np.testing.assert_array_equal(out1, out2)
        touched += int((np.abs(st1[:, 13 + 3 * nd + 3 * nsph:]).sum(1) > 0).sum())
    assert touched > 0.1 * 3 * n                      # envs whose self contacts carry load


def test_shadow_hand_asset_filters_leave_no_hand_to_hand_contact():
    """shadow_hand.py:357-358 creates the hand with collision filter -1 = "use asset collision filters set in mjcf loader".  The asset's
    filters (shared.xml:21: every collision geom of the hand is contype 1 / conaffinity 0) let no hand shape accept another one; only
    two 1 mm placeholder boxes at the thumb's joint origins (default class) would.  The engine therefore builds no hand self-collision
    rows, while the Humanoid (filter 0, humanoid.py:194) gets its pair list from geometry."""
    import os
    from isaacgymenvs_amd.assets.model import mjcf_self_collision_filter
    from isaacgymenvs_amd.registry import load_extras
    rec = load_extras("shadow_hand")["self_collision_filter"]
    assert rec["collision_geoms"] == 21
    assert sorted(g for _, g, _ in rec["accepting"]) == ["robot0:V_thbase", "robot0:V_thhub"]
    assert all(ext <= 0.001 for _, _, ext in rec["accepting"])
    assert rec["n_pairs"] == 2 * (21 - 2) + 1                                # each placeholder against the other shapes, and each other
    assert load_selfcol("shadow_hand") is None
    ref = "/root/reference/assets/mjcf"
    if not os.path.isdir(ref):
        return
    flt = mjcf_self_collision_filter(os.path.join(ref, "open_ai_assets/hand/shadow_hand.xml"))
    assert flt["collision_geoms"] == rec["collision_geoms"] and len(flt["pairs"]) == rec["n_pairs"]
    assert all(("V_thbase" in a or "V_thhub" in a or "V_thbase" in b or "V_thhub" in b) for a, b in flt["pairs"])
    hum = mjcf_self_collision_filter(os.path.join(ref, "nv_humanoid.xml"))
    assert len(hum["accepting"]) == hum["collision_geoms"] and len(hum["pairs"]) > 100   # default contype = conaffinity = 1 everywhere

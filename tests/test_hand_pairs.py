"""The Shadow Hand's explicit hand-to-hand contact pairs (MJCF <contact><pair>, reference assets/mjcf/open_ai_assets/hand/shared.xml:31-51) as
compliant contacts: tables from the asset, the model's behaviour (oracle), and both forms of the engine's hand sub-step -- host builds of
csrc/core/hand_engine.hpp (one wave) and hand_engine_mw.hpp (finger per wave) -- against oracle/hand.c, which oracle/hand.py (numpy) cross-checks."""
import ctypes as C
import os

import numpy as np
import pytest

import hostsim  # tests/hostbuild (path added by conftest.py)

SIM = dict(dt=1.0 / 60.0, substeps=2, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.002, rest_offset=0.0,
           max_depen_vel=1000.0, erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-4, warm=0.9)
SHARED = "/root/reference/assets/mjcf/open_ai_assets/hand/shared.xml"


def _load():
    from isaacgymenvs_amd.registry import load_model, load_extras, sensor_bodies
    return load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand")


def test_pair_tables_are_the_assets():
    spec, ex, _ = _load()
    pairs = ex["pairs"]
    assert len(pairs) == 18 and all(p["condim"] == 1 for p in pairs)
    names = {frozenset((p["a"]["geom"], p["b"]["geom"])) for p in pairs}
    assert len(names) == 18                                                   # (shared.xml lists C_lfdistal / C_rfdistal twice: kept once)
    boxes = [p for p in pairs if p["a"]["kind"] == "box"]
    assert len(boxes) == 1 and boxes[0]["a"]["geom"] == "robot0:C_palm0" and boxes[0]["b"]["geom"] == "robot0:C_thdistal"
    assert sum("robot0:C_thdistal" in (p["a"]["geom"], p["b"]["geom"]) for p in pairs) == 8      # shared.xml:32-39: the thumb tip against 7 finger links + the palm
    bn = list(spec.body_names)
    for p in pairs:
        for side in "ab":
            assert bn[p[side]["body"]] == p[side]["geom"].replace("C_", "").replace("palm0", "palm")      # geom C_<body> sits on body <body>
    th = next(p["b"] for p in pairs if p["b"]["geom"] == "robot0:C_thdistal")
    assert np.allclose(th["p0"], [0, 0, 0]) and np.allclose(th["p1"], [0, 0, 0.026]) and abs(th["r"] - 0.00918) < 1e-9     # robot.xml:148: pos 0 0 0.013, size 0.00918 0.013
    if os.path.isfile(SHARED):       # the development container: the committed table is what the asset file says today
        import xml.etree.ElementTree as ET
        listed = [frozenset((p.get("geom1"), p.get("geom2"))) for p in ET.parse(SHARED).getroot().find("contact").findall("pair")]
        assert len(listed) == 19 and set(listed) == names
    # the generated header carries them
    from isaacgymenvs_amd.registry import generate_headers
    hdr = next(h for h in generate_headers() if h.endswith("model_shadow_hand.h"))
    assert "static constexpr int NHP = 18;" in open(hdr).read()


def _engines(N, form, seed, pose=(0.0, 1.0)):
    from oracle.hand import OracleHandEngine
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    spec, ex, sens = _load()
    kw = dict(solver="blocks", blocks=hand_solver_blocks(spec)) if form == "finger_per_wave" else {}
    orc = OracleHandEngine(spec, ex, N, SIM, sens, **kw)
    rng = np.random.default_rng(seed)
    lo, up = orc.lo, orc.up
    orc.q[:] = lo + (up - lo) * rng.uniform(pose[0], pose[1], (N, spec.nd))
    orc.qd[:] = 0.0
    orc.targets[:] = lo + (up - lo) * rng.uniform(0.0, 1.0, (N, spec.nd))
    orc.obj[:, 0:3] = [0.0, 0.0, 5.0]                 # the cube out of reach: hand-to-hand contacts only
    return spec, ex, sens, orc, kw


@pytest.mark.parametrize("form", ["one_wave", "finger_per_wave"])
def test_host_builds_of_both_hand_forms_push_the_same_pairs_as_the_oracle(form):
    """random poses over the whole joint range (two thirds of them have overlapping pair shapes), random drive targets, 8 control steps:
    the engine form's state follows oracle/hand.c with the pairs on (fp32 vs fp64: 2e-4 on positions), the pair sides pushed in env 0 are the
    oracle's, and the same run with the pairs off ends somewhere else entirely (they matter)."""
    from oracle.hand import OracleHandEngine, CUBE_HALF as half, CUBE_MASS as mass, CUBE_INERTIA as inertia
    lib = hostsim.build_hand()
    lib.hs_set_hand_pair_stiffness.argtypes = [C.c_float]
    N = 48
    spec, ex, sens, orc, kw = _engines(N, form, 3)
    nd = spec.nd
    off = OracleHandEngine(spec, ex, N, SIM, sens, **kw)
    off.pair_k = 0.0
    off.q[:] = orc.q; off.qd[:] = 0.0; off.targets[:] = orc.targets; off.obj[:] = orc.obj
    state = np.zeros((N, 4 * nd + 13), np.float32)
    state[:, 0:nd] = orc.q; state[:, 3 * nd:4 * nd] = orc.targets; state[:, 4 * nd:] = orc.obj
    root13 = np.zeros(13, np.float32); root13[:7] = orc.eng.root[0, :7]
    ns = len(orc.sens)
    out = np.zeros((N, 6 * ns + nd + 1), np.float32)
    P = hostsim.make_params(SIM)
    dims = np.zeros(3, np.float32)
    p_ = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    lib.hs_set_hand_pair_stiffness(C.c_float(orc.pair_k))
    seen, fmax = 0, 0.0
    try:
        for it in range(8):
            orc.step(); off.step()
            if form == "one_wave":
                rc = lib.hs_step_hand(C.byref(P), N, p_(state), p_(out), p_(root13), C.c_float(half), C.c_float(mass), C.c_float(inertia), C.c_float(1.0), None, None)
            else:
                rc = lib.hs_step_hand_mw(C.byref(P), N, p_(state), p_(out), p_(root13), C.c_float(half), C.c_float(mass), C.c_float(inertia), C.c_float(1.0),
                                         None, None, 0, p_(dims), p_(dims))
            assert rc == 0
            assert lib.hs_hand_pair_sides() == int(orc.pair_sides[0])
            seen += int((orc.pair_sides > 0).sum())
            np.testing.assert_allclose(state[:, 0:nd], orc.q, atol=2e-4)
            np.testing.assert_allclose(state[:, nd:2 * nd], orc.qd, atol=5e-3 * max(1.0, np.abs(orc.qd).max()))
            # the fingertip force sensors carry the pairs' forces (round 6): the cube is out of reach, so whatever they read IS the hand's own contacts
            fs, fo = out[:, 0:6 * ns], np.asarray(orc.sensor)
            np.testing.assert_allclose(fs, fo, atol=2e-2 * max(1.0, np.abs(fo).max()))
            fmax = max(fmax, float(np.abs(fo[:, [0, 1, 2, 6, 7, 8, 12, 13, 14, 18, 19, 20, 24, 25, 26]]).max()))
        assert seen > 8 * N // 3, "scenario must exercise the pairs"
        assert fmax > 1.0, fmax          # newtons: pressed fingertips do register
        assert np.abs(orc.q - off.q).max() > 0.2                 # rad: without the pairs the fingers pass through each other
    finally:
        lib.hs_set_hand_pair_stiffness(C.c_float(2.0e4))


def test_numpy_and_c_oracles_agree_on_the_pairs():
    """oracle/hand.py (numpy) is the independent restatement of oracle/hand.c's h_pairs(): the same states after six steps from pair-rich poses"""
    from oracle.hand import OracleHandEngine
    N = 6
    spec, ex, sens, orc, _ = _engines(N, "one_wave", 9)
    ref = OracleHandEngine(spec, ex, N, SIM, sens, backend="numpy")
    ref.q[:] = orc.q; ref.qd[:] = 0.0; ref.targets[:] = orc.targets; ref.obj[:] = orc.obj
    hit, fsum = 0, 0.0
    for it in range(6):
        orc.step(); ref.step()
        np.testing.assert_array_equal(orc.pair_sides, ref.pair_sides)
        hit += int(orc.pair_sides.sum())
        np.testing.assert_allclose(orc.q, ref.q, atol=1e-9)
        np.testing.assert_allclose(orc.qd, ref.qd, atol=1e-7)
        np.testing.assert_allclose(orc.sensor, ref.sensor, atol=1e-6 * max(1.0, np.abs(ref.sensor).max()))      # incl. the pairs' forces on the fingertips
        fsum += float(np.abs(ref.sensor).sum())
    assert hit > 0 and fsum > 0


def test_pairs_hold_saturated_drives_within_a_millimetre_or_two():
    """What the compliant pairs are for.  The hand starts open (every pair clear), then every drive is sent to a random target across its range
    with its force limit in place (shared.xml:250-269) for one second: with the pairs on the deepest overlap left once the fingers have come to
    rest is about F / k -- the drives deliver <= 10 N at the links, k = 2e4 N/m: well under 2 mm; with the pairs off the same run leaves links
    most of a radius (> 5 mm) inside each other."""
    from oracle.hand import OracleHandEngine, segment_closest
    spec, ex, sens, _, _ = _engines(1, "one_wave", 0)
    N = 8

    def deepest(eng):
        worst = 0.0
        for e in range(eng.N):
            bp = eng._poses(e)
            for pr in eng.pairs:
                if pr["a"]["kind"] == "box":
                    continue
                (ba, bb) = pr["a"]["body"], pr["b"]["body"]
                Ra, ra, Rb, rb = bp[ba, 3:12].reshape(3, 3), bp[ba, 0:3], bp[bb, 3:12].reshape(3, 3), bp[bb, 0:3]
                ca, cb = segment_closest(ra + Ra @ np.array(pr["a"]["p0"]), ra + Ra @ np.array(pr["a"]["p1"]), rb + Rb @ np.array(pr["b"]["p0"]), rb + Rb @ np.array(pr["b"]["p1"]))
                worst = max(worst, pr["a"]["r"] + pr["b"]["r"] - np.linalg.norm(ca - cb))
        return worst

    res = {}
    for k in (2.0e4, 0.0):
        eng = OracleHandEngine(spec, ex, N, SIM, sens)
        eng.pair_k = k
        rng = np.random.default_rng(21)
        eng.q[:] = np.clip(0.0, eng.lo, eng.up); eng.qd[:] = 0.0
        eng.obj[:, 0:3] = [0.0, 0.0, 5.0]
        assert deepest(eng) <= 1e-9                                     # open hand: nothing overlaps
        eng.targets[:] = eng.lo + (eng.up - eng.lo) * rng.uniform(0.0, 1.0, (N, spec.nd))
        for _ in range(60):
            eng.step()
        res[k] = (deepest(eng), float(np.abs(eng.qd).max()))
    assert res[2.0e4][0] < 2.0e-3 and res[2.0e4][1] < 1.0, res              # at rest, less than 2 mm inside
    assert res[0.0][0] > 5.0e-3, res


def test_mjcf_frame_orientations_follow_mujoco():
    """assets/model.py _mjcf_orientation: `axisangle` (dropped until round 5: the Shadow Hand's thumb base, robot.xml:126, is rotated 0.785 rad
    about y -- at the zero pose the thumb points 45 degrees away from the fingers, not alongside them), `euler` as MuJoCo's intrinsic x-y-z,
    `xyaxes`, `zaxis`, `quat` (wxyz)."""
    from isaacgymenvs_amd.assets.model import _mjcf_orientation, quat_to_mat
    R = lambda **a: quat_to_mat(_mjcf_orientation({k: v for k, v in a.items()}, 1.0))  # noqa: E731
    c, s = np.cos(0.785), np.sin(0.785)
    np.testing.assert_allclose(R(axisangle="0 1 0 0.785"), [[c, 0, s], [0, 1, 0], [-s, 0, c]], atol=1e-12)
    np.testing.assert_allclose(quat_to_mat(_mjcf_orientation(dict(axisangle="0 0 2 90"), np.pi / 180.0)), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)
    a, b, g = 0.3, -0.7, 1.1
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0], [0, 0, 1]])
    np.testing.assert_allclose(R(euler=f"{a} {b} {g}"), Rx @ Ry @ Rz, atol=1e-12)
    np.testing.assert_allclose(R(xyaxes="0 1 0 -1 0 0"), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)
    np.testing.assert_allclose(R(zaxis="0 0 1"), np.eye(3), atol=1e-12)
    np.testing.assert_allclose(R(quat="0.7071067811865476 0 0 0.7071067811865476"), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)
    np.testing.assert_allclose(R(), np.eye(3))
    # the compiled model: the thumb base's frame in its parent's (the palm's)
    spec, _, _ = _load()
    th = list(spec.body_names).index("robot0:thbase")
    np.testing.assert_allclose(quat_to_mat(np.asarray(spec.bquat[th], float)), [[c, 0, s], [0, 1, 0], [-s, 0, c]], atol=1e-6)

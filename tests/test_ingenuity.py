"""Ingenuity on the CPU: the procedural asset, and known answers of the oracle twin of the task (reference isaacgymenvs/tasks/
ingenuity.py) that pin what the GPU parity test then compares the kernels with -- Mars free fall, hover thrust, the periodic
targets, the marker actor, the passive rotor joints."""
import math

import numpy as np
import pytest

from isaacgymenvs_amd.registry import load_model, sensor_bodies
from isaacgymenvs_amd.utils.config import compose

SIM = dict(dt=0.01, substeps=2, iters=6, gravity=(0.0, 0.0, -3.721), contact_offset=0.02, rest_offset=0.001, max_depen_vel=1000.0,
           erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=0.9)


def _params():
    from isaacgymenvs_amd.tasks.ingenuity import ingenuity_params_from_cfg
    return ingenuity_params_from_cfg(compose(overrides=["task=Ingenuity"])["task"])


def _env(n, seed=3):
    from oracle.tasks import OracleIngenuityEnv
    return OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), SIM, _params(), n, seed=seed, precision="f64")


def test_procedural_asset_matches_the_generator_in_the_reference():
    spec = load_model("ingenuity")
    assert list(spec.body_names) == ["chassis", "rotor_physics_0", "rotor_visual_0", "rotor_physics_1", "rotor_visual_1"]   # ingenuity.py:156-229
    assert list(spec.dof_names) == ["rotor_roll0", "rotor_roll0", "rotor_roll1", "rotor_roll1"]                             # 4 dofs (:61)
    # range 0 0 (:200-201): the two locked rotor joints keep their dof slots and are welded in the mass matrix (assets/model.py
    # LOCKED_ARMATURE: armature 100 kg m^2 against a rotor inertia of 8e-3, overdamped centering spring), not held by limit rows
    assert list(spec.dof_limited) == [0, 0, 0, 0] and not np.any(spec.dof_lower) and not np.any(spec.dof_upper)
    assert list(spec.dof_armature) == [100.0, 0.0, 100.0, 0.0] and list(spec.dof_stiffness) == [1000.0, 0.0, 1000.0, 0.0]
    assert sensor_bodies("ingenuity") == [1, 3]                                                                              # forces[:, 1], forces[:, 3] (:347-348)
    box = 0.12 ** 3 * 50.0                                        # chassis box, half extent 0.06, density 50 (:161-166)
    rotor = math.pi * 0.15 ** 2 * 0.01 * 1000.0                   # rotor cylinder radius 0.15, thickness 0.01, density 1000 (:196-199)
    m = np.asarray(spec.body_mass if hasattr(spec, "body_mass") else spec.mass, float)
    assert m[0] == pytest.approx(box, rel=1e-6) and m[1] == pytest.approx(rotor, rel=1e-6) and m[3] == pytest.approx(rotor, rel=1e-6)
    assert m[2] < 0.02 * rotor and m[4] < 0.02 * rotor             # the stand-in blades of the mesh-only bodies are light
    assert len(spec.sph_body) == 0                                 # nothing can reach the ground before the reset at z < 0.5 (:449)


def test_params_are_the_constants_of_the_task_file():
    p = _params()
    assert (p.max_episode_length, p.dt, p.clip_actions) == (2000.0, pytest.approx(0.01), 1.0)      # cfg/task/Ingenuity.yaml
    assert (p.thrust_upper_limit, p.thrust_action_speed_scale, p.target_period) == (2000.0, 2000.0, 500)
    assert p.thrust_lateral_component == pytest.approx(0.2) and p.max_angular_velocity == pytest.approx(4 * math.pi)
    assert p.rotor_speed == 50.0 and p.init_height == 1.0


def test_free_fall_under_mars_gravity_and_hover_thrust():
    n = 4
    env = _env(n)
    for step in range(20):                                         # zero action: zero thrust (:337-345)
        env.step(np.zeros((n, 6), np.float32))
        if step == 0:
            z2 = env.eng.root[:, 2].copy()                         # the reset happened inside this step, then 2 sub-steps of free fall
    # semi-implicit Euler with h = 0.005: v_k = -g h k, z_k = z_0 - g h^2 k (k + 1) / 2 after k sub-steps
    g, h = 3.721, 0.005
    np.testing.assert_allclose(env.eng.root[:, 2] - z2, -g * h * h * (40 * 41 / 2 - 2 * 3 / 2), atol=1e-9)
    np.testing.assert_allclose(env.eng.root[:, 9], -g * h * 40, atol=1e-9)
    # hover: each rotor carries half the weight -> no vertical acceleration
    env2 = _env(n)
    hover = load_model("ingenuity").total_mass() * 3.721 / (2 * 0.01 * 2000.0)
    a = np.zeros((n, 6), np.float32); a[:, 2] = hover; a[:, 5] = hover
    env2.step(a)                                                   # (thrusts are cleared in the step that resets: it falls for 2 sub-steps)
    z1, v1 = env2.eng.root[:, 2].copy(), env2.eng.root[:, 7:13].copy()
    for _ in range(50):
        env2.step(a)
    np.testing.assert_allclose(env2.eng.root[:, 7:13], v1, atol=1e-6)                  # no acceleration any more
    np.testing.assert_allclose(env2.eng.root[:, 2], z1 + v1[:, 2] * 0.5, atol=1e-5)
    assert np.abs(v1[:, [0, 1, 3, 4, 5]]).max() < 1e-9
    assert np.allclose(env2.thrusts[:, :, 2], 0.01 * 2000.0 * hover) and np.allclose(env2.forces[:, [1, 3], 2], env2.thrusts[:, :, 2])


def test_thrust_clamps_and_lateral_fraction():
    env = _env(2)
    a = np.array([[1.0, -1.0, 1.0, 0.1, -0.05, -1.0], [0.3, 0.3, 0.5, 0.0, 0.0, 0.25]], np.float32)
    env.step(np.zeros((2, 6), np.float32))                         # the first step resets every env: thrusts cleared (:350-352)
    assert not env.thrusts.any() and not env.forces.any()
    env.step(a)
    np.testing.assert_allclose(env.thrusts[0, 0], [20.0 * 0.2, -20.0 * 0.2, 20.0], rtol=1e-6)      # dt * clamp(2000 a, +-2000), lateral clamp 0.2
    np.testing.assert_allclose(env.thrusts[0, 1], [-20.0 * 0.1, 20.0 * 0.05, -20.0], rtol=1e-6)    # negative thrust is allowed (:339-340)
    np.testing.assert_allclose(env.thrusts[1, 0], [10.0 * 0.2, 10.0 * 0.2, 10.0], rtol=1e-6)
    np.testing.assert_allclose(env.thrusts[1, 1], [0.0, 0.0, 5.0], rtol=1e-6)
    assert not env.forces[:, [0, 2, 4, 5]].any()


def test_targets_marker_and_reset_draws():
    n = 64
    env = _env(n, seed=11)
    hover = load_model("ingenuity").total_mass() * 3.721 / (2 * 0.01 * 2000.0)
    a = np.zeros((n, 6), np.float32); a[:, 2] = hover; a[:, 5] = hover
    env.step(a)
    # reset_idx: x, y in +-1.5, z in 1 + (-0.2, 1.5) (:305-307); target x, y in +-5, z in (1, 2) (:286-287); marker 0.4 above (:290)
    r = env.eng.root
    assert (np.abs(r[:, 0:2]) <= 1.5 + 1e-6).all() and (r[:, 2] >= 0.8 - 1e-2).all() and (r[:, 2] <= 2.5 + 1e-6).all()
    t_reset = env.target.copy()
    env.step(a)                                                    # progress is 1 by now: no periodic draw (:324)
    t1 = env.target.copy()
    assert (t_reset == t1).all()
    for _ in range(300):
        env.step(a)
    alive = env.progress_buf > 250
    assert alive.mean() > 0.5
    assert (env.target[alive] == t1[alive]).all()                  # unchanged between multiples of 500
    assert (np.abs(env.target[:, 0:2]) <= 5).all() and (env.target[:, 2] >= 1).all() and (env.target[:, 2] <= 2).all()
    np.testing.assert_allclose(env.marker[:, 0:3] - env.target, np.tile([0, 0, 0.4], (n, 1)), atol=1e-6)
    for _ in range(210):                                           # past progress 500: the periodic draw (:324-327)
        env.step(a)
    old = env.progress_buf > 500
    assert old.mean() > 0.3 and (env.target[old] != t1[old]).any(axis=1).all()
    young = (env.progress_buf > 100) & (env.progress_buf < 500)    # reset in between, not yet at 500: still the target of their reset
    np.testing.assert_allclose(env.marker[:, 0:3] - env.target, np.tile([0, 0, 0.4], (n, 1)), atol=1e-6)
    # visual rotors spin at -+50 rad/s (set at reset, nothing brakes them), the locked physics rotors stay put
    np.testing.assert_allclose(env.eng.qd[:, 1], -50.0, atol=1e-3)
    np.testing.assert_allclose(env.eng.qd[:, 3], 50.0, atol=1e-3)
    assert np.abs(env.eng.q[:, [0, 2]]).max() < 1e-3
    # two runs, same seed: identical draws; another env offset: different ones
    from oracle.tasks import OracleIngenuityEnv
    e2 = _env(n, seed=11); e2.step(a)
    assert (e2.target == t_reset).all()
    e3 = OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), SIM, _params(), n, seed=11, env_id_offset=n, precision="f64")
    e3.step(a)
    assert (e3.target != t_reset).any(axis=1).all()


def test_lateral_thrust_tilts_the_craft_and_the_angular_speed_is_clamped():
    n = 2
    env = _env(n)
    a = np.zeros((n, 6), np.float32)
    a[:, 2] = 0.5; a[:, 3] = 1.0; a[:, 5] = 0.5                    # upper rotor (0.025 m above the origin) pushes sideways: a torque about y
    for _ in range(40):
        env.step(a)
    w = np.linalg.norm(env.eng.root[:, 10:13], axis=1)
    assert (w <= 4 * math.pi + 1e-9).all()
    assert np.abs(env.eng.root[:, 11]).max() > 0.05                # it does pitch
    assert np.isfinite(env.obs_buf).all() and env.obs_buf.shape == (n, 13)

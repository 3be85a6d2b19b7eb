"""AllegroHand on the HIP engine (csrc/kernels_allegro_hand*.hip: the hand task template of hand_task_kernels.hpp on the Allegro model) against
the CPU restatement of the task (oracle/tasks.py OracleAllegroHandEnv on oracle/hand.c, fp64 physics) -- reference
isaacgymenvs/tasks/allegro_hand.py.  Same protocol and tolerances as the ShadowHand trajectories in tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from isaacgymenvs_amd.registry import load_extras, load_model
from isaacgymenvs_amd.utils.config import compose

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sim_dict(sp):
    return dict(dt=sp.dt, substeps=sp.substeps, iters=sp.iters, gravity=tuple(sp.gravity), contact_offset=sp.contact_offset,
                rest_offset=sp.rest_offset, max_depen_vel=sp.max_depen_vel, erp=sp.erp, plane_mu=sp.plane_mu,
                ground_z=sp.ground_z, cfm=sp.cfm, warm=sp.warm)


def _make(n, seed, **env_over):
    import isaacgymenvs_amd
    cfg = compose(overrides=["task=AllegroHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["env"].update(env_over)
    return isaacgymenvs_amd.make(seed=seed, task="AllegroHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=cfg)


def _oracle(env, n, seed):
    """the restatement in the solver order the engine runs: option multi_wave != 0 is the finger-per-wave kernel (block sweeps, contacts kept per
    limb: csrc/core/hand_engine_mw.hpp), 0 the one-wave kernel (one Gauss-Seidel sequence)"""
    from oracle.tasks import OracleAllegroHandEnv
    from isaacgymenvs_amd.assets.model import hand_solver_blocks
    order = dict(solver="gs")
    if int(env.engine.get_option("multi_wave")) != 0:
        order = dict(solver="blocks", blocks=hand_solver_blocks(load_model("allegro_hand")))
    return OracleAllegroHandEnv(load_model("allegro_hand"), load_extras("allegro_hand"), [], _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed,
                                control_freq_inv=env.control_freq_inv, **order)


@pytest.mark.parametrize("multi_wave", [0, 32, 64])
@pytest.mark.parametrize("object_type", ["block", "egg", "pen"])
def test_allegro_hand_trajectory_matches_cpu_restatement(object_type, multi_wave):
    """multi_wave 32 / 64: the finger-per-wave kernel (four fingers = four waves, 32 / 64 envs per workgroup; make() picks it) against the
    block solver order of the restatement; 0: the one-wave kernel against one Gauss-Seidel sequence."""
    n, seed = 64, 13
    env = _make(n, seed, objectType=object_type)
    assert env._task_params_struct.object_shape == {"block": 0, "pen": 1, "egg": 2}[object_type]
    assert (env.num_obs, env.num_actions, env.num_states) == (88, 16, 0)
    assert int(env.engine.get_option("multi_wave")) == 32          # what make() selects below 8192 envs
    env.engine.set_option("multi_wave", multi_wave)
    orc = _oracle(env, n, seed)
    g = torch.Generator(device="cpu").manual_seed(7)
    touched = np.zeros(n, bool)
    for step in range(5):      # 5 control steps x controlFrequencyInv 2 x 2 sub-steps = 20 sub-steps (the ShadowHand test: 8 x 1 x 2 = 16)
        a = torch.rand((n, 16), generator=g) * 2 - 1
        obs_d, rew, reset, extras = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        torch.cuda.synchronize()
        obs = env.obs_buf.cpu().numpy()
        assert np.isfinite(obs).all()
        if step == 0:   # identical reset draws: cube pose, hand pose, goal
            np.testing.assert_allclose(env.goal_states.cpu().numpy(), orc.goal_states, atol=1e-6)
        nc = env.engine.tensors["object_contact_count"].cpu().numpy()
        np.testing.assert_array_equal(nc > 0, orc.eng.ncontacts > 0)
        touched |= nc > 0
        d = np.abs(obs - o_obs)
        tol = 5e-3 * (1 + step)
        kin = np.concatenate([d[:, :32], d[:, 48:]], axis=1)       # force-like columns (dof forces x10: 32:48) scale with contact impulses
        ok = kin.max(axis=1) < tol
        assert ok.mean() > 0.9, (step, ok.mean(), kin.max())
        assert np.median(d[:, 32:48].max(axis=1)) < 0.2 * max(1.0, np.abs(o_obs[:, 32:48]).max())
        np.testing.assert_array_equal(reset.cpu().numpy()[ok], o_reset[ok])
        np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], atol=0.05 * (1 + step), rtol=1e-2)
    assert touched.mean() > 0.5                                     # the object does land on the hand within the window
    assert obs_d["obs"].shape == (n, 88) and float(obs_d["obs"].abs().max()) <= 5.0 + 1e-6   # clipObservations 5.0
    assert "consecutive_successes" in extras


@pytest.mark.parametrize("obs_type,nobs", [("full_no_vel", 50), ("full", 72)])
def test_allegro_hand_observation_layouts_asymmetric_states_and_object_forces(obs_type, nobs):
    """allegro_hand.py:441-460 layouts as column subsets of the full state, states_buf = the full state (:462-484), random forces on the
    object (:616-625)."""
    n, seed = 48, 31
    env = _make(n, seed, observationType=obs_type, asymmetric_observations=True, forceScale=1.0, forceProbRange=[0.2, 0.6])
    assert env.num_obs == nobs and env.num_states == 88
    orc = _oracle(env, n, seed)
    g = torch.Generator(device="cpu").manual_seed(3)
    for step in range(6):
        a = torch.rand((n, 16), generator=g) * 2 - 1
        out, rew, reset, _ = env.step(a.to(DEV))
        o_obs, o_rew, o_reset = orc.step(a.numpy())
        obs, st = env.obs_buf.cpu().numpy(), env.states_buf.cpu().numpy()
        assert obs.shape == (n, nobs) and st.shape == (n, 88) and out["states"].shape == (n, 88)
        from isaacgymenvs_amd.tasks.allegro_hand import obs_columns
        np.testing.assert_array_equal(obs, st[:, obs_columns(obs_type)])         # the narrow layout IS columns of the full state
        np.testing.assert_allclose(env.rb_forces_object.cpu().numpy(), orc.rb_forces, atol=1e-5)   # same draws, same decay
        d = np.abs(obs - o_obs)
        ok = d.max(axis=1) < 5e-3 * (1 + step)
        assert ok.mean() > 0.9, (step, ok.mean())
    assert float(np.abs(orc.rb_forces).max()) > 0


def test_allegro_hand_is_bit_reproducible_and_reset_idx_follows_the_reference():
    n = 96
    a, b = _make(n, 5), _make(n, 5)
    g = torch.Generator(device="cpu").manual_seed(1)
    for step in range(12):
        act = (torch.rand((n, 16), generator=g) * 2 - 1).to(DEV)
        a.step(act); b.step(act)
    for name in ("dof_state", "object_state", "obs_buf", "rew_buf", "reset_buf", "cur_targets"):
        assert torch.equal(a.engine.tensors[name], b.engine.tensors[name]), name
    # reset_idx (allegro_hand.py:526-590): hand pose inside the noise interval around 0, targets = the new pose, zero velocities
    ids = torch.tensor([3, 17, 40], device=DEV)
    a.reset_idx(ids)
    torch.cuda.synchronize()
    q = a.shadow_hand_dof_pos[ids].cpu().numpy()
    lo, up = a.shadow_hand_dof_lower_limits.cpu().numpy(), a.shadow_hand_dof_upper_limits.cpu().numpy()
    assert ((q >= 0.2 * lo - 1e-6) & (q <= 0.2 * up + 1e-6)).all()                    # resetDofPosRandomInterval 0.2
    np.testing.assert_array_equal(a.cur_targets[ids].cpu().numpy(), q)
    np.testing.assert_array_equal(a.prev_targets[ids].cpu().numpy(), q)
    assert float(a.shadow_hand_dof_vel[ids].abs().max()) == 0.0                       # resetDofVelRandomInterval 0
    obj = a.object_state[ids].cpu().numpy()
    assert (np.abs(obj[:, 0:3] - np.array([0, -0.2, 0.56])) <= 0.01 + 1e-6).all() and np.abs(obj[:, 7:13]).max() == 0.0
    assert (a.progress_buf[ids] == 0).all() and (a.reset_buf[ids] == 0).all()


def test_allegro_hand_rigid_body_states_match_the_oracle_kinematics():
    """gym.refresh_rigid_body_state_tensor (allegro_hand.py:146,156,413): every body's pose from the dof state, against the oracle's frames."""
    n, seed = 32, 2
    env = _make(n, seed)
    orc = _oracle(env, n, seed)
    g = torch.Generator(device="cpu").manual_seed(5)
    for step in range(3):
        a = torch.rand((n, 16), generator=g) * 2 - 1
        env.step(a.to(DEV)); orc.step(a.numpy())
    env.engine.refresh_rigid_body_states()
    bs = env.engine.tensors["rigid_body_state"].cpu().numpy()
    assert bs.shape == (n, 17, 13)
    orc.eng.eng.q[:] = env.shadow_hand_dof_pos.cpu().numpy(); orc.eng.eng.qd[:] = env.shadow_hand_dof_vel.cpu().numpy()
    for e in (0, 7, 31):
        bp = orc.eng._poses(e)
        np.testing.assert_allclose(bs[e, :, 0:3], bp[:, 0:3], atol=2e-5)


@pytest.mark.parametrize("task", ["AllegroHand", "ShadowHand"])
def test_render_draws_the_hand_by_its_object_contact_spheres_and_the_object(task):
    """VecTask.render (vec_task.py:457-512) on the software viewer for a manipulator: the hand's bodies carry no ground-contact samples, so they are
    drawn by the spheres they collide the OBJECT with (models/*_extras.json), plus the object and the goal (`_viewer_extras`)."""
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=16, sim_device=DEV, rl_device=DEV, headless=True)
    for _ in range(3):
        env.step(torch.zeros((16, env.num_actions), device=DEV))
    img = env.render(mode="rgb_array")
    assert img.shape == (480, 640, 3) and img.dtype == np.uint8
    c = img.reshape(-1, 3).astype(int)
    coloured = (c.max(axis=1) - c.min(axis=1)) > 60
    assert coloured.mean() > 0.004                                       # the camera frames the hand (utils/viewer.py backs off by the bodies' extent)
    yellow = (c[:, 0] - c[:, 2] > 70) & (c[:, 1] - c[:, 2] > 45) & (c[:, 0] >= c[:, 1])      # the object's colour, lit or shaded
    assert yellow.sum() > 15

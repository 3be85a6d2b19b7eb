"""`actor_params.hand.rigid_body_properties.mass` of the ShadowHand with the reference's granularity -- one factor per BODY and env
(vec_task.py:783-828 walks the actor's rigid-body property list; ShadowHand.yaml:104-110) -- in the hand's sub-step kernels: the
`hand_body_mass_scale` tensor, read by the Sim<Scaled<M>> instantiations of both forms while option `hand_body_mass` is on (rounds 2-4: one
factor per env).  The oracle is given the same robot with every link's mass and inertia multiplied by its factor."""
import copy

import numpy as np
import pytest
import torch

from isaacgymenvs_amd.registry import load_extras, load_model, sensor_bodies


def _sim_dict(sp):
    return dict(dt=sp.dt, substeps=sp.substeps, iters=sp.iters, gravity=tuple(sp.gravity), contact_offset=sp.contact_offset, rest_offset=sp.rest_offset,
                max_depen_vel=sp.max_depen_vel, erp=sp.erp, plane_mu=sp.plane_mu, ground_z=sp.ground_z, cfm=sp.cfm, warm=sp.warm)


def _run(device, multi_wave, n=32, steps=6, seed=13):
    import isaacgymenvs_amd
    from oracle.tasks import OracleShadowHandEnv
    env = isaacgymenvs_amd.make(seed=seed, task="ShadowHand", num_envs=n, sim_device=device, rl_device=device, headless=True)
    order = {}
    if device != "cpu":
        env.engine.set_option("multi_wave", multi_wave)
        if multi_wave:
            from isaacgymenvs_amd.assets.model import hand_solver_blocks
            order = dict(solver="blocks", blocks=hand_solver_blocks(load_model("shadow_hand")))
    spec = load_model("shadow_hand")
    rng = np.random.default_rng(3)
    f = rng.uniform(0.5, 1.5, spec.nb)                       # ShadowHand.yaml:106: uniform [0.5, 1.5], scaling
    env.engine.tensors["hand_body_mass_scale"][:] = torch.as_tensor(f, dtype=torch.float32, device=device)
    assert int(env.engine.get_option("hand_body_mass")) == 0
    env.engine.set_option("hand_body_mass", 1)
    assert int(env.engine.get_option("hand_body_mass")) == 1
    heavy = copy.deepcopy(spec)
    heavy.mass = np.asarray(spec.mass, float) * f
    heavy.inertia = np.asarray(spec.inertia, float) * f[:, None]
    ex, sb = load_extras("shadow_hand"), sensor_bodies("shadow_hand")
    orc = OracleShadowHandEnv(heavy, ex, sb, _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, **order)
    plain = OracleShadowHandEnv(spec, ex, sb, _sim_dict(env.sim_params), env._task_params_struct, n, seed=seed, **order)
    g = torch.Generator().manual_seed(7)
    kin = np.r_[0:48, 72:161, 191:211]                        # everything but the force-like columns (joint forces, fingertip force-torques)
    for step in range(steps):
        a = torch.rand((n, 20), generator=g) * 2 - 1
        env.step(a.to(device))
        o_obs, _, _ = orc.step(a.numpy())
        p_obs, _, _ = plain.step(a.numpy())
        obs = env.obs_buf.cpu().numpy()
        d = np.abs(obs - o_obs)[:, kin].max(1)
        assert (d < 5e-3 * (1 + step)).mean() >= 0.95, (step, d.max())
        assert np.median(d) < 2e-4 * (1 + step)
    assert np.abs(o_obs - p_obs)[:, :24].max() > 0.02        # the factors matter: the joint angles differ from the unscaled robot's
    return env


def test_cpu_backend_reads_one_mass_factor_per_body():
    from isaacgymenvs_amd import native
    native.build_cpu()
    _run("cpu", 0)


def test_task_config_draws_one_factor_per_body_and_switches_the_scaled_kernels_in():
    """ShadowHand.yaml's own randomization_params through isaacgymenvs_amd.tasks.base.vec_task.VecTask.apply_randomizations (the vectorised restatement)"""
    import isaacgymenvs_amd
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.utils.config import compose
    native.build_cpu()
    n = 48
    cfg = compose(overrides=["task=ShadowHand"])
    cfg["task"]["env"]["numEnvs"] = n
    cfg["task"]["task"]["randomize"] = True
    np.random.seed(3)
    env = isaacgymenvs_amd.make(seed=1, task="ShadowHand", num_envs=n, sim_device="cpu", rl_device="cpu", headless=True, cfg=cfg)
    env.step(torch.zeros((n, 20)))
    bm = env.engine.tensors["hand_body_mass_scale"].numpy()
    assert int(env.engine.get_option("hand_body_mass")) == 1
    assert (bm >= 0.5 - 1e-6).all() and (bm <= 1.5 + 1e-6).all() and bm.std(0).min() > 0.1 and bm.std(1).min() > 0.1
    assert np.allclose(env.engine.tensors["actor_scale"].numpy()[:, 0], 1.0)
    st = env.get_env_state()
    assert st["hand_body_mass"] is True
    for _ in range(3):
        o, r, _, _ = env.step(torch.zeros((n, 20)))
    assert torch.isfinite(o["obs"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("multi_wave", [64, 32, 0])
def test_hip_kernels_read_one_mass_factor_per_body(multi_wave):
    _run("cuda:0", multi_wave)

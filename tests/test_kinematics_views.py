"""gym.acquire_jacobian_tensor / acquire_mass_matrix_tensor (reference franka_cube_stack.py:388-392,551-552): mi_engine_compute_jacobians /
mi_engine_compute_mass_matrices (include/mi_engine.h; core/engine.hpp Sim::body_jacobian / mass_matrix).  Pinned three ways: J qd equals the
velocities of the rigid-body state tensor; J's joint columns equal finite differences of the body positions; the mass matrix equals the
oracle's dense joint-space inertia (oracle/physics.c or_dynamics, fp64) plus the armatures and gives the oracle's kinetic energy."""
import numpy as np
import pytest
import torch

from isaacgymenvs_amd.registry import load_model, sensor_bodies


def _env(task, n, device, seed=3):
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=seed, task=task, num_envs=n, sim_device=device, rl_device=device, headless=True)
    g = torch.Generator().manual_seed(seed)
    for _ in range(6):      # a few random steps: a generic state with velocities
        env.step((torch.rand((n, env.num_actions), generator=g) * 2 - 1).to(device))
    return env


def _check(task, model, device, n=8):
    env = _env(task, n, device)
    eng, spec = env.engine, load_model(model)
    nb, nd = spec.nb, spec.nd
    off = 0 if spec.fixed_base else 6
    nv = nd + off
    J = eng.compute_jacobians().cpu().numpy().astype(np.float64)
    H = eng.compute_mass_matrices().cpu().numpy().astype(np.float64)
    assert J.shape == (n, nb, 6, nv) and H.shape == (n, nv, nv)
    eng.refresh_rigid_body_states()
    t = eng.tensors
    bs = t["rigid_body_state"].cpu().numpy().astype(np.float64)
    root = t["root_states"].cpu().numpy().astype(np.float64)
    q = t["dof_state"][..., 0].cpu().numpy().astype(np.float64); qd = t["dof_state"][..., 1].cpu().numpy().astype(np.float64)
    v = np.concatenate([root[:, 7:13], qd], axis=1) if off else qd
    # 1. J v == [linear, angular] velocity of every body (the rigid-body state tensor's own velocity block)
    jv = np.einsum("nbkc,nc->nbk", J, v)
    np.testing.assert_allclose(jv, bs[:, :, 7:13], atol=2e-4 * max(1.0, np.abs(bs[:, :, 7:13]).max()))
    # 2. joint columns: d(body position) / dq by central differences through the engine's own forward kinematics
    eps = 1e-2       # fp32 positions tens of metres from the origin (terrain rows) carry ~4e-6 of rounding: 2e-4 in the quotient; truncation ~1e-4
    dof = t["dof_state"]
    for d in range(0, nd, max(1, nd // 6)):
        dof[:, d, 0] += eps; eng.refresh_rigid_body_states(); p1 = t["rigid_body_state"][:, :, 0:3].cpu().numpy().astype(np.float64)
        dof[:, d, 0] -= 2 * eps; eng.refresh_rigid_body_states(); p0 = t["rigid_body_state"][:, :, 0:3].cpu().numpy().astype(np.float64)
        dof[:, d, 0] += eps
        np.testing.assert_allclose((p1 - p0) / (2 * eps), J[:, :, 0:3, off + d], atol=3e-3)
    if off:   # base columns: unit linear velocity moves every body alike; unit angular velocity: omega x (p_body - p_root)
        np.testing.assert_allclose(J[:, :, 0:3, 0:3], np.broadcast_to(np.eye(3), (n, nb, 3, 3)), atol=1e-6)
        np.testing.assert_allclose(J[:, :, 3:6, 3:6], np.broadcast_to(np.eye(3), (n, nb, 3, 3)), atol=1e-6)
        r = bs[:, :, 0:3] - root[:, None, 0:3]
        for k in range(3):
            ek = np.zeros(3); ek[k] = 1.0
            np.testing.assert_allclose(J[:, :, 0:3, 3 + k], np.cross(ek, r), atol=1e-5 * max(1.0, np.abs(root[:, 0:3]).max()))   # r: a difference of fp32 world positions
    # 3. mass matrix: symmetric positive definite, equal to the oracle's dense joint-space inertia + armatures, same kinetic energy
    from oracle.engine import OracleEngine
    from isaacgymenvs_amd.tasks.base.vec_task import VecTask  # noqa: F401
    sp = env.sim_params
    sim = dict(dt=sp.dt, substeps=sp.substeps, iters=sp.iters, gravity=tuple(sp.gravity), contact_offset=sp.contact_offset, rest_offset=sp.rest_offset,
               max_depen_vel=sp.max_depen_vel, erp=sp.erp, plane_mu=sp.plane_mu, ground_z=sp.ground_z, cfm=sp.cfm, warm=sp.warm)
    orc = OracleEngine(spec, n, params=sim, sensor_bodies=sensor_bodies(model), precision="f64")
    orc.root[:] = root; orc.q[:] = q; orc.qd[:] = qd
    arm = np.concatenate([np.zeros(off), np.asarray(spec.dof_armature, float)])
    for e in range(n):
        np.testing.assert_allclose(H[e], H[e].T, atol=0)
        assert np.linalg.eigvalsh(H[e]).min() > 0
        M, _ = orc.dynamics(e)
        ref = M + np.diag(arm)
        np.testing.assert_allclose(H[e], ref, atol=2e-5 * max(1.0, np.abs(ref).max()))
        ke, _ = orc.energy(e)
        assert abs(0.5 * v[e] @ (H[e] - np.diag(arm)) @ v[e] - ke) <= 1e-4 * max(1.0, ke)


@pytest.mark.parametrize("task,model", [("Cartpole", "cartpole"), ("Ant", "ant"), ("Humanoid", "humanoid")])
def test_jacobians_and_mass_matrices_on_the_cpu_backend(task, model):
    _check(task, model, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("task,model", [("Ant", "ant"), ("Humanoid", "humanoid"), ("ShadowHand", "shadow_hand"), ("AllegroHand", "allegro_hand"),
                                        ("AnymalTerrain", "anymal")])
def test_jacobians_and_mass_matrices_on_the_hip_engine(task, model):
    _check(task, model, "cuda:0", n=64)


def test_entry_points_refuse_null_and_wrong_sized_tensors():
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=0, task="Cartpole", num_envs=4, sim_device="cpu", rl_device="cpu", headless=True)
    with pytest.raises(AssertionError):
        env.engine.compute_jacobians(torch.zeros((4, 3, 6, 1)))
    from isaacgymenvs_amd import native
    with pytest.raises(RuntimeError, match="null argument"):
        native.check(env.engine.L.mi_engine_compute_jacobians(env.engine.h, None, None), env.engine.L)


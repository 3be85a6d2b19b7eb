"""The Articulation task: `gym.load_asset` of a robot whose kinematic tree no task of the engine was written for.
The engine steps it through gym.simulate -- efforts, per-dof position drives, ground contacts, force sensors, net contact forces, rigid-body
states, Jacobians / mass matrices -- and the task file keeps its own observation / reward code.

  * the stock library carries mjcf/amp_humanoid.xml as HumanoidAMP configures it (every dof DOF_MODE_POS, amp/humanoid_amp_base.py:219-222):
    engine vs oracle/physics.c (or_step_drive_v: per-dof gains) on the CPU backend here, on the HIP backend with -m gpu;
  * a three-link hopper written by the test: parsed, compiled at run time into a library of its own, stepped, compared with the oracle
    given the same parsed spec;
  * the reference's unmodified tasks/humanoid_amp.py through the `isaacgym` stand-in (where the reference tree is reachable).
"""
import ctypes as C
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import pytest
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MI_REFERENCE_ROOT") or next((p for p in ("/root/reference", os.path.join(_HERE, "..", "ab", "ref_stage"))
                                                    if os.path.isdir(os.path.join(p, "isaacgymenvs", "tasks", "amp"))), "/root/reference")
HAVE_REF = os.path.isfile(os.path.join(REF, "isaacgymenvs", "tasks", "humanoid_amp.py"))

SIM = dict(dt=1 / 60.0, substeps=2, iters=4, gravity=(0.0, 0.0, -9.81), contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0, erp=0.2, plane_mu=1.0,
           ground_z=0.0, cfm=1e-6, warm=0.9)


def _sim_params():
    from isaacgymenvs_amd import native
    p = native.MiSimParams(dt=SIM["dt"], substeps=SIM["substeps"], iters=SIM["iters"], contact_offset=SIM["contact_offset"], rest_offset=SIM["rest_offset"],
                           max_depen_vel=SIM["max_depen_vel"], erp=SIM["erp"], plane_mu=SIM["plane_mu"], ground_z=SIM["ground_z"], cfm=SIM["cfm"], warm=SIM["warm"])
    for i, g in enumerate(SIM["gravity"]):
        p.gravity[i] = g
    return p


def _engine_vs_oracle(device, steps=40):
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.registry import load_model, sensor_bodies
    from oracle.engine import OracleEngine
    if device == "cpu":
        native.build_cpu()
    spec, sens = load_model("articulation"), sensor_bodies("articulation")
    N, nd = 64, spec.nd
    rng = np.random.default_rng(3)
    kp = rng.uniform(50, 600, nd)
    kd = kp / 10
    tp = native.MiArticulationParams()
    for d in range(nd):
        tp.kp[d], tp.kd[d] = kp[d], kd[d]
    root0 = [0, 0, 0.89, 0, 0, 0, 1] + [0] * 6
    for k in range(13):
        tp.init_root[k] = root0[k]
    eng = native.Engine("Articulation", _sim_params(), tp, N, device, seed=1)
    T = eng.tensors
    # the state mi_engine_init_state leaves: the actor at its start pose, joints at zero, targets = joint positions
    assert torch.allclose(T["root_states"].cpu(), torch.tensor(root0).expand(N, 13))
    assert float(T["dof_state"].abs().max()) == 0.0 and float(T["dof_position_targets"].abs().max()) == 0.0
    orc = OracleEngine(spec, N, params=SIM, sensor_bodies=sens, precision="f32")
    orc.root[:] = np.array(root0)
    orc.root[:, 2] += rng.uniform(0, 0.3, N)
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    orc.q[:] = np.clip(rng.uniform(-0.2, 0.2, (N, nd)), lo, up)
    T["root_states"].copy_(torch.tensor(orc.root, dtype=torch.float32))
    T["dof_state"][..., 0].copy_(torch.tensor(orc.q, dtype=torch.float32))
    touched = 0
    for it in range(steps):
        target = np.clip(rng.uniform(-0.5, 0.5, (N, nd)), lo, up)
        tau = rng.uniform(-5, 5, (N, nd))
        T["dof_position_targets"].copy_(torch.tensor(target, dtype=torch.float32))
        T["dof_actuation_force"].copy_(torch.tensor(tau, dtype=torch.float32))
        eng.simulate()
        orc.step_drive_v(tau, kp, kd, target)
        tol = 2e-4 * (1 + it) if it < 25 else 5e-2        # contacts from ~step 25 on: trajectories of a falling humanoid part ways slowly
        np.testing.assert_allclose(T["root_states"].cpu().numpy(), orc.root, atol=tol * 5)
        np.testing.assert_allclose(T["dof_state"][..., 0].cpu().numpy(), orc.q, atol=tol * 5)
        nf = T["net_contact_force"].cpu().numpy()
        touched += int((np.abs(orc.netf).sum(axis=(1, 2)) > 0).sum())
        if it < 25:
            np.testing.assert_allclose(nf, orc.netf, atol=0.5 + 0.02 * np.abs(orc.netf).max())
            np.testing.assert_allclose(T["force_sensor"].cpu().numpy().reshape(N, -1), orc.sensor, atol=0.5 + 0.02 * np.abs(orc.sensor).max())
            np.testing.assert_allclose(T["dof_force"].cpu().numpy(), orc.dof_force, atol=0.5 + 0.02 * np.abs(orc.dof_force).max())
    assert touched > 0                      # the comparison covered contacts
    # rigid-body states against the oracle's forward kinematics; reset_idx back to the start pose
    eng.refresh_rigid_body_states()
    rb = T["rigid_body_state"].cpu().numpy()
    np.testing.assert_allclose(rb[:, 0, :7], T["root_states"].cpu().numpy()[:, :7], atol=1e-5)
    ids = torch.tensor([1, 5], dtype=torch.int64, device=device)
    eng.reset_idx(ids)
    assert torch.allclose(T["root_states"][ids].cpu(), torch.tensor(root0).expand(2, 13))
    assert float(T["dof_state"][ids].abs().max()) == 0.0
    # Jacobian / mass matrix shapes of a floating base (gym: all links, 6 base columns)
    J, H = eng.compute_jacobians(), eng.compute_mass_matrices()
    assert tuple(J.shape) == (N, spec.nb, 6, spec.nv) and tuple(H.shape) == (N, spec.nv, spec.nv)
    Hn = H.cpu().numpy()
    np.testing.assert_allclose(Hn, Hn.transpose(0, 2, 1), atol=1e-4)
    assert np.all(np.linalg.eigvalsh(Hn.astype(np.float64)) > 0)
    assert abs(float(Hn[0, 0, 0]) - spec.total_mass()) < 1e-3
    # step() is not this task's entry point
    with pytest.raises(RuntimeError, match="simulate"):
        eng.step(torch.zeros((N, nd), device=device))


def test_articulation_amp_humanoid_matches_oracle_cpu():
    _engine_vs_oracle("cpu")


@pytest.mark.gpu
def test_articulation_amp_humanoid_matches_oracle_hip():
    _engine_vs_oracle("cuda:0")


HOPPER = """<mujoco model="hopper3">
  <compiler angle="degree" inertiafromgeom="true"/>
  <default><joint armature="0.02" damping="0.5" limited="true"/><geom density="900" friction="1.0 0.5 0.5"/></default>
  <worldbody>
    <body name="trunk" pos="0 0 1.0">
      <freejoint name="root"/>
      <geom name="trunk_geom" type="capsule" fromto="0 0 -0.15 0 0 0.2" size="0.06"/>
      <body name="thigh" pos="0 0 -0.15">
        <joint name="hip" type="hinge" axis="0 1 0" range="-60 60" stiffness="4"/>
        <geom name="thigh_geom" type="capsule" fromto="0 0 0 0 0 -0.35" size="0.045"/>
        <body name="shin" pos="0 0 -0.35">
          <joint name="knee" type="hinge" axis="0 1 0" range="-120 0"/>
          <geom name="shin_geom" type="capsule" fromto="0 0 0 0 0 -0.35" size="0.035"/>
          <body name="foot" pos="0 0 -0.35">
            <joint name="ankle_y" type="hinge" axis="0 1 0" range="-40 40"/>
            <joint name="ankle_x" type="hinge" axis="1 0 0" range="-20 20"/>
            <geom name="foot_geom" type="capsule" fromto="-0.08 0 -0.03 0.14 0 -0.03" size="0.03"/>
          </body>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="hip" gear="80"/><motor joint="knee" gear="80"/><motor joint="ankle_y" gear="30"/><motor joint="ankle_x" gear="30"/>
  </actuator>
</mujoco>
"""


def _hopper(device):
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.assets import runtime
    from oracle.engine import OracleEngine
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "hopper3.xml"), "w") as f:
        f.write(HOPPER)
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.max_depenetration_velocity = 10.0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.default_dof_drive_mode = gymapi.DOF_MODE_NONE
    opts.max_angular_velocity = 0.0
    asset = gym.load_asset(sim, tmp, "hopper3.xml", opts)
    assert asset.generic and asset.task == "Articulation"
    assert gym.get_asset_dof_count(asset) == 4 and gym.get_asset_rigid_body_count(asset) == 4
    assert gym.get_asset_dof_names(asset) == ["hip", "knee", "ankle_y", "ankle_x"]
    assert [p.motor_effort for p in gym.get_asset_actuator_properties(asset)] == [80.0, 80.0, 30.0, 30.0]
    gym.create_asset_force_sensor(asset, gym.find_asset_rigid_body_index(asset, "foot"), gymapi.Transform())
    n = 16
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 4)
        h = gym.create_actor(env, asset, gymapi.Transform(gymapi.Vec3(0, 0, 1.05)), "hopper", i, 1, 0)
        dp = gym.get_asset_dof_properties(asset)
        assert np.all(dp["driveMode"] == gymapi.DOF_MODE_NONE) and dp["stiffness"][0] == 4.0
        # hip stays effort-controlled with its passive spring; knee and ankles become position drives with gains of the task's choosing
        dp["driveMode"][1:] = gymapi.DOF_MODE_POS
        dp["stiffness"][1:] = [120.0, 40.0, 40.0]
        dp["damping"][1:] = [6.0, 2.0, 2.0]
        gym.set_actor_dof_properties(env, h, dp)
    gym.prepare_sim(sim)                     # compiles the hopper's library (cached under isaacgymenvs_amd/_variants/<hash>/)
    assert sim.engine.L is not (native.lib_cpu() if device == "cpu" else native.lib())
    es = asset.engine_spec
    assert list(es.dof_stiffness) == [4.0, 0.0, 0.0, 0.0] and list(es.dof_damping) == [0.5, 0.0, 0.0, 0.0]
    lib_path = runtime.variant_library("articulation", es, sim.device, sensors=[3])
    assert "_variants" in lib_path and sim.engine.L is native.variant_lib(lib_path)
    root = gym.acquire_actor_root_state_tensor(sim)
    dof = gym.acquire_dof_state_tensor(sim)
    sensor = gym.acquire_force_sensor_tensor(sim)
    netf = gym.acquire_net_contact_force_tensor(sim)
    rb = gym.acquire_rigid_body_state_tensor(sim)
    assert tuple(root.shape) == (n, 13) and tuple(dof.shape) == (n * 4, 2) and tuple(sensor.shape) == (n, 6) and tuple(netf.shape) == (n * 4, 3)
    assert tuple(rb.shape) == (n * 4, 13)
    prm = dict(dt=1 / 60.0, substeps=2, iters=4, gravity=(0, 0, -9.81), contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0, erp=0.5, plane_mu=1.0,
               ground_z=0.0, cfm=1e-6, warm=1.0)
    orc = OracleEngine(es, n, params=prm, sensor_bodies=[3], precision="f32")
    orc.root[:] = sim.engine.tensors["root_states"].cpu().numpy()
    orc.q[:] = sim.engine.tensors["dof_state"][..., 0].cpu().numpy()
    kp, kd = np.array([0.0, 120.0, 40.0, 40.0]), np.array([0.0, 6.0, 2.0, 2.0])
    rng = np.random.default_rng(5)
    lo, up = np.minimum(es.dof_lower, es.dof_upper), np.maximum(es.dof_lower, es.dof_upper)
    contact = 0
    for it in range(60):
        target = np.clip(rng.uniform(-0.6, 0.3, (n, 4)), lo, up)
        tau = np.zeros((n, 4))
        tau[:, 0] = rng.uniform(-20, 20, n)
        gym.set_dof_position_target_tensor(sim, torch.tensor(target, dtype=torch.float32, device=sim.device).view(-1))
        gym.set_dof_actuation_force_tensor(sim, torch.tensor(tau, dtype=torch.float32, device=sim.device).view(-1))
        gym.simulate(sim)
        gym.fetch_results(sim, True)
        orc.step_drive_v(tau, kp, kd, target)
        gym.refresh_actor_root_state_tensor(sim)
        gym.refresh_dof_state_tensor(sim)
        gym.refresh_net_contact_force_tensor(sim)
        gym.refresh_force_sensor_tensor(sim)
        tol = 1e-3 * (1 + it)
        np.testing.assert_allclose(root.cpu().numpy(), orc.root, atol=tol)
        np.testing.assert_allclose(dof.view(n, 4, 2)[..., 0].cpu().numpy(), orc.q, atol=tol)
        f = netf.view(n, 4, 3).cpu().numpy()
        contact += int((np.abs(orc.netf[:, 3]).sum(axis=1) > 0).sum())
        np.testing.assert_allclose(f, orc.netf, atol=1.0 + 0.05 * np.abs(orc.netf).max())
        np.testing.assert_allclose(sensor.cpu().numpy(), orc.sensor.reshape(n, 6), atol=1.0 + 0.05 * np.abs(orc.sensor).max())
    assert contact > n                      # the hopper landed on its foot and the comparison covered it
    assert float(root[:, 2].min()) > 0.3    # and did not fall through the ground


def test_new_kinematic_tree_compiles_and_steps_cpu():
    _hopper("cpu")


@pytest.mark.gpu
def test_new_kinematic_tree_compiles_and_steps_hip():
    _hopper("cuda:0")


FRANKA = os.path.join(REF, "assets", "urdf", "franka_description", "robots", "franka_panda_gripper.urdf")


def _franka(device):
    """franka_cube_stack.py's robot (`:189` franka_panda_gripper.urdf: fixed base, 7 arm + 2 finger dofs, collision MESHES and no
    <inertial> anywhere -- masses from the meshes' convex hulls, contact spheres inscribed in them) loaded as the task loads it (`:180-190`,
    `:255-275`: gravity off, arm in effort mode, fingers position drives 5000 / 100) and driven by operational-space control computed from
    gym.acquire_jacobian_tensor / acquire_mass_matrix_tensor (the task's control law, `:601-627`, restated): the end effector converges on
    its target, the fingers open; one simulate() from the same state follows the oracle."""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    from oracle.engine import OracleEngine
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.num_position_iterations, sp.physx.num_velocity_iterations = 8, 1
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.flip_visual_attachments, opts.fix_base_link, opts.collapse_fixed_joints, opts.disable_gravity = True, True, False, True
    opts.thickness, opts.default_dof_drive_mode, opts.use_mesh_materials = 0.001, gymapi.DOF_MODE_EFFORT, True
    asset = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/franka_description/robots/franka_panda_gripper.urdf", opts)
    assert asset.generic and gym.get_asset_dof_count(asset) == 9
    names = gym.get_asset_rigid_body_names(asset)
    assert "panda_hand" in names and "panda_link7" in names and len(names) == gym.get_asset_rigid_body_count(asset)
    spec = asset.spec
    assert spec.fixed_base and 15.0 < spec.total_mass() < 25.0           # hull volumes at 1000 kg / m^3 (the real arm: 18 kg)
    dp = gym.get_asset_dof_properties(asset)
    assert list(dp["effort"][:7]) == [87.0, 87.0, 87.0, 87.0, 12.0, 12.0, 12.0]
    dp["driveMode"][:7], dp["stiffness"][:7], dp["damping"][:7] = gymapi.DOF_MODE_EFFORT, 0.0, 0.0
    dp["driveMode"][7:], dp["stiffness"][7:], dp["damping"][7:] = gymapi.DOF_MODE_POS, 5000.0, 100.0
    n = 8
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 4)
        h = gym.create_actor(env, asset, gymapi.Transform(gymapi.Vec3(-0.45, 0.0, 1.0)), "franka", i, 0, 0)
        gym.set_actor_dof_properties(env, h, dp)
    gym.prepare_sim(sim)                    # compiles the arm's library once (cached)
    assert [sim.engine.get_option(k) for k in ("gravity_x", "gravity_y", "gravity_z")] == [0.0, 0.0, 0.0]
    es = asset.engine_spec
    dof = gym.acquire_dof_state_tensor(sim).view(n, 9, 2)
    rb = gym.acquire_rigid_body_state_tensor(sim).view(n, len(names), 13)
    jac = gym.acquire_jacobian_tensor(sim, "franka")
    mm = gym.acquire_mass_matrix_tensor(sim, "franka")
    assert tuple(jac.shape) == (n, len(names) - 1, 6, 9) and tuple(mm.shape) == (n, 9, 9)     # one Jacobian per LINK gym lists (collapse_fixed_joints off), base link left out
    q0 = torch.tensor([0.0, 0.1963, 0.0, -2.618, 0.0, 2.9416, 0.7854, 0.035, 0.035], device=sim.device)      # the task's default pose (:75-77)
    ds = torch.zeros((n, 9, 2), device=sim.device)
    ds[..., 0] = q0
    gym.set_dof_state_tensor(sim, ds.view(-1, 2))
    gym.set_dof_position_target_tensor(sim, q0.repeat(n, 1).view(-1))
    l7 = names.index("panda_link7")
    e7 = list(spec.body_names).index("panda_link7")
    gym.refresh_rigid_body_state_tensor(sim)
    x0 = rb[:, l7, 0:3].clone()
    goal = x0 + torch.tensor([0.10, 0.10, -0.10], device=sim.device)
    kp, kp_null = 150.0, 10.0
    kd, kd_null = 2.0 * kp ** 0.5, 2.0 * kp_null ** 0.5
    effort = torch.tensor(np.array(dp["effort"][:7], dtype=np.float32), device=sim.device)
    lo, up = np.minimum(es.dof_lower, es.dof_upper), np.maximum(es.dof_lower, es.dof_upper)
    err0 = float((goal - x0).norm(dim=1).max())
    # ---- one simulate() against the oracle from this state: efforts on the arm, position drives on the fingers, no gravity, no contact
    prm = dict(dt=1 / 60.0, substeps=2, iters=9, gravity=(0, 0, 0), contact_offset=0.02, rest_offset=0.0, max_depen_vel=100.0, erp=0.5, plane_mu=1.0,
               ground_z=0.0, cfm=1e-6, warm=1.0)
    orc = OracleEngine(es, n, params=prm, sensor_bodies=[0], precision="f64")
    orc.root[:] = sim.engine.tensors["root_states"].cpu().numpy()
    orc.q[:] = q0.cpu().numpy()
    tau0 = np.zeros((n, 9)); tau0[:, :7] = np.random.default_rng(1).uniform(-5, 5, (n, 7))
    tg0 = np.tile(q0.cpu().numpy(), (n, 1)); tg0[:, 7:] = up[7:]
    kpv, kdv = np.array([0.0] * 7 + [5000.0] * 2), np.array([0.0] * 7 + [100.0] * 2)
    gym.set_dof_actuation_force_tensor(sim, torch.tensor(tau0, dtype=torch.float32, device=sim.device).view(-1))
    gym.set_dof_position_target_tensor(sim, torch.tensor(tg0, dtype=torch.float32, device=sim.device).view(-1))
    gym.simulate(sim)
    orc.step_drive_v(tau0, kpv, kdv, tg0)
    gym.refresh_dof_state_tensor(sim)
    np.testing.assert_allclose(dof[..., 0].cpu().numpy(), orc.q, atol=5e-4)
    np.testing.assert_allclose(dof[..., 1].cpu().numpy(), orc.qd, atol=5e-2)
    assert np.abs(orc.netf).max() == 0.0
    # ---- operational-space control on the engine's Jacobian and mass matrix
    gym.set_dof_state_tensor(sim, ds.view(-1, 2))
    errs = []
    for it in range(150):
        gym.refresh_dof_state_tensor(sim); gym.refresh_rigid_body_state_tensor(sim)
        gym.refresh_jacobian_tensors(sim); gym.refresh_mass_matrix_tensors(sim)
        q, qd = dof[:, :7, 0], dof[:, :7, 1]
        J = jac[:, l7 - 1, :, :7]                                   # [n, 6, 7]: the fixed base link has no row
        Mq = mm[:, :7, :7]
        x, xd = rb[:, l7, 0:3], rb[:, l7, 7:13]
        np.testing.assert_allclose((J @ qd.unsqueeze(-1)).squeeze(-1).cpu().numpy(), xd.cpu().numpy(), atol=2e-3)     # J qd IS the link's twist
        dpose = torch.cat([goal - x, torch.zeros((n, 3), device=sim.device)], dim=1)
        Minv = torch.inverse(Mq)
        Lam = torch.inverse(J @ Minv @ J.transpose(1, 2))          # task-space inertia
        u = J.transpose(1, 2) @ Lam @ (kp * dpose - kd * xd).unsqueeze(-1)
        Jbar = Lam @ J @ Minv
        u_null = Mq @ (kd_null * -qd + kp_null * (q0[:7] - q)).unsqueeze(-1)
        u = u + (torch.eye(7, device=sim.device) - J.transpose(1, 2) @ Jbar) @ u_null
        u = torch.max(torch.min(u.squeeze(-1), effort), -effort)
        tau = torch.zeros((n, 9), device=sim.device); tau[:, :7] = u
        tg = q0.repeat(n, 1).clone(); tg[:, 7:] = torch.tensor(up[7:], dtype=torch.float32, device=sim.device)
        gym.set_dof_actuation_force_tensor(sim, tau.view(-1))
        gym.set_dof_position_target_tensor(sim, tg.view(-1))
        gym.simulate(sim)
        errs.append(float((goal - x).norm(dim=1).max()))
    gym.refresh_dof_state_tensor(sim); gym.refresh_rigid_body_state_tensor(sim)
    assert torch.isfinite(dof).all()
    assert abs(errs[0] - err0) < 1e-3 and errs[-1] < 0.01 and errs[60] < 0.5 * err0, (errs[0], errs[60], errs[-1])
    assert float((dof[:, 7:, 0] - torch.tensor(up[7:], dtype=torch.float32, device=sim.device)).abs().max()) < 2e-3      # fingers open to their stops
    Mn = mm.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(Mn, Mn.transpose(0, 2, 1), atol=1e-4)
    assert np.all(np.linalg.eigvalsh(Mn) > 0)
    # the welded hand rides on link 7 (collapse_fixed_joints False: gym lists it as a body of its own)
    hand = names.index("panda_hand")
    assert float((rb[:, hand, 0:3] - rb[:, l7, 0:3]).norm(dim=1).max()) < 0.2 and float((rb[:, hand, 0:3] - rb[:, l7, 0:3]).norm(dim=1).min()) > 0.05


@pytest.mark.skipif(not os.path.isfile(FRANKA), reason="the reference's franka_description is not reachable")
def test_franka_arm_from_urdf_meshes_runs_osc_cpu():
    _franka("cpu")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isfile(FRANKA), reason="the reference's franka_description is not reachable")
def test_franka_arm_from_urdf_meshes_runs_osc_hip():
    _franka("cuda:0")


KUKA = os.path.join(REF, "assets", "urdf", "kuka_allegro_description", "kuka_allegro_touch_sensor.urdf")


def _kuka_allegro(device):
    """The arm + hand of the reference's allegro_kuka tasks (`tasks/allegro_kuka/allegro_kuka_base.py:559-573`: kuka_allegro_touch_sensor.urdf --
    a 7-dof iiwa arm carrying the 16-dof Allegro hand, all collision shapes MESHES --, fixed base, collapse_fixed_joints, gravity disabled, every
    dof a position drive) through `gym.load_asset`: a kinematic tree no compiled model has, so it becomes the Articulation task's robot (23 dofs,
    spheres inscribed in the meshes' hulls).  One simulate() from the task's default arm pose (`:285`) follows the oracle; driven to targets a
    little away it settles on them; Jacobian and mass-matrix tensors are consistent with the body states.  (SURVEY 8f.4 lists
    `kuka_allegro_description` among the asset formats either side of the path; the tasks' scenes -- table, objects -- are not built.)"""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    from oracle.engine import OracleEngine
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.num_position_iterations, sp.physx.num_velocity_iterations = 8, 0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.fix_base_link, opts.flip_visual_attachments, opts.collapse_fixed_joints, opts.disable_gravity = True, False, True, True
    opts.thickness, opts.angular_damping, opts.linear_damping, opts.default_dof_drive_mode = 0.001, 0.01, 0.01, gymapi.DOF_MODE_POS
    asset = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/kuka_allegro_description/kuka_allegro_touch_sensor.urdf", opts)
    nd = gym.get_asset_dof_count(asset)
    assert asset.generic and nd == 23                                     # allegro_kuka_base.py: 7 arm + 16 hand dofs
    names = gym.get_asset_dof_names(asset)
    assert names[:7] == [f"iiwa7_joint_{k}" for k in range(1, 8)] and sum("index" in x or "middle" in x or "ring" in x or "thumb" in x for x in names) == 16
    spec = asset.spec
    assert spec.fixed_base and 20.0 < spec.total_mass() < 40.0            # the URDF's own inertials: 27 kg of arm + 1.8 kg of hand
    dp = gym.get_asset_dof_properties(asset)
    dp["driveMode"][:] = gymapi.DOF_MODE_POS
    dp["stiffness"][:7], dp["damping"][:7] = 400.0, 40.0
    dp["stiffness"][7:], dp["damping"][7:] = 3.0, 0.1                     # the Allegro hand's gains (allegro_hand.py:256-264)
    n = 4
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 2)
        h = gym.create_actor(env, asset, gymapi.Transform(gymapi.Vec3(0.0, 0.8, 0.0)), "allegro", i, 0, 0)        # allegro_kuka_base.py:604-606
        gym.set_actor_dof_properties(env, h, dp)
    gym.prepare_sim(sim)                    # compiles the robot's library once (cached)
    es = asset.engine_spec
    dof = gym.acquire_dof_state_tensor(sim).view(n, nd, 2)
    body_names = gym.get_asset_rigid_body_names(asset)
    rb = gym.acquire_rigid_body_state_tensor(sim).view(n, len(body_names), 13)
    jac = gym.acquire_jacobian_tensor(sim, "allegro")
    mm = gym.acquire_mass_matrix_tensor(sim, "allegro")
    assert tuple(mm.shape) == (n, nd, nd) and jac.shape[0] == n and jac.shape[2:] == (6, nd)
    q0 = torch.zeros(nd, device=sim.device)
    q0[:7] = torch.tensor([-1.571, 1.571, -0.000, 1.376, -0.000, 1.485, 2.358])           # "pose v1" (:285)
    lo, up = np.minimum(es.dof_lower, es.dof_upper), np.maximum(es.dof_lower, es.dof_upper)
    q0 = torch.max(torch.min(q0, torch.tensor(up, dtype=torch.float32, device=sim.device)), torch.tensor(lo, dtype=torch.float32, device=sim.device))
    ds = torch.zeros((n, nd, 2), device=sim.device)
    ds[..., 0] = q0
    gym.set_dof_state_tensor(sim, ds.view(-1, 2))
    rng = np.random.default_rng(4)
    tg = np.clip(q0.cpu().numpy()[None, :] + rng.uniform(-0.15, 0.15, (n, nd)), lo + 0.02, up - 0.02)
    gym.set_dof_position_target_tensor(sim, torch.tensor(tg, dtype=torch.float32, device=sim.device).view(-1))
    # ---- one simulate() against the oracle: position drives on every dof, no gravity, whatever touches the ground does so in both
    prm = dict(dt=1 / 60.0, substeps=2, iters=8, gravity=(0, 0, 0), contact_offset=0.02, rest_offset=0.0, max_depen_vel=100.0, erp=0.5, plane_mu=1.0,
               ground_z=0.0, cfm=1e-6, warm=1.0)
    orc = OracleEngine(es, n, params=prm, sensor_bodies=[0], precision="f64")
    orc.root[:] = sim.engine.tensors["root_states"].cpu().numpy()
    orc.q[:] = q0.cpu().numpy()
    kpv, kdv = np.asarray(dp["stiffness"], float), np.asarray(dp["damping"], float)
    gym.simulate(sim)
    orc.step_drive_v(np.zeros((n, nd)), kpv, kdv, tg)
    gym.refresh_dof_state_tensor(sim)
    np.testing.assert_allclose(dof[..., 0].cpu().numpy(), orc.q, atol=5e-4)
    np.testing.assert_allclose(dof[..., 1].cpu().numpy(), orc.qd, atol=5e-2)
    # ---- driven on: the joints settle on their targets (the light finger joints within a second, the arm's 400 / 40 drives too)
    for it in range(90):
        gym.simulate(sim)
    gym.refresh_dof_state_tensor(sim); gym.refresh_rigid_body_state_tensor(sim); gym.refresh_jacobian_tensors(sim); gym.refresh_mass_matrix_tensors(sim)
    assert torch.isfinite(dof).all()
    err = np.abs(dof[..., 0].cpu().numpy() - tg)
    assert err[:, :7].max() < 0.02 and err[:, 7:].max() < 0.05, (err[:, :7].max(), err[:, 7:].max())
    Mn = mm.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(Mn, Mn.transpose(0, 2, 1), atol=1e-4)
    assert np.all(np.linalg.eigvalsh(Mn) > 0)
    # J qd is the twist of every link (qd small but non-zero while the drives settle)
    qd = dof[..., 1]
    tw = (jac @ qd[:, None, :, None]).squeeze(-1)                                    # [n, links - 1, 6]
    dyn = [i for i, nm in enumerate(body_names) if nm in list(spec.body_names)][1:]  # gym bodies that are engine bodies, the fixed base excluded
    got = rb[:, dyn, 7:13]
    assert tw.shape[1] == spec.nb - 1
    np.testing.assert_allclose(tw[:, [list(spec.body_names).index(body_names[i]) - 1 for i in dyn]].cpu().numpy(), got.cpu().numpy(), atol=2e-3)


@pytest.mark.skipif(not os.path.isfile(KUKA), reason="the reference's kuka_allegro_description is not reachable")
def test_kuka_allegro_arm_hand_from_urdf_meshes_cpu():
    _kuka_allegro("cpu")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isfile(KUKA), reason="the reference's kuka_allegro_description is not reachable")
def test_kuka_allegro_arm_hand_from_urdf_meshes_hip():
    _kuka_allegro("cuda:0")


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not reachable")
@pytest.mark.parametrize("backend", [pytest.param("cpu", marks=pytest.mark.skipif(torch.cuda.is_available(), reason="the HIP variant runs here")),
                                     pytest.param("hip", marks=[pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X")])])
def test_unmodified_humanoid_amp_task_file_steps(backend):
    """/root/reference/isaacgymenvs/tasks/humanoid_amp.py (with amp/humanoid_amp_base.py, its motion library and poselib) as it is: loads
    mjcf/amp_humanoid.xml -- the Articulation task's stock robot --, switches every dof to DOF_MODE_POS, resets from the run motion clip,
    steps; observations and the AMP observation buffer come from its own jitted functions on the engine's state."""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    if dev == "cpu":
        native.build_cpu()
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")}
    for k in saved:
        del sys.modules[k]
    try:
        shims.install(force=True)
        for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"), ("isaacgymenvs.utils", "isaacgymenvs/utils"),
                          ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
            mod = types.ModuleType(name)
            mod.__path__ = [os.path.join(REF, rel)]
            sys.modules[name] = mod
        m = importlib.import_module("isaacgymenvs.tasks.humanoid_amp")
        assert os.path.samefile(m.__file__, os.path.join(REF, "isaacgymenvs", "tasks", "humanoid_amp.py"))
        cfg = omegaconf_to_dict(compose("config", overrides=["task=HumanoidAMP"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
        n = 32
        cfg["env"]["numEnvs"] = n
        cfg["sim"]["use_gpu_pipeline"] = dev != "cpu"
        env = m.HumanoidAMP(cfg=cfg, rl_device=dev, sim_device=dev, graphics_device_id=-1, headless=True, virtual_screen_capture=False, force_render=False)
        sim = env.sim
        assert sim.asset.generic and sim.asset.task == "Articulation"
        assert sim.engine.L is (native.lib_cpu() if dev == "cpu" else native.lib())          # the stock library: no run-time compile for the default config
        assert env.num_dof == 28 and env.num_bodies == 15 and env.num_actions == 28 and env.num_obs == 105
        kp = np.array([sim.engine._tp.kp[d] for d in range(28)])
        assert kp.min() > 0 and kp.max() == 600.0                                              # the MJCF's joint stiffness became drive gains
        torch.manual_seed(0)
        z0 = env._root_states[:, 2].clone()
        saw_reset = False
        for i in range(60):
            a = torch.rand((n, env.num_actions), device=dev) * 2 - 1
            obs, rew, done, info = env.step(a)
            assert torch.isfinite(obs["obs"]).all() and torch.isfinite(info["amp_obs"]).all()
            assert tuple(info["amp_obs"].shape) == (n, 210)
            saw_reset |= bool(done.any())
        assert saw_reset                                                                       # random actions: the humanoids fall, early termination fires
        assert float((env._root_states[:, 2] - z0).abs().max()) > 0.05
        # the state the task reads is the engine's: rigid-body rows of the welded links ride on their engine bodies
        env.gym.refresh_rigid_body_state_tensor(env.sim)
        assert torch.allclose(env._rigid_body_pos[:, 0], env._root_states[:, :3], atol=1e-5)
        # its position targets reach the engine
        tg = sim.engine.tensors["dof_position_targets"]
        assert float(tg.abs().max()) > 0.1
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")]:
            del sys.modules[k]
        sys.modules.update(saved)

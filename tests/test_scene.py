"""Multi-actor envs: a fixed-base articulated actor in a SCENE of free and static boxes (csrc/core/scene_engine.hpp; include/mi_engine.h MiScene).

What the reference does with them: franka_cube_stack.py:204-233,323-339 creates, in every env, the Franka arm, a table and its stand
(gym.create_box with fix_base_link) and two free cubes; it reads the cubes' rows of the actor root state tensor (:377-386,399-400) and teleports
them at reset through set_actor_root_state_tensor_indexed (:509-511).  Here: the same scene built through the `isaacgym` stand-in, stepped by the
product (CPU backend / HIP) and by oracle/scene.py from identical states; first-principles known answers (cubes at rest on the table and on each
other, the weight carried by the contact rows, a cube pushed by the arm's fingers).  Physics parity against PhysX itself is unpinned (closed)."""
import os

import numpy as np
import pytest
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MI_REFERENCE_ROOT") or next((p for p in ("/root/reference", os.path.join(_HERE, "..", "ab", "ref_stage"))
                                                    if os.path.isdir(os.path.join(p, "assets", "urdf", "franka_description"))), "/root/reference")
FRANKA = os.path.join(REF, "assets", "urdf", "franka_description", "robots", "franka_panda_gripper.urdf")
pytestmark = pytest.mark.skipif(not os.path.isfile(FRANKA), reason="the reference's franka_description is not reachable")

TABLE_Z, TABLE_T, STAND_H = 1.0, 0.05, 0.1           # franka_cube_stack.py:207-223
TOP = TABLE_Z + TABLE_T / 2
SIZE_A, SIZE_B = 0.050, 0.070                        # :225-226
Q0 = [0.0, 0.1963, 0.0, -2.618, 0.0, 2.9416, 0.7854, 0.035, 0.035]       # the task's default arm pose (:75-77)


def _build(device, n=8, finger_damping=100.0):
    """the env of franka_cube_stack.py:180-345, actor by actor"""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.num_position_iterations, sp.physx.num_velocity_iterations = 8, 1
    sp.physx.contact_offset, sp.physx.rest_offset = 0.005, 0.0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.flip_visual_attachments, opts.fix_base_link, opts.collapse_fixed_joints, opts.disable_gravity = True, True, False, True
    opts.thickness, opts.default_dof_drive_mode, opts.use_mesh_materials = 0.001, gymapi.DOF_MODE_EFFORT, True
    franka = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/franka_description/robots/franka_panda_gripper.urdf", opts)
    fixed = gymapi.AssetOptions(); fixed.fix_base_link = True
    table = gym.create_box(sim, 1.2, 1.2, TABLE_T, fixed)
    stand = gym.create_box(sim, 0.2, 0.2, STAND_H, fixed)
    cube_a = gym.create_box(sim, SIZE_A, SIZE_A, SIZE_A, gymapi.AssetOptions())
    cube_b = gym.create_box(sim, SIZE_B, SIZE_B, SIZE_B, gymapi.AssetOptions())
    dp = gym.get_asset_dof_properties(franka)
    dp["driveMode"][:7], dp["stiffness"][:7], dp["damping"][:7] = gymapi.DOF_MODE_EFFORT, 0.0, 0.0
    dp["driveMode"][7:], dp["stiffness"][7:], dp["damping"][7:] = gymapi.DOF_MODE_POS, 5000.0, finger_damping
    T = gymapi.Transform
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 4)
        h = gym.create_actor(env, franka, T(gymapi.Vec3(-0.45, 0.0, TOP + STAND_H)), "franka", i, 0, 0)
        gym.set_actor_dof_properties(env, h, dp)
        gym.create_actor(env, table, T(gymapi.Vec3(0.0, 0.0, TABLE_Z)), "table", i, 1, 0)
        gym.create_actor(env, stand, T(gymapi.Vec3(-0.5, 0.0, TOP + STAND_H / 2)), "table_stand", i, 1, 0)
        ia = gym.create_actor(env, cube_a, T(gymapi.Vec3(-1.0, 0.0, 0.0)), "cubeA", i, 2, 0)
        ib = gym.create_actor(env, cube_b, T(gymapi.Vec3(1.0, 0.0, 0.0)), "cubeB", i, 4, 0)
    gym.prepare_sim(sim)
    assert sim.scene == {1: ("static", 0), 2: ("static", 1), 3: ("free", 0), 4: ("free", 1)} and (ia, ib) == (3, 4)
    assert gym.find_actor_rigid_body_handle(sim.envs[0], ia, "box") == len(franka.body_names) + 2       # env domain: arm links, table, stand, cube A
    return gym, sim, franka, dp


def _oracle(sim, franka, n):
    from oracle.scene import OracleSceneEngine
    es, tp, P = franka.engine_spec, sim.engine._tp, sim.engine._sim
    prm = dict(dt=P.dt, substeps=P.substeps, iters=P.iters, gravity=tuple(P.gravity), contact_offset=P.contact_offset, rest_offset=P.rest_offset,
               max_depen_vel=P.max_depen_vel, erp=P.erp, plane_mu=P.plane_mu, ground_z=P.ground_z, cfm=P.cfm, warm=P.warm)
    sc = tp.scene
    scene = dict(arm_gravity=bool(sc.arm_gravity), arm_mu=float(sc.arm_mu),
                 free=[dict(half=list(sc.free_half[i]), mass=float(sc.free_mass[i]), inertia=list(sc.free_inertia[i]), mu=float(sc.free_mu[i]),
                            pose=list(sc.free_init[i])) for i in range(sc.n_free)],
                 static=[dict(pos=list(sc.static_pos[i]), quat=list(sc.static_quat[i]), half=list(sc.static_half[i]), mu=float(sc.static_mu[i]))
                         for i in range(sc.n_static)])
    orc = OracleSceneEngine(es, n, prm, scene, kp=[tp.kp[d] for d in range(es.nd)], kd=[tp.kd[d] for d in range(es.nd)],
                            drive_vmax=[tp.drive_vmax[d] for d in range(es.nd)])
    orc.root[:] = sim.engine.tensors["root_states"].cpu().numpy()
    return orc


def _place(gym, sim, n, cube_a, cube_b):
    """teleport the cubes the way the task does (franka_cube_stack.py:505-511): write their rows of the root tensor, commit them by index"""
    root = gym.acquire_actor_root_state_tensor(sim).view(n, 5, 13)
    root[:, 3] = torch.as_tensor(cube_a, dtype=torch.float32, device=sim.device)
    root[:, 4] = torch.as_tensor(cube_b, dtype=torch.float32, device=sim.device)
    ids = (torch.arange(n, device=sim.device, dtype=torch.int32).view(n, 1) * 5 + torch.tensor([3, 4], device=sim.device, dtype=torch.int32)).flatten()
    gym.set_actor_root_state_tensor_indexed(sim, root.view(-1, 13), ids, len(ids))
    return root


def _arm_home(gym, sim, n):
    q0 = torch.tensor(Q0, device=sim.device)
    ds = torch.zeros((n, 9, 2), device=sim.device)
    ds[..., 0] = q0
    gym.set_dof_state_tensor(sim, ds.view(-1, 2))
    gym.set_dof_position_target_tensor(sim, q0.repeat(n, 1).view(-1))
    gym.set_dof_actuation_force_tensor(sim, torch.zeros(n * 9, device=sim.device))


def _states(rng, n):
    """cube A and cube B somewhere on (or a little above / tilted over) the table, one env with A stacked on B, one with A dropped from 10 cm"""
    a, b = np.zeros((n, 13)), np.zeros((n, 13))
    a[:, 6] = b[:, 6] = 1.0
    a[:, 0:2] = rng.uniform(-0.2, 0.2, (n, 2)); b[:, 0:2] = a[:, 0:2] + rng.uniform(0.12, 0.2, (n, 2)) * rng.choice([-1, 1], (n, 2))
    a[:, 2], b[:, 2] = TOP + SIZE_A / 2, TOP + SIZE_B / 2
    for s in (a, b):                                       # a yaw for every cube
        yaw = rng.uniform(-np.pi, np.pi, n)
        s[:, 5], s[:, 6] = np.sin(yaw / 2), np.cos(yaw / 2)
    a[0, 0:2] = b[0, 0:2] + [0.004, -0.003]; a[0, 2] = TOP + SIZE_B + SIZE_A / 2           # env 0: A on top of B
    a[1, 2] += 0.10                                                                            # env 1: A falls 10 cm
    ax = np.array([1.0, 0.3, 0.0]); ax /= np.linalg.norm(ax)                                   # env 2: B tilted 0.3 rad, lifted clear of the table, comes down on an edge
    b[2, 3:6], b[2, 6] = ax * np.sin(0.15), np.cos(0.15); b[2, 2] += 0.02
    a[3, 7:10] = [0.4, -0.2, 0.0]                                                              # env 3: A slides (friction stops it)
    return a, b


def _parity(device):
    n = 8
    gym, sim, franka, dp = _build(device, n)
    a, b = _states(np.random.default_rng(0), n)
    _arm_home(gym, sim, n)
    root = _place(gym, sim, n, a, b)
    orc = _oracle(sim, franka, n)
    orc.q[:] = np.array(Q0); orc.targets[:] = np.array(Q0)
    orc.box[:, 0], orc.box[:, 1] = a, b
    sc = sim.engine.tensors["scene_state"]
    np.testing.assert_allclose(sc[:, :2].cpu().numpy(), orc.box, atol=1e-6)
    dof = gym.acquire_dof_state_tensor(sim).view(n, 9, 2)
    worst, touched = 0.0, 0
    for step in range(30):
        gym.simulate(sim)
        orc.step(np.zeros((n, 9)))
        gym.refresh_actor_root_state_tensor(sim); gym.refresh_dof_state_tensor(sim)
        got = root[:, 3:5].cpu().numpy()
        d = np.abs(got - orc.box)
        d[..., 3:7] = np.minimum(d[..., 3:7], np.abs(got[..., 3:7] + orc.box[..., 3:7]))
        # positions / quaternions to 0.3 mm / 1e-3, velocities to 2 cm/s (contact onsets in fp32 against fp64 sit one sub-step apart at worst)
        assert d[..., :7].max() < 1e-3 and d[..., 7:].max() < 5e-2, (step, d[..., :7].max(), d[..., 7:].max())
        worst = max(worst, d[..., :7].max())
        nc = sim.engine.tensors["scene_contacts"].cpu().numpy()
        assert (nc[:, 0] == orc.ncontacts).all(), (step, nc[:, 0], orc.ncontacts)
        touched += int(nc[:, 0].sum())
        np.testing.assert_allclose(dof[..., 0].cpu().numpy(), orc.q, atol=5e-4)
    assert touched > 30 * n * 4 and int(sim.engine.tensors["scene_contacts"][:, 1].sum()) == 0        # every cube rests on >= 4 corners, nothing was refused
    # ---- known answers on the settled scene (0.5 s on): nothing sank into the table, the stack stands, the dropped cube lies on the table,
    # the sliding cube has stopped, the tilted cube lies flat
    for _ in range(60):
        gym.simulate(sim)
        orc.step(np.zeros((n, 9)))
    gym.refresh_actor_root_state_tensor(sim)
    A, Bc = root[:, 3].cpu().numpy(), root[:, 4].cpu().numpy()
    assert np.abs(Bc[:, 2] - (TOP + SIZE_B / 2)).max() < 1.5e-3 and np.abs(A[1:, 2] - (TOP + SIZE_A / 2)).max() < 1.5e-3
    assert abs(A[0, 2] - (TOP + SIZE_B + SIZE_A / 2)) < 2e-3 and np.abs(A[0, 0:2] - Bc[0, 0:2]).max() < 8e-3         # the stack
    assert np.abs(A[:, 7:]).max() < 0.05 and np.abs(Bc[:, 7:]).max() < 0.05
    zc = 1.0 - 2.0 * (Bc[:, 3] ** 2 + Bc[:, 4] ** 2)               # z component of cube B's z axis: the tilted one lies flat again
    assert np.abs(zc - 1.0).max() < 2e-3
    # the weight of every cube is carried by its contact rows (oracle side: sum of the normal forces on side A of the box contacts)
    m = [float(sim.engine._tp.scene.free_mass[i]) for i in range(2)]
    for e in range(n):
        fz = [0.0, 0.0]
        for ia, ib, f in orc.contact_forces[e]:
            if ia >= 0:
                fz[ia] += f[2]
            if ib >= 0:
                fz[ib] -= f[2]
        for i in range(2):
            assert abs(fz[i] - m[i] * 9.81) < 0.03 * m[i] * 9.81 + 1e-3, (e, i, fz[i], m[i] * 9.81)
    assert abs(m[0] - 1000.0 * SIZE_A ** 3) < 1e-6
    # and the product agrees with the oracle on that settled state
    d = np.abs(root[:, 3:5].cpu().numpy()[..., :3] - orc.box[..., :3])
    assert d.max() < 3e-3, d.max()


def test_scene_cubes_on_the_table_follow_the_oracle_cpu():
    _parity("cpu")


@pytest.mark.gpu
def test_scene_cubes_on_the_table_follow_the_oracle_hip():
    _parity("cuda:0")


@pytest.mark.gpu
def test_scene_hip_matches_the_cpu_backend_on_every_env_of_a_ragged_batch():
    """the two PRODUCT backends on the same 1027 scenes (not a multiple of the kernel's 8 envs per workgroup): random cube poses on / above / stacked on
    the table, random arm poses and finger targets, 6 simulate() calls -- the HIP kernel (row store, warm-start table and box work area in LDS) against
    the host build of the same engine source, every env"""
    n = 1027
    rng = np.random.default_rng(5)
    a, b = np.zeros((n, 13)), np.zeros((n, 13))
    a[:, 6] = b[:, 6] = 1.0
    a[:, 0:2] = rng.uniform(-0.25, 0.25, (n, 2)); b[:, 0:2] = a[:, 0:2] + rng.uniform(0.10, 0.2, (n, 2)) * rng.choice([-1, 1], (n, 2))
    a[:, 2] = TOP + SIZE_A / 2 + rng.uniform(0.0, 0.03, n); b[:, 2] = TOP + SIZE_B / 2 + rng.uniform(0.0, 0.03, n)
    for s_ in (a, b):
        ax = rng.normal(size=(n, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        ang = rng.uniform(0.0, 0.4, n)
        s_[:, 3:6], s_[:, 6] = ax * np.sin(ang / 2)[:, None], np.cos(ang / 2)
    st = rng.random(n) < 0.25                                  # a quarter of the envs: A dropped onto B
    a[st, 0:2] = b[st, 0:2] + rng.uniform(-0.01, 0.01, (int(st.sum()), 2)); a[st, 2] = b[st, 2] + SIZE_B / 2 + SIZE_A / 2 + 0.005
    q = np.tile(np.array(Q0), (n, 1)) + rng.uniform(-0.15, 0.15, (n, 9)); q[:, 7:] = rng.uniform(0.0, 0.04, (n, 2))
    tg = q.copy(); tg[:, 7:] = rng.choice([0.0, 0.04], (n, 2))
    tau = np.zeros((n, 9)); tau[:, :7] = rng.uniform(-8, 8, (n, 7))
    out = {}
    for device in ("cpu", "cuda:0"):
        gym, sim, franka, dp = _build(device, n)
        ds = torch.zeros((n, 9, 2), device=sim.device); ds[..., 0] = torch.tensor(q, dtype=torch.float32, device=sim.device)
        gym.set_dof_state_tensor(sim, ds.view(-1, 2))
        gym.set_dof_position_target_tensor(sim, torch.tensor(tg, dtype=torch.float32, device=sim.device).view(-1))
        gym.set_dof_actuation_force_tensor(sim, torch.tensor(tau, dtype=torch.float32, device=sim.device).view(-1))
        root = _place(gym, sim, n, a, b)
        for _ in range(6):
            gym.simulate(sim)
        gym.refresh_actor_root_state_tensor(sim); gym.refresh_dof_state_tensor(sim)
        out[device] = (root[:, 3:5].cpu().numpy().copy(), sim.engine.tensors["dof_state"].cpu().numpy().copy(), sim.engine.tensors["scene_contacts"].cpu().numpy().copy())
    (bc, dc, nc), (bh, dh, nh) = out["cpu"], out["cuda:0"]
    assert np.isfinite(bh).all() and np.isfinite(dh).all()
    d = np.abs(bh - bc)
    d[..., 3:7] = np.minimum(d[..., 3:7], np.abs(bh[..., 3:7] + bc[..., 3:7]))
    # fp32 on both sides, different instruction selection (hardware rcp / rsq / sin / cos on the device): 99 % of the envs to 0.2 mm, nobody beyond 5 mm
    per_env = d[..., :7].reshape(n, -1).max(axis=1)
    assert np.quantile(per_env, 0.99) < 2e-4 and per_env.max() < 5e-3, (np.quantile(per_env, 0.99), per_env.max())
    assert np.abs(dh[..., 0] - dc[..., 0]).max() < 2e-3
    assert (nh[:, 0] == nc[:, 0]).mean() > 0.97 and int(nh[:, 1].sum()) == int(nc[:, 1].sum()) == 0
    assert np.median(nh[:, 0]) >= 8 and nh[:, 0].max() > 12        # most cubes already on 4 corners each (tilted / dropped ones are still coming down); stacks / fingers add contacts


def test_get_rigid_transform_before_and_after_prepare_sim_cpu():
    """gym.get_rigid_transform (franka_cabinet.py:262,308-310 calls it while the envs are being created): the start pose carried down the tree at zero joint
    positions before the engine exists, the rigid-body state tensor's row afterwards -- the same poses while the joints are at zero"""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, False
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.flip_visual_attachments, opts.fix_base_link, opts.collapse_fixed_joints, opts.disable_gravity = True, True, False, True
    opts.thickness, opts.default_dof_drive_mode, opts.use_mesh_materials = 0.001, gymapi.DOF_MODE_EFFORT, True
    franka = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/franka_description/robots/franka_panda_gripper.urdf", opts)
    cube = gym.create_box(sim, SIZE_A, SIZE_A, SIZE_A, gymapi.AssetOptions())
    yaw = gymapi.Quat(0.0, 0.0, float(np.sin(0.3)), float(np.cos(0.3)))
    names = ("panda_link0", "panda_link4", "panda_hand", "panda_leftfinger_tip", "panda_grip_site")
    before = {}
    for i in range(2):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 2)
        h = gym.create_actor(env, franka, gymapi.Transform(gymapi.Vec3(-0.45, 0.1 * i, 1.0), yaw), "franka", i, 0, 0)
        c = gym.create_actor(env, cube, gymapi.Transform(gymapi.Vec3(0.2, 0.0, 1.2)), "cube", i, 1, 0)
        for nm in names:
            t = gym.get_rigid_transform(env, gym.find_actor_rigid_body_handle(env, h, nm))
            before[(i, nm)] = np.array([t.p.x, t.p.y, t.p.z, t.r.x, t.r.y, t.r.z, t.r.w])
        t = gym.get_rigid_transform(env, gym.find_actor_rigid_body_handle(env, c, "box"))
        before[(i, "box")] = np.array([t.p.x, t.p.y, t.p.z, t.r.x, t.r.y, t.r.z, t.r.w])
    assert np.allclose(before[(0, "panda_link0")], [-0.45, 0.0, 1.0, 0.0, 0.0, np.sin(0.3), np.cos(0.3)], atol=1e-6)
    assert np.allclose(before[(1, "box")], [0.2, 0.0, 1.2, 0, 0, 0, 1]) and before[(0, "panda_hand")][2] > 1.5       # the arm stands straight up at q = 0
    gym.prepare_sim(sim)
    ds = torch.zeros((2 * 9, 2))
    gym.set_dof_state_tensor(sim, ds)                  # (reset leaves the fingers at their lower limit 0 and the arm inside its limits: q = 0 where allowed)
    q = gym.acquire_dof_state_tensor(sim).view(2, 9, 2)[0, :, 0].numpy()
    zero = np.abs(q).max() < 1e-6
    for (i, nm), want in before.items():
        env = sim.envs[i]
        k = 1 if nm == "box" else 0
        t = gym.get_rigid_transform(env, gym.find_actor_rigid_body_handle(env, k, nm))
        got = np.array([t.p.x, t.p.y, t.p.z, t.r.x, t.r.y, t.r.z, t.r.w])
        if zero or nm in ("panda_link0", "box"):
            assert np.abs(got[:3] - want[:3]).max() < 1e-4 and min(np.abs(got[3:] - want[3:]).max(), np.abs(got[3:] + want[3:]).max()) < 1e-4, (nm, got, want)


def _ramp(device, angle_deg, n=2):
    """a scene whose second static box is a RAMP: a 0.6 x 0.6 x 0.04 m slab pitched by `angle_deg` about y, with a cube lying on it"""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.num_position_iterations, sp.physx.num_velocity_iterations = 8, 1
    sp.physx.contact_offset, sp.physx.rest_offset = 0.005, 0.0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.flip_visual_attachments, opts.fix_base_link, opts.collapse_fixed_joints, opts.disable_gravity = True, True, False, True
    opts.thickness, opts.default_dof_drive_mode, opts.use_mesh_materials = 0.001, gymapi.DOF_MODE_EFFORT, True
    franka = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/franka_description/robots/franka_panda_gripper.urdf", opts)
    fixed = gymapi.AssetOptions(); fixed.fix_base_link = True
    slab = gym.create_box(sim, 0.6, 0.6, 0.04, fixed)
    cube = gym.create_box(sim, SIZE_A, SIZE_A, SIZE_A, gymapi.AssetOptions())
    for a_ in (slab, cube):                     # shape friction 0.5 on both sides (the way ant.py / anymal_terrain.py set it on their assets)
        pr = gym.get_asset_rigid_shape_properties(a_)
        pr[0].friction = 0.5
        gym.set_asset_rigid_shape_properties(a_, pr)
    th = np.radians(angle_deg)
    q = gymapi.Quat(0.0, float(np.sin(th / 2)), 0.0, float(np.cos(th / 2)))               # pitch about +y: the slab's +x end goes DOWN
    nrm = np.array([np.sin(th), 0.0, np.cos(th)])                                         # its upper face's normal
    c0 = np.array([0.5, 0.0, 1.0])
    pc = c0 + nrm * (0.02 + SIZE_A / 2)                                                   # the cube's centre, resting on the face
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 2)
        gym.create_actor(env, franka, gymapi.Transform(gymapi.Vec3(-1.5, 0.0, 1.0)), "franka", i, 0, 0)       # far away
        gym.create_actor(env, slab, gymapi.Transform(gymapi.Vec3(*c0), q), "ramp", i, 1, 0)
        gym.create_actor(env, cube, gymapi.Transform(gymapi.Vec3(*pc), q), "cube", i, 2, 0)
    gym.prepare_sim(sim)
    root = gym.acquire_actor_root_state_tensor(sim).view(n, 3, 13)
    return gym, sim, root, th, pc


def _known_answers(device):
    """first-principles answers of the box contacts: (1) on a ramp below the friction angle (mu = 0.5: 26.6 degrees) a cube stays put; (2) above it, it
    slides down with a = g (sin th - mu cos th) (mu < 1: the resultant stays inside the cube's base, it does not tip); (3) a cube sliding on the level table
    at v0 stops after v0^2 / (2 mu g); (4) a cube dropped from h reaches the table after sqrt(2 h / g) and does not bounce"""
    g = 9.81
    gym, sim, root, th, pc = _ramp(device, 20.0)
    for _ in range(60):
        gym.simulate(sim)
    gym.refresh_actor_root_state_tensor(sim)
    x = root[:, 2].cpu().numpy()
    assert sim.engine._tp.scene.free_mu[0] == 0.5 and sim.engine._tp.scene.static_mu[0] == 0.5
    assert np.abs(x[:, 0:3] - pc).max() < 1.5e-3 and np.abs(x[:, 7:10]).max() < 0.01, (x[0, :3], pc)          # (1) static friction holds at 20 degrees
    gym, sim, root, th, pc = _ramp(device, 40.0)
    T = 24
    for _ in range(T):
        gym.simulate(sim)
    gym.refresh_actor_root_state_tensor(sim)
    x = root[:, 2].cpu().numpy()
    a = g * (np.sin(th) - 0.5 * np.cos(th))                                                # (2) 2.55 m/s^2 down the slope
    t = T / 60.0
    down = np.array([np.cos(th), 0.0, -np.sin(th)])
    s = (x[:, 0:3] - pc) @ down
    v = x[:, 7:10] @ down
    assert np.abs(v - a * t).max() < 0.05 * a * t and np.abs(s - 0.5 * a * t * t).max() < 0.08 * 0.5 * a * t * t + 1e-3, (s, 0.5 * a * t * t, v, a * t)
    assert np.abs((x[:, 0:3] - pc) @ np.array([np.sin(th), 0.0, np.cos(th)])).max() < 1.5e-3                    # ... staying on the face
    # (3), (4): the table scene
    n = 2
    gym, sim, franka, dp = _build(device, n)
    _arm_home(gym, sim, n)
    a_, b_ = np.zeros((n, 13)), np.zeros((n, 13))
    a_[:, 6] = b_[:, 6] = 1.0
    a_[:, 0:3] = [0.0, -0.3, TOP + SIZE_A / 2]; a_[:, 7] = 0.6                             # slides along +x at 0.6 m/s
    h = 0.12
    b_[:, 0:3] = [0.3, 0.3, TOP + SIZE_B / 2 + h]                                          # dropped from 12 cm
    root = _place(gym, sim, n, a_, b_)
    t_hit, vmax_up = None, 0.0
    for k in range(90):
        gym.simulate(sim)
        gym.refresh_actor_root_state_tensor(sim)
        zb, vzb = float(root[0, 4, 2]), float(root[0, 4, 9])
        if t_hit is None and zb < TOP + SIZE_B / 2 + 2e-3:
            t_hit = (k + 1) / 60.0
        if t_hit is not None:
            vmax_up = max(vmax_up, vzb)
    A = root[:, 3].cpu().numpy()
    d_stop = 0.6 ** 2 / (2 * 1.0 * g)                                                      # 18.3 mm
    assert np.abs(A[:, 0] - d_stop).max() < 0.15 * d_stop + 1e-3 and np.abs(A[:, 7:10]).max() < 5e-3, (A[:, 0], d_stop)
    assert abs(t_hit - np.sqrt(2 * h / g)) < 1.5 / 60.0 and vmax_up < 0.05, (t_hit, np.sqrt(2 * h / g), vmax_up)      # inelastic landing on time


def test_scene_first_principles_known_answers_cpu():
    _known_answers("cpu")


@pytest.mark.gpu
def test_scene_first_principles_known_answers_hip():
    _known_answers("cuda:0")


def _boxes(device, statics, frees, n=2, mu=0.5):
    """a scene of static and free boxes given as (size3, position3, quaternion xyzw); the arm stands far away"""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.num_position_iterations, sp.physx.num_velocity_iterations = 8, 1
    sp.physx.contact_offset, sp.physx.rest_offset = 0.005, 0.0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.flip_visual_attachments, opts.fix_base_link, opts.collapse_fixed_joints, opts.disable_gravity = True, True, False, True
    opts.thickness, opts.default_dof_drive_mode, opts.use_mesh_materials = 0.001, gymapi.DOF_MODE_EFFORT, True
    franka = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/franka_description/robots/franka_panda_gripper.urdf", opts)
    fixed = gymapi.AssetOptions(); fixed.fix_base_link = True

    def asset(size, o):
        a_ = gym.create_box(sim, *[float(x) for x in size], o)
        pr = gym.get_asset_rigid_shape_properties(a_)
        pr[0].friction = mu
        gym.set_asset_rigid_shape_properties(a_, pr)
        return a_
    sa = [asset(sz, fixed) for sz, _, _ in statics]
    fa = [asset(sz, gymapi.AssetOptions()) for sz, _, _ in frees]
    T = lambda p_, q_: gymapi.Transform(gymapi.Vec3(*[float(x) for x in p_]), gymapi.Quat(*[float(x) for x in q_]))  # noqa: E731
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 2)
        gym.create_actor(env, franka, gymapi.Transform(gymapi.Vec3(-2.5, 0.0, 1.0)), "franka", i, 0, 0)
        for k, (a_, (_, p_, q_)) in enumerate(zip(sa, statics)):
            gym.create_actor(env, a_, T(p_, q_), f"static{k}", i, 1, 0)
        for k, (a_, (_, p_, q_)) in enumerate(zip(fa, frees)):
            gym.create_actor(env, a_, T(p_, q_), f"free{k}", i, 2, 0)
    gym.prepare_sim(sim)
    root = gym.acquire_actor_root_state_tensor(sim).view(n, 1 + len(statics) + len(frees), 13)
    return gym, sim, franka, root


def _roll(axis, deg):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    return list(a * np.sin(np.radians(deg) / 2)) + [float(np.cos(np.radians(deg) / 2))]


I4 = [0.0, 0.0, 0.0, 1.0]
S2 = float(np.sqrt(2.0))


def _edge_and_outline_contacts(device):
    """Box contacts that have no corner of one box inside the other (round 6: scene_box_edge, scene_face_crossings, the static boxes' corners):
    (1) EDGE-EDGE: a bar rolled 45 degrees about its long axis x, edge down, dropped across a static bar rolled 45 degrees about its long axis y,
        edge up: it lands on the crossing point of the two edges and stays there (an unstable balance: looked at for 0.4 s, perturbations
        grow with exp(t / 0.06 s) from rounding noise), carried by ONE contact;
    (2) a plank lying flat on two static knife edges: rests on 2 + 2 outline crossings;
    (3) two planks lying crossed: rest on the 4 corners of the overlap rectangle, none of which is a corner of a box;
    (4) a plate on a stand smaller than itself: rests on the stand's 4 corners.
    Known answers: the rest heights (to 1 mm), no velocity, the contact counts; every scene also against oracle/scene.py."""
    z0 = 1.0
    cases = {
        "edge-edge": dict(statics=[((0.06, 0.4, 0.06), (0.5, 0.0, z0), _roll([0, 1, 0], 45))],
                          frees=[((0.4, 0.06, 0.06), (0.5, 0.0, z0 + 0.06 * S2 + 0.01), _roll([1, 0, 0], 45))],
                          rest=z0 + 0.06 * S2, contacts=1, steps=24),
        "knife edges": dict(statics=[((0.06, 0.4, 0.06), (0.38, 0.0, z0), _roll([0, 1, 0], 45)), ((0.06, 0.4, 0.06), (0.62, 0.0, z0), _roll([0, 1, 0], 45))],
                            frees=[((0.4, 0.06, 0.02), (0.5, 0.0, z0 + 0.03 * S2 + 0.01 + 0.01), I4)],
                            rest=z0 + 0.03 * S2 + 0.01, contacts=4, steps=60),
        "crossed planks": dict(statics=[((0.06, 0.4, 0.02), (0.5, 0.0, z0), I4)],
                               frees=[((0.4, 0.06, 0.02), (0.5, 0.0, z0 + 0.02 + 0.01), I4)],
                               rest=z0 + 0.02, contacts=4, steps=60),
        "plate on a stand": dict(statics=[((0.1, 0.1, 0.1), (0.5, 0.0, z0), I4)],
                                 frees=[((0.3, 0.3, 0.02), (0.5, 0.0, z0 + 0.05 + 0.01 + 0.01), I4)],
                                 rest=z0 + 0.05 + 0.01, contacts=4, steps=60),
    }
    for name, c in cases.items():
        n = 2
        gym, sim, franka, root = _boxes(device, c["statics"], c["frees"], n)
        orc = _oracle(sim, franka, n)
        q0 = sim.engine.tensors["dof_state"].cpu().numpy()[..., 0].reshape(n, -1)
        orc.q[:] = q0; orc.targets[:] = q0
        ifree = 1 + len(c["statics"])
        for k in range(c["steps"]):
            gym.simulate(sim)
            orc.step(np.zeros((n, orc.nd)))
            gym.refresh_actor_root_state_tensor(sim)
            got = root[:, ifree].cpu().numpy()
            d = np.abs(got - orc.box[:, 0])
            d[:, 3:7] = np.minimum(d[:, 3:7], np.abs(got[:, 3:7] + orc.box[:, 0, 3:7]))
            assert d[:, :7].max() < 1e-3 and d[:, 7:].max() < 5e-2, (name, k, d[:, :7].max(), d[:, 7:].max())
        x = root[:, ifree].cpu().numpy()
        nc = sim.engine.tensors["scene_contacts"].cpu().numpy()
        assert np.abs(x[:, 2] - c["rest"]).max() < 1e-3, (name, x[:, 2], c["rest"])
        assert np.abs(x[:, 7:10]).max() < 0.02 and np.abs(x[:, 10:13]).max() < 0.2, (name, x[:, 7:13])
        assert np.abs(x[:, 0:2] - np.array(c["frees"][0][1][:2])).max() < 2e-3, (name, x[:, 0:3])
        assert (nc[:, 0] == c["contacts"]).all() and (orc.ncontacts == c["contacts"]).all() and int(nc[:, 1].sum()) == 0, (name, nc, orc.ncontacts)


def test_scene_edge_and_outline_contacts_cpu():
    _edge_and_outline_contacts("cpu")


@pytest.mark.gpu
def test_scene_edge_and_outline_contacts_hip():
    _edge_and_outline_contacts("cuda:0")


def _netf_parity(device):
    """the actor's rows of gym's net contact force tensor in a scene (round 6), against oracle/scene.py: cube A sits between the open fingers of
    the arm at its home pose, the finger drives close on it and hold it (the arm itself, gravity off and without drives, sags a little under the
    cube's weight); 20 steps from identical states"""
    n = 2
    gym, sim, franka, dp = _build(device, n)
    _arm_home(gym, sim, n)
    rb = gym.acquire_rigid_body_state_tensor(sim).view(n, -1, 13)
    gym.simulate(sim)                                               # (one step with the cubes far away: the body states of the home pose)
    gym.refresh_rigid_body_state_tensor(sim)
    names = franka.body_names
    site = rb[0, names.index("panda_grip_site"), 0:3].cpu().numpy()
    _arm_home(gym, sim, n)
    a, b = np.zeros((n, 13)), np.zeros((n, 13))
    a[:, 6] = b[:, 6] = 1.0
    a[:, 0:3] = site; a[1, 2] += 0.004
    b[:, 0:3] = [0.3, 0.3, TOP + SIZE_B / 2]
    root = _place(gym, sim, n, a, b)
    orc = _oracle(sim, franka, n)
    tg = np.array(Q0); tg[7:] = 0.0
    orc.q[:] = np.array(Q0); orc.targets[:] = tg
    orc.box[:, 0], orc.box[:, 1] = a, b
    gym.set_dof_position_target_tensor(sim, torch.tensor(np.tile(tg, (n, 1)), dtype=torch.float32, device=sim.device).view(-1))
    netf = gym.acquire_net_contact_force_tensor(sim).view(n, -1, 3)
    dyn = [franka.spec.body_names.index(nm) for nm in ("panda_leftfinger", "panda_rightfinger")]
    rows = [names.index(nm) for nm in ("panda_leftfinger", "panda_rightfinger")]
    seen = 0.0
    for step in range(20):
        gym.simulate(sim)
        orc.step(np.zeros((n, 9)))
        gym.refresh_net_contact_force_tensor(sim); gym.refresh_actor_root_state_tensor(sim)
        got, want = netf[:, rows].cpu().numpy(), orc.netf[:, dyn]
        assert np.abs(root[:, 3, :3].cpu().numpy() - orc.box[:, 0, :3]).max() < 1e-3
        assert np.abs(got - want).max() < 0.05 * np.abs(want).max() + 0.05, (step, got, want)
        seen = max(seen, float(np.abs(want).max()))
    assert seen > 5.0, seen                                         # the fingers did close on the cube
    others = [i for i in range(netf.shape[1]) if i not in rows]
    assert float(netf[:, others].abs().max()) < 1e-6                # nobody else touches anything


def test_scene_net_contact_forces_follow_the_oracle_cpu():
    _netf_parity("cpu")


@pytest.mark.gpu
def test_scene_net_contact_forces_follow_the_oracle_hip():
    _netf_parity("cuda:0")


def _single_body_urdf_scene(tmp_path, device):
    """a scene whose stage and object come from URDF files the way trifinger.py:1169-1255 loads them (gym.load_asset of a one-link file, not
    gym.create_box): a static plate given as a <box> with a collision origin, a static plate given as a MESH (simulated as its bounding box), a free
    cube with an <inertial> of its own, and a goal marker in another collision group (touches nothing, lives in the stand-in)"""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    if device == "cpu":
        native.build_cpu()
    shims.install(force=True)
    from isaacgym import gymapi
    (tmp_path / "plate.urdf").write_text("""<robot name="plate"><link name="base"/><link name="plate_link">
      <collision><origin xyz="0 0 -0.01" rpy="0 0 0"/><geometry><box size="0.6 0.6 0.02"/></geometry></collision>
      <inertial><mass value="2"/><inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial></link>
      <joint name="j" type="fixed"><parent link="base"/><child link="plate_link"/></joint></robot>""")
    # a 0.2 x 0.1 x 0.04 m block as a mesh whose vertices are NOT centred on the link frame: x in [0, 0.2], y in [-0.05, 0.05], z in [0, 0.04]
    vs = [(x, y, z) for x in (0.0, 0.2) for y in (-0.05, 0.05) for z in (0.0, 0.04)]
    faces = [(1, 2, 4), (1, 4, 3), (5, 8, 6), (5, 7, 8), (1, 5, 6), (1, 6, 2), (3, 4, 8), (3, 8, 7), (1, 3, 7), (1, 7, 5), (2, 6, 8), (2, 8, 4)]
    (tmp_path / "block.obj").write_text("".join(f"v {x} {y} {z}\n" for x, y, z in vs) + "".join(f"f {a} {b} {c}\n" for a, b, c in faces))
    (tmp_path / "block.urdf").write_text("""<robot name="block"><link name="block_link">
      <collision><geometry><mesh filename="block.obj" scale="1 1 1"/></geometry></collision></link></robot>""")
    (tmp_path / "cube.urdf").write_text("""<robot name="cube"><link name="object">
      <collision><origin xyz="0 0 0"/><geometry><box size="0.065 0.065 0.065"/></geometry></collision>
      <inertial><mass value="0.094"/><inertia ixx="6.6e-5" ixy="0" ixz="0" iyy="6.6e-5" iyz="0" izz="6.6e-5"/></inertial></link></robot>""")
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.num_position_iterations, sp.physx.num_velocity_iterations = 8, 1
    sp.physx.contact_offset, sp.physx.rest_offset = 0.005, 0.0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    opts = gymapi.AssetOptions()
    opts.flip_visual_attachments, opts.fix_base_link, opts.collapse_fixed_joints, opts.disable_gravity = True, True, False, True
    opts.thickness, opts.default_dof_drive_mode, opts.use_mesh_materials = 0.001, gymapi.DOF_MODE_EFFORT, True
    franka = gym.load_asset(sim, os.path.join(REF, "assets"), "urdf/franka_description/robots/franka_panda_gripper.urdf", opts)
    fixed = gymapi.AssetOptions(); fixed.fix_base_link = True
    with pytest.warns(UserWarning, match="bounding box"):
        block = gym.load_asset(sim, str(tmp_path), "block.urdf", fixed)
    plate = gym.load_asset(sim, str(tmp_path), "plate.urdf", fixed)
    cube = gym.load_asset(sim, str(tmp_path), "cube.urdf", gymapi.AssetOptions())
    goal = gym.load_asset(sim, str(tmp_path), "cube.urdf", fixed)
    assert gym.get_asset_rigid_body_count(cube) == 1 and gym.get_asset_dof_count(cube) == 0 and gym.get_asset_rigid_shape_count(plate) == 1
    n, z0 = 2, 1.0
    T = lambda x, y, z: gymapi.Transform(gymapi.Vec3(x, y, z))  # noqa: E731
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 2)
        gym.create_actor(env, franka, T(-2.5, 0.0, 1.0), "franka", i, 0, 0)
        gym.create_actor(env, plate, T(0.5, 0.0, z0), "plate", i, 1, 0)                 # its top face is the link frame's z = 0
        gym.create_actor(env, block, T(0.4, 0.0, z0), "block", i, 1, 0)                 # lies on the plate: top face at z0 + 0.04
        gym.create_actor(env, cube, T(0.5, 0.0, z0 + 0.04 + 0.0325 + 0.01), "object", i, 0, 0)       # dropped 1 cm onto the block
        gym.create_actor(env, goal, T(0.5, 0.0, z0 + 0.0325), "goal", i + n, 0, 0)      # inside the block -- in another collision group
    gym.prepare_sim(sim)
    sc = sim.engine._tp.scene
    assert sim.scene == {1: ("static", 0), 2: ("static", 1), 3: ("free", 0)}               # the goal marker is not part of the scene
    assert abs(sc.free_mass[0] - 0.094) < 1e-7 and abs(sc.free_inertia[0][0] - 6.6e-5) < 1e-9      # the file's <inertial>, not density x volume
    assert np.allclose(list(sc.static_pos[0]), [0.5, 0.0, z0 - 0.01]) and np.allclose(list(sc.static_half[0]), [0.3, 0.3, 0.01])
    assert np.allclose(list(sc.static_pos[1]), [0.5, 0.0, z0 + 0.02], atol=1e-6) and np.allclose(list(sc.static_half[1]), [0.1, 0.05, 0.02], atol=1e-6)
    root = gym.acquire_actor_root_state_tensor(sim).view(n, 5, 13)
    for _ in range(60):
        gym.simulate(sim)
    gym.refresh_actor_root_state_tensor(sim)
    x = root.cpu().numpy()
    assert np.abs(x[:, 3, 2] - (z0 + 0.04 + 0.0325)).max() < 1e-3 and np.abs(x[:, 3, 7:]).max() < 0.02, x[:, 3]     # the cube rests on the block
    assert np.allclose(x[:, 4, :3], [0.5, 0.0, z0 + 0.0325]) and np.allclose(x[:, 1, :3], [0.5, 0.0, z0])            # marker and plate rows: the actors' poses


def test_scene_actors_loaded_from_single_body_urdf_files_cpu(tmp_path):
    _single_body_urdf_scene(tmp_path, "cpu")


@pytest.mark.gpu
def test_scene_actors_loaded_from_single_body_urdf_files_hip(tmp_path):
    _single_body_urdf_scene(tmp_path, "cuda:0")


def _ori_err(qd_, q_):
    """rotation vector that takes orientation q_ to qd_ (xyzw): 2 vec(qd * conj(q)), the shorter way round"""
    x1, y1, z1, w1 = qd_.unbind(-1)
    x2, y2, z2, w2 = -q_[:, 0], -q_[:, 1], -q_[:, 2], q_[:, 3]
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    v = torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)
    return 2.0 * v * torch.sign(w).unsqueeze(-1)


class _Osc:
    """operational-space control of the hand link on the engine's Jacobian and mass matrix -- the task's control law (franka_cube_stack.py:601-627)
    restated, with an orientation term that holds the gripper pointing down"""

    def __init__(self, gym, sim, franka, n):
        self.gym, self.sim, self.n = gym, sim, n
        names = franka.body_names
        self.hb, self.site, self.tip = names.index("panda_hand"), names.index("panda_grip_site"), names.index("panda_leftfinger_tip")
        self.rb = gym.acquire_rigid_body_state_tensor(sim).view(n, -1, 13)
        self.dof = gym.acquire_dof_state_tensor(sim).view(n, 9, 2)
        self.jac = gym.acquire_jacobian_tensor(sim, "franka")
        self.mm = gym.acquire_mass_matrix_tensor(sim, "franka")
        self.hand = gym.get_actor_joint_dict(sim.envs[0], 0)["panda_hand_joint"]
        assert self.hand == self.hb - 1 and tuple(self.jac.shape) == (n, len(names) - 1, 6, 9)
        gym.refresh_rigid_body_state_tensor(sim)
        self.q_init = self.rb[:, self.hb, 3:7].clone()
        self.q0 = torch.tensor(Q0, device=sim.device)

    def step(self, goal, fingers):
        gym, sim, n, rb, dof = self.gym, self.sim, self.n, self.rb, self.dof
        gym.refresh_jacobian_tensors(sim); gym.refresh_mass_matrix_tensors(sim); gym.refresh_rigid_body_state_tensor(sim); gym.refresh_dof_state_tensor(sim)
        J, Mq = self.jac[:, self.hand, :, :7], self.mm[:, :7, :7]
        kp, kd = 150.0, 2.0 * 150.0 ** 0.5
        dpose = torch.cat([torch.as_tensor(goal, dtype=torch.float32, device=sim.device) - rb[:, self.hb, 0:3], _ori_err(self.q_init, rb[:, self.hb, 3:7])], dim=1)
        dpose = torch.clamp(dpose, -0.1, 0.1)
        Minv = torch.inverse(Mq)
        Lam = torch.inverse(J @ Minv @ J.transpose(1, 2))
        u = J.transpose(1, 2) @ Lam @ (kp * dpose - kd * rb[:, self.hb, 7:13]).unsqueeze(-1)
        u_null = Mq @ (2.0 * 10.0 ** 0.5 * -dof[:, :7, 1] + 10.0 * (self.q0[:7] - dof[:, :7, 0])).unsqueeze(-1)
        u = u + (torch.eye(7, device=sim.device) - J.transpose(1, 2) @ (Lam @ J @ Minv)) @ u_null
        tau = torch.zeros((n, 9), device=sim.device)
        tau[:, :7] = torch.clamp(u.squeeze(-1), -80.0, 80.0)
        tg = self.q0.repeat(n, 1).clone()
        tg[:, 7:] = fingers
        gym.set_dof_actuation_force_tensor(sim, tau.view(-1)); gym.set_dof_position_target_tensor(sim, tg.view(-1))
        gym.simulate(sim)


def _push(device):
    """the closed gripper is lowered beside cube A and swept through it: the cube is pushed along the table (actor sphere vs free box rows) and stays
    on it; then the gripper is pressed down onto the table top and stops there (actor sphere vs static box rows)"""
    n = 4
    gym, sim, franka, dp = _build(device, n)
    _arm_home(gym, sim, n)
    a, b = np.zeros((n, 13)), np.zeros((n, 13))
    a[:, 6] = b[:, 6] = 1.0
    b[:, 0:3] = [0.3, 0.3, TOP + SIZE_B / 2]
    a[:, 0:3] = [0.078, 0.0, TOP + SIZE_A / 2]
    root = _place(gym, sim, n, a, b)
    osc = _Osc(gym, sim, franka, n)
    for step in range(300):
        osc.step([-0.005 + 0.15 * min(max(step - 100, 0), 150) / 150.0, 0.0, TOP + 0.125], 0.0)      # finger tips 2 cm above the table
    gym.refresh_actor_root_state_tensor(sim)
    A = root[:, 3].cpu().numpy()
    assert np.isfinite(A).all() and np.abs(A[:, 2] - (TOP + SIZE_A / 2)).max() < 4e-3        # still lying on the table
    assert (A[:, 0] > 0.078 + 0.08).all() and np.abs(A[:, 1]).max() < 0.03, A[:, :3]           # pushed 10 cm along +x by the finger tips
    tip_x = float(osc.rb[0, osc.tip, 0])
    assert abs(float(A[0, 0]) - tip_x - 0.033) < 0.01                                          # and it sits against them: half a cube + half a finger ahead
    assert int(sim.engine.tensors["scene_contacts"][:, 0].max()) > 8 + 4
    for step in range(120):                                                                    # press down: the table holds the gripper
        osc.step([0.14, 0.0, TOP + 0.05], 0.0)
    gym.refresh_rigid_body_state_tensor(sim)
    assert float(osc.rb[:, osc.tip, 2].min()) > TOP - 4e-3 and float(osc.rb[:, osc.tip, 2].max()) < TOP + 0.02
    assert torch.isfinite(osc.dof).all()


def _undamped_drive(device):
    """a DOF_MODE_POS dof with a URDF velocity limit, kp > 0 and kd = 0 inside a scene.  The drive's error clamp is `vmax kd / kp` -- the error
    at which spring and damper balance at the limit speed --, which is 0 without a damper: the target collapsed onto q every sub-step and the finger
    never moved.  Such a drive is now bounded by the clamp of the solved joint velocity alone: the fingers close at the URDF's 0.2 m/s
    (franka_panda_gripper.urdf:247) and arrive."""
    n = 4
    gym, sim, franka, dp = _build(device, n, finger_damping=0.0)
    _arm_home(gym, sim, n)
    a, b = np.zeros((n, 13)), np.zeros((n, 13))
    a[:, 6] = b[:, 6] = 1.0
    a[:, 0:3] = [0.3, -0.3, TOP + SIZE_A / 2]; b[:, 0:3] = [0.3, 0.3, TOP + SIZE_B / 2]        # both cubes out of the gripper's way
    _place(gym, sim, n, a, b)
    dof = gym.acquire_dof_state_tensor(sim).view(n, 9, 2)
    tg = torch.tensor(Q0, device=sim.device).repeat(n, 1).clone()
    tg[:, 7:] = 0.005                                                      # close the fingers from 35 mm to 5 mm
    gym.set_dof_position_target_tensor(sim, tg.view(-1))
    vmax = 0.0
    for step in range(30):
        gym.simulate(sim)
        gym.refresh_dof_state_tensor(sim)
        vmax = max(vmax, float(dof[:, 7:, 1].abs().max()))
        if step == 2:
            assert float(dof[:, 7:, 0].max()) < 0.035 - 0.5 * 3 * 0.2 / 60.0, dof[0, 7:, 0]      # they do move, at about the limit speed
    q = dof[:, 7:, 0].cpu().numpy()
    assert np.abs(q - 0.005).max() < 2e-3 and vmax <= 0.2 + 1e-3, (q, vmax)


def test_scene_position_drive_without_a_damper_moves_cpu():
    _undamped_drive("cpu")


@pytest.mark.gpu
def test_scene_position_drive_without_a_damper_moves_hip():
    _undamped_drive("cuda:0")


def test_scene_arm_pushes_a_cube_cpu():
    _push("cpu")


@pytest.mark.gpu
def test_scene_arm_pushes_a_cube_hip():
    _push("cuda:0")


def _grasp(device):
    """what the task is about (franka_cube_stack.py:697-752: lift cube A, carry it over cube B, put it down): open gripper over cube A, down, close
    (the finger drives are velocity limited, URDF :247), lift -- the cube comes up between the fingers, held by friction alone; carried over cube B
    and released, it stays on top of it"""
    n = 4
    gym, sim, franka, dp = _build(device, n)
    assert abs(sim.engine._tp.drive_vmax[7] - 0.2) < 1e-6 and 2.0 < sim.engine._tp.drive_vmax[0] < 2.7      # franka_panda_gripper.urdf: fingers 0.2 m/s, arm joints 2.175 .. 2.61 rad/s
    _arm_home(gym, sim, n)
    a, b = np.zeros((n, 13)), np.zeros((n, 13))
    a[:, 6] = b[:, 6] = 1.0
    a[:, 0:3] = [0.058, 0.0, TOP + SIZE_A / 2]
    b[:, 0:3] = [0.058, 0.16, TOP + SIZE_B / 2]
    root = _place(gym, sim, n, a, b)
    osc = _Osc(gym, sim, franka, n)
    hx = 0.058 - 0.013                                                 # hand origin over the cube: the grip site sits 13 mm ahead of it at this pose
    for step in range(100):
        osc.step([hx, 0.0, TOP + 0.127], 0.04)                         # grip site at the cube's centre, fingers open
    for step in range(60):
        osc.step([hx, 0.0, TOP + 0.127], 0.0)                          # close
    gym.refresh_dof_state_tensor(sim)
    assert float(osc.dof[:, 7:, 0].min()) > 0.020 and float(osc.dof[:, 7:, 0].max()) < 0.030      # the fingers stopped on the 5 cm cube
    for step in range(100):
        osc.step([hx, 0.0, TOP + 0.127 + 0.15 * min(step, 80) / 80.0], 0.0)     # lift 15 cm
    gym.refresh_actor_root_state_tensor(sim); gym.refresh_rigid_body_state_tensor(sim)
    A = root[:, 3].cpu().numpy()
    assert (A[:, 2] > TOP + SIZE_A / 2 + 0.13).all(), A[:, 2]                              # the cube came along
    assert np.abs(A[:, 2] - osc.rb[:, osc.site, 2].cpu().numpy()).max() < 0.01            # ... between the finger tips
    # gym's net contact force tensor on the held cube's two fingers (round 6: scenes fill the actor's rows): friction carries the cube's weight,
    # the two fingers squeeze with equal and opposite forces of at most the finger drives' kd vmax = 20 N
    netf = gym.acquire_net_contact_force_tensor(sim).view(n, -1, 3)
    gym.refresh_net_contact_force_tensor(sim)
    names = franka.body_names
    fl, fr_ = netf[:, names.index("panda_leftfinger")].cpu().numpy(), netf[:, names.index("panda_rightfinger")].cpu().numpy()
    w_cube = float(sim.engine._tp.scene.free_mass[0]) * 9.81
    assert np.abs(fl[:, 2] + fr_[:, 2] + w_cube).max() < 0.1 * w_cube, (fl, fr_, w_cube)     # the cube hangs on the fingers: they feel its weight, downwards
    sq = np.linalg.norm(fl[:, :2], axis=1)
    assert (sq > 2.0).all() and (sq < 21.0).all() and np.abs(fl[:, :2] + fr_[:, :2]).max() < 0.1 * sq.max(), (fl, fr_)
    def over_b(z, fingers, steps, ramp):
        """servo the hand so that CUBE A (its position is an observation of the task) comes over cube B: the cube does not sit exactly at the grip site"""
        gym.refresh_rigid_body_state_tensor(sim)
        start = osc.rb[:, osc.hb, 0:3].clone()
        for step in range(steps):
            gym.refresh_actor_root_state_tensor(sim)
            want = osc.rb[:, osc.hb, 0:2] + (root[:, 4, 0:2] - root[:, 3, 0:2])
            f = min(step, ramp) / float(ramp)
            goal = torch.cat([start[:, 0:2] + f * (want - start[:, 0:2]), torch.full((n, 1), z(step), device=sim.device)], dim=1)
            osc.step(goal, fingers)
    over_b(lambda k: TOP + 0.277, 0.0, 150, 100)                                           # carry it over cube B
    over_b(lambda k: TOP + 0.277 - 0.078 * min(k, 60) / 60.0, 0.0, 80, 1)                  # down until it stands on B
    gym.refresh_rigid_body_state_tensor(sim)
    hold = osc.rb[:, osc.hb, 0:3].clone()
    for step in range(80):
        osc.step(hold, 0.04)                                                               # let go
    for step in range(60):
        osc.step(torch.cat([hold[:, 0:2], torch.full((n, 1), TOP + 0.30, device=sim.device)], dim=1), 0.04)      # and retreat
    gym.refresh_actor_root_state_tensor(sim)
    A, Bc = root[:, 3].cpu().numpy(), root[:, 4].cpu().numpy()
    assert np.abs(A[:, 2] - (TOP + SIZE_B + SIZE_A / 2)).max() < 4e-3, A[:, 2]             # stacked
    assert np.abs(A[:, 0:2] - Bc[:, 0:2]).max() < 0.02 and np.abs(A[:, 7:]).max() < 0.05   # ... over B, at rest


def test_scene_arm_grasps_lifts_and_stacks_a_cube_cpu():
    _grasp("cpu")


@pytest.mark.gpu
def test_scene_arm_grasps_lifts_and_stacks_a_cube_hip():
    _grasp("cuda:0")


# ------------------------------------------------------------------------------------------------ the reference's own task file, unmodified
@pytest.fixture()
def franka_task():
    """isaacgymenvs/tasks/franka_cube_stack.py imported as it is, with the stand-ins registered as `isaacgym` / `gym` (the way
    tests/test_gymapi_shim.py imports the other task files)"""
    import importlib
    import sys
    import types
    import isaacgymenvs_amd.shims as shims
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")}
    for k in saved:
        del sys.modules[k]
    shims.install(force=True)
    for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"),
                      ("isaacgymenvs.utils", "isaacgymenvs/utils"), ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = mod
    mod = importlib.import_module("isaacgymenvs.tasks.franka_cube_stack")
    vt = importlib.import_module("isaacgymenvs.tasks.base.vec_task")
    mod.load = lambda name: importlib.import_module("isaacgymenvs.tasks." + name)       # (the other task files of the same tree: trifinger.py)
    yield mod, vt
    for k in [k for k in sys.modules if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _reference_task(franka_task, device):
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    from oracle import jit_twins as J
    if device == "cpu":
        native.build_cpu()
    mod, vt = franka_task
    assert os.path.samefile(mod.__file__, os.path.join(REF, "isaacgymenvs", "tasks", "franka_cube_stack.py"))
    vt.EXISTING_SIM = None
    n = 16
    cfg = omegaconf_to_dict(compose("config", overrides=["task=FrankaCubeStack"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
    cfg["env"]["numEnvs"] = n
    cfg["sim"]["use_gpu_pipeline"] = device != "cpu"
    torch.manual_seed(3)
    env = mod.FrankaCubeStack(cfg, rl_device=device, sim_device=device, graphics_device_id=-1, headless=True, virtual_screen_capture=False,
                              force_render=False)
    sim = env.sim
    assert sim.scene == {1: ("static", 0), 2: ("static", 1), 3: ("free", 0), 4: ("free", 1)}
    assert env.num_dofs == 9 and tuple(env._root_state.shape) == (n, 5, 13) and env.obs_buf.shape[1] == 19 and env.num_actions == 7
    g = torch.Generator().manual_seed(0)
    table = float(env.reward_settings["table_height"])
    for step in range(60):
        act = (torch.rand((n, 7), generator=g) * 2 - 1).to(device)
        obs, rew, reset, _ = env.step(act)
        assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all()
        # the task's own jitted reward (it just ran on the engine's state) against the restated twin on the same states
        st = {k: v.detach().cpu().numpy() for k, v in env.states.items()}
        if step % 10 == 0:
            prog = env.progress_buf.cpu().numpy()
            r2, _ = J.compute_franka_cube_stack_reward(np.zeros(n, np.int64), prog, act.cpu().numpy(), st, env.reward_settings, env.max_episode_length)
            env._refresh()
            st2 = {k: v.detach().cpu().numpy() for k, v in env.states.items()}
            r3, _ = J.compute_franka_cube_stack_reward(np.zeros(n, np.int64), prog, act.cpu().numpy(), st2, env.reward_settings, env.max_episode_length)
            np.testing.assert_allclose(r3, r2, rtol=1e-6)        # _refresh is idempotent between steps
            np.testing.assert_allclose(env.rew_buf.detach().cpu().numpy(), r2, rtol=1e-5, atol=1e-6)
    # the cubes the task placed at reset (:522-594: on the table, B first, A clear of B) lie on the table: physics holds them there
    A, B = env._cubeA_state.cpu().numpy(), env._cubeB_state.cpu().numpy()
    on_a = np.abs(A[:, 2] - (table + env.cubeA_size / 2)) < 3e-3
    on_b = np.abs(B[:, 2] - (table + env.cubeB_size / 2)) < 3e-3
    assert on_b.mean() >= 0.8 and on_a.mean() >= 0.7, (A[:, 2], B[:, 2])           # (a flailing arm may knock one about)
    assert np.isfinite(A).all() and np.isfinite(B).all()
    assert float(env._eef_state[:, 2].min()) > table - 0.01                        # the gripper does not dive through the table
    # the arm answers the OSC command: ask every env for +z and the end effector rises
    z0 = env._eef_state[:, 2].clone()
    up = torch.zeros((n, 7), device=device); up[:, 2] = 1.0
    for _ in range(15):
        env.step(up)
    assert float((env._eef_state[:, 2] - z0).mean()) > 0.03
    # episodes end at the horizon and reset_idx puts the cubes back on the table
    env.progress_buf[:] = env.max_episode_length - 2
    env.step(torch.zeros((n, 7), device=device)); env.step(torch.zeros((n, 7), device=device))
    assert int(env.progress_buf.max()) <= 2
    A = env._cubeA_state.cpu().numpy()
    assert np.abs(A[:, 2] - (table + env.cubeA_size / 2)).max() < 5e-3 and np.abs(A[:, 7:]).max() < 0.5


def test_reference_franka_cube_stack_steps_unmodified_cpu(franka_task):
    _reference_task(franka_task, "cpu")


@pytest.mark.gpu
def test_reference_franka_cube_stack_steps_unmodified_hip(franka_task):
    _reference_task(franka_task, "cuda:0")


TRIFINGER = os.path.join(REF, "assets", "trifinger", "robot_properties_fingers", "urdf", "pro", "trifingerpro.urdf")


def _reference_trifinger(franka_task, device):
    """The reference's unmodified trifinger.py (round 6): three fingers on a fixed base (gym.load_asset of trifingerpro.urdf with its package://
    collision meshes -> the Articulation robot), and its stage and object loaded from ONE-BODY URDF files (trifinger.py:1169-1255) -- the table plate
    (a mesh, simulated as its bounding box: exact for a plate) is the scene's static box, the cube its free box; the goal marker (another collision
    group, :561-563) and the arena's ring wall (a concave mesh: created without a collision shape, with a warning -- the stated gap) live in the
    stand-in.  Random actions through the task's own code (torque control with its safety checks, asymmetric observations with fingertip wrenches
    and joint torques, domain randomisation through apply_randomizations); the task's jitted reward against the restated twin on the same buffers."""
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    from oracle import jit_twins as J
    if device == "cpu":
        native.build_cpu()
    mod0, vt = franka_task
    mod = mod0.load("trifinger")
    assert os.path.samefile(mod.__file__, os.path.join(REF, "isaacgymenvs", "tasks", "trifinger.py"))
    vt.EXISTING_SIM = None
    n = 16
    cfg = omegaconf_to_dict(compose("config", overrides=["task=Trifinger"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
    cfg["env"]["numEnvs"] = n
    cfg["sim"]["use_gpu_pipeline"] = device != "cpu"
    torch.manual_seed(3)
    with pytest.warns(UserWarning, match="WITHOUT a collision shape"):
        env = mod.Trifinger(cfg, rl_device=device, sim_device=device, graphics_device_id=-1, headless=True, virtual_screen_capture=False, force_render=False)
    sim = env.sim
    sc = sim.engine._tp.scene
    assert sim.scene == {1: ("static", 0), 3: ("free", 0)} and sim.nactors == 5                  # robot, table, (boundary), cube, (goal)
    assert np.allclose(list(sc.static_half[0]), [0.355, 0.38, 0.005], atol=1e-4) and np.allclose(list(sc.free_half[0]), [0.0325] * 3, atol=1e-6)
    assert env.num_actions == 9 and env.obs_buf.shape == (n, 41)
    g = torch.Generator().manual_seed(0)
    for step in range(80):
        act = (torch.rand((n, 9), generator=g) * 2 - 1).to(device)
        obs, rew, reset, _ = env.step(act)
        assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all(), step
        if step % 10 == 0:
            c = lambda t: t.detach().cpu().numpy()  # noqa: E731
            rt = cfg["env"]["reward_terms"]
            r2, _, _ = J.compute_trifinger_reward(c(env.obs_buf), np.zeros(n, np.int64), c(env.progress_buf), env.max_episode_length, cfg["sim"]["dt"],
                                                  rt["finger_move_penalty"]["weight"], rt["finger_reach_object_rate"]["weight"], rt["object_dist"]["weight"],
                                                  rt["object_rot"]["weight"], env.env_steps_count, c(env._object_goal_poses_buf), c(env._object_state_history[0]),
                                                  c(env._object_state_history[1]), c(env._fingertips_frames_state_history[0]),
                                                  c(env._fingertips_frames_state_history[1]), rt["keypoints_dist"]["activate"])
            np.testing.assert_allclose(c(env.rew_buf), r2, rtol=2e-4, atol=2e-4)
    # physics: the cube lies on the table plate (z = half its edge; a finger may have knocked it about, nothing throws it through the plate)
    cube = sim.engine.tensors["scene_state"][:, 0].cpu().numpy()
    assert np.isfinite(cube).all() and (cube[:, 2] > 0.0325 - 3e-3).all() and (np.abs(cube[:, 2] - 0.0325) < 3e-3).mean() >= 0.7, cube[:, :3]
    # the fingers answer their torques: the joint positions moved away from the start, inside their limits (the task's own safety clamps)
    q = env._dof_position.cpu().numpy()
    assert np.isfinite(q).all() and np.abs(q).max() < 3.0 and q.std(axis=0).max() > 0.01
    # an episode ends at its horizon and reset_idx puts the cube back on the table
    env.progress_buf[:] = env.max_episode_length - 2
    for _ in range(3):
        env.step(torch.zeros((n, 9), device=device))
    assert int(env.progress_buf.max()) <= 3
    cube = sim.engine.tensors["scene_state"][:, 0].cpu().numpy()
    assert np.abs(cube[:, 2] - 0.0325).max() < 0.02 and np.hypot(cube[:, 0], cube[:, 1]).max() < 0.25


@pytest.mark.skipif(not os.path.isfile(TRIFINGER), reason="the reference's trifinger assets are not reachable")
def test_reference_trifinger_steps_unmodified_cpu(franka_task):
    _reference_trifinger(franka_task, "cpu")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isfile(TRIFINGER), reason="the reference's trifinger assets are not reachable")
def test_reference_trifinger_steps_unmodified_hip(franka_task):
    _reference_trifinger(franka_task, "cuda:0")

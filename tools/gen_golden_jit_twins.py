#!/usr/bin/env python
"""Golden vectors for the remaining @torch.jit.script task functions, produced by running the REFERENCE's
own functions (same recipe as tools/gen_golden.py, whose import stubs are reused).  Development container only
(needs /root/reference); writes tests/golden/jit_twins_*.npz, which are committed.

Functions captured:
  isaacgymenvs/tasks/ball_balance.py:459      compute_bbot_reward
  isaacgymenvs/tasks/ingenuity.py:410         compute_ingenuity_reward
  isaacgymenvs/tasks/franka_cabinet.py:488    compute_franka_reward, :556 compute_grasp_transforms
  isaacgymenvs/tasks/franka_cube_stack.py:40  axisangle2quat, :697 compute_franka_reward
  isaacgymenvs/tasks/allegro_hand.py:663      compute_hand_reward, :728 randomize_rotation_pen
  isaacgymenvs/tasks/trifinger.py:1260        lgsk_kernel, :1277 gen_keypoints, :1292 compute_trifinger_reward,
                                              :1386 compute_trifinger_observations_states
  isaacgymenvs/tasks/amp/humanoid_amp_base.py:462 dof_to_obs, :494 compute_humanoid_observations, :536 compute_humanoid_reset
  isaacgymenvs/tasks/humanoid_amp.py:299      build_amp_observations
  isaacgymenvs/tasks/dextreme/allegro_hand_dextreme.py:1598 compute_hand_reward
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

OUT = G.OUT


def save(name, **kw):
    out = {}
    for k, v in kw.items():
        out[k] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "jit_twins_" + name + ".npz"), **out)
    print(name, {k: tuple(np.shape(v)) for k, v in out.items() if np.ndim(v) > 0 and k in ("rew", "obs", "out", "quat")})


def unit(g, n, bias=None):
    q = torch.randn(n, 4, generator=g)
    if bias is not None:
        q = q * 0.3 + torch.tensor(bias)
    return q / q.norm(dim=-1, keepdim=True)


def main():
    torch.set_num_threads(1)
    G.import_reference()
    tk = types.ModuleType("tkinter")   # dextreme does `from tkinter import W` at module scope; tkinter is not installed here
    tk.__getattr__ = lambda name: None
    sys.modules.setdefault("tkinter", tk)
    oc = types.ModuleType("omegaconf")   # adr_vec_task.py imports omegaconf (absent here) for type names only
    oc.__getattr__ = lambda name: type(name, (), {})
    sys.modules.setdefault("omegaconf", oc)
    imp = lambda n: importlib.import_module("isaacgymenvs.tasks." + n)  # noqa: E731
    n = 256

    # ---- BallBalance
    g = torch.Generator().manual_seed(11)
    m = imp("ball_balance")
    ball_pos = torch.randn(n, 3, generator=g) * torch.tensor([0.3, 0.3, 0.3]) + torch.tensor([0.0, 0.0, 0.6])
    ball_pos[:6, 2] = torch.tensor([0.149, 0.15, 0.151, 0.1, 0.7, 0.2])
    ball_vel = torch.randn(n, 3, generator=g)
    tray = torch.randn(n, 3, generator=g)
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, 502, (n,), generator=g); progress[:4] = torch.tensor([497, 498, 499, 500])
    rew, reset = m.compute_bbot_reward(tray, ball_pos, ball_vel, 0.1, reset_in, progress, 500.0)
    save("bbot", tray_positions=tray, ball_positions=ball_pos, ball_velocities=ball_vel, reset_in=reset_in, progress=progress, rew=rew, reset=reset,
         scalar_ball_radius=0.1, scalar_max_episode_length=500.0)

    # ---- Ingenuity
    g = torch.Generator().manual_seed(12)
    m = imp("ingenuity")
    pos = torch.randn(n, 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 2.0])
    pos[:4, 2] = torch.tensor([0.49, 0.5, 0.51, 3.0])
    target = torch.randn(n, 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 2.0])
    target[4] = pos[4] + torch.tensor([8.5, 0.0, 0.0]); target[5] = pos[5] + torch.tensor([7.5, 0.0, 0.0])
    quat = unit(g, n, [0.0, 0.0, 0.0, 1.0])
    linvel, angvel = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g) * 2
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, 2002, (n,), generator=g); progress[:4] = torch.tensor([1997, 1998, 1999, 2000])
    rew, reset = m.compute_ingenuity_reward(pos, target, quat, linvel, angvel, reset_in, progress, 2000.0)
    save("ingenuity", root_positions=pos, target_root_positions=target, root_quats=quat, root_linvels=linvel, root_angvels=angvel,
         reset_in=reset_in, progress=progress, rew=rew, reset=reset, scalar_max_episode_length=2000.0)

    # ---- FrankaCabinet
    g = torch.Generator().manual_seed(13)
    m = imp("franka_cabinet")
    sc = dict(dist_reward_scale=2.0, rot_reward_scale=0.5, around_handle_reward_scale=0.25, open_reward_scale=7.5,
              finger_dist_reward_scale=5.0, action_penalty_scale=0.01, distX_offset=0.04, max_episode_length=500.0)
    dg = torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0.5, 0.0, 0.6])
    fg = dg + torch.randn(n, 3, generator=g) * 0.05
    fg[:16] = dg[:16] + torch.randn(16, 3, generator=g) * 0.008          # inside the d <= 0.02 bonus
    lf = dg + torch.randn(n, 3, generator=g) * 0.05 + torch.tensor([0.02, 0.0, 0.02])
    rf = dg + torch.randn(n, 3, generator=g) * 0.05 + torch.tensor([0.02, 0.0, -0.02])
    fgr, dgr = unit(g, n), unit(g, n)
    actions = torch.rand(n, 9, generator=g) * 2 - 1
    cab = torch.rand(n, 4, generator=g) * 0.45
    cab[:6, 3] = torch.tensor([0.009, 0.011, 0.19, 0.21, 0.389, 0.391])
    axes = [torch.tensor(a).repeat(n, 1) for a in ([0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0])]
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, 502, (n,), generator=g); progress[:4] = torch.tensor([497, 498, 499, 500])
    rew, reset = m.compute_franka_reward(reset_in, progress, actions, cab, fg, dg, fgr, dgr, lf, rf, axes[0], axes[1], axes[2], axes[3], n,
                                         sc["dist_reward_scale"], sc["rot_reward_scale"], sc["around_handle_reward_scale"], sc["open_reward_scale"],
                                         sc["finger_dist_reward_scale"], sc["action_penalty_scale"], sc["distX_offset"], sc["max_episode_length"])
    save("franka_cabinet", reset_in=reset_in, progress=progress, actions=actions, cabinet_dof_pos=cab, franka_grasp_pos=fg, drawer_grasp_pos=dg,
         franka_grasp_rot=fgr, drawer_grasp_rot=dgr, franka_lfinger_pos=lf, franka_rfinger_pos=rf, gripper_forward_axis=axes[0],
         drawer_inward_axis=axes[1], gripper_up_axis=axes[2], drawer_up_axis=axes[3], rew=rew, reset=reset,
         **{"scalar_" + k: v for k, v in sc.items()})
    hand_rot, drawer_rot, flr, dlr = unit(g, n), unit(g, n), unit(g, n), unit(g, n)
    hand_pos, drawer_pos, flp, dlp = (torch.randn(n, 3, generator=g) for _ in range(4))
    gfr, gfp, gdr, gdp = m.compute_grasp_transforms(hand_rot, hand_pos, flr, flp, drawer_rot, drawer_pos, dlr, dlp)
    save("grasp_transforms", hand_rot=hand_rot, hand_pos=hand_pos, franka_local_grasp_rot=flr, franka_local_grasp_pos=flp, drawer_rot=drawer_rot,
         drawer_pos=drawer_pos, drawer_local_grasp_rot=dlr, drawer_local_grasp_pos=dlp, global_franka_rot=gfr, global_franka_pos=gfp,
         global_drawer_rot=gdr, global_drawer_pos=gdp)

    # ---- FrankaCubeStack
    g = torch.Generator().manual_seed(14)
    m = imp("franka_cube_stack")
    vec = torch.randn(n, 3, generator=g) * 1.5
    vec[:3] = torch.tensor([[0.0, 0.0, 0.0], [5e-7, 0.0, 0.0], [2e-6, 0.0, 0.0]])
    quat = m.axisangle2quat(vec)
    save("axisangle2quat", vec=vec, quat=quat, scalar_eps=1e-6)
    rs = dict(r_dist_scale=0.1, r_lift_scale=1.5, r_align_scale=2.0, r_stack_scale=16.0, table_height=1.025)
    cubeA_size, cubeB_size = torch.full((n,), 0.050), torch.full((n,), 0.070)
    cubeA_pos = torch.randn(n, 3, generator=g) * torch.tensor([0.1, 0.1, 0.08]) + torch.tensor([0.0, 0.0, 1.025 + 0.06])
    cubeB_pos = torch.randn(n, 3, generator=g) * torch.tensor([0.1, 0.1, 0.0]) + torch.tensor([0.0, 0.0, 1.025 + 0.035])
    # a block of stacked cases: A on B, aligned, gripper away / close
    k = 24
    cubeA_pos[:k] = cubeB_pos[:k] + torch.tensor([0.0, 0.0, 0.0]) + torch.randn(k, 3, generator=g) * torch.tensor([0.008, 0.008, 0.0])
    cubeA_pos[:k, 2] = 1.025 + 0.070 + 0.025 + torch.randn(k, generator=g) * 0.01
    eef = cubeA_pos + torch.randn(n, 3, generator=g) * 0.06
    eef[:k // 2] = cubeA_pos[:k // 2] + torch.tensor([0.0, 0.0, 0.1])
    lfp, rfp = eef + torch.tensor([0.0, 0.03, 0.0]), eef + torch.tensor([0.0, -0.03, 0.0])
    states = {"cubeA_size": cubeA_size, "cubeB_size": cubeB_size, "cubeA_pos": cubeA_pos, "cubeA_pos_relative": cubeA_pos - eef,
              "eef_lf_pos": lfp, "eef_rf_pos": rfp, "cubeA_to_cubeB_pos": cubeB_pos - cubeA_pos}
    actions = torch.rand(n, 7, generator=g) * 2 - 1
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, 302, (n,), generator=g); progress[-4:] = torch.tensor([297, 298, 299, 300])
    rew, reset = m.compute_franka_reward(reset_in, progress, actions, states, rs, 300.0)
    save("franka_cube_stack", reset_in=reset_in, progress=progress, actions=actions, rew=rew, reset=reset, scalar_max_episode_length=300.0,
         **states, **{"scalar_" + k_: v for k_, v in rs.items()})

    # ---- AllegroHand
    g = torch.Generator().manual_seed(15)
    m = imp("allegro_hand")
    hp = dict(max_episode_length=600.0, dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1, action_penalty_scale=-0.0002,
              success_tolerance=0.1, reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=0.0, max_consecutive_successes=0, av_factor=0.1)
    for tag, mcs, ignore_z, fall_pen in (("a", 0, False, 0.0), ("b", 50, True, -50.0)):
        object_pos = torch.randn(n, 3, generator=g) * 0.12
        target_pos = torch.zeros(n, 3)
        target_rot = unit(g, n)
        object_rot = unit(g, n)
        object_rot[: n // 4] = torch.nn.functional.normalize(target_rot[: n // 4] + 0.04 * torch.randn(n // 4, 4, generator=g), dim=-1)
        actions = torch.rand(n, 16, generator=g) * 2 - 1
        reset_in = (torch.rand(n, generator=g) < 0.05).long()
        reset_goal_in = (torch.rand(n, generator=g) < 0.05).long()
        progress = torch.randint(0, 602, (n,), generator=g)
        successes = torch.randint(0, 52, (n,), generator=g).float()
        cons = torch.tensor([3.25])
        out = m.compute_hand_reward(torch.zeros(n), reset_in, reset_goal_in, progress.clone(), successes.clone(), cons.clone(), hp["max_episode_length"],
                                    object_pos, object_rot, target_pos, target_rot, hp["dist_reward_scale"], hp["rot_reward_scale"], hp["rot_eps"],
                                    actions, hp["action_penalty_scale"], hp["success_tolerance"], hp["reach_goal_bonus"], hp["fall_dist"],
                                    fall_pen, mcs, hp["av_factor"], ignore_z)
        save("allegro_hand_reward_" + tag, reset_in=reset_in, reset_goal_in=reset_goal_in, progress_in=progress, successes_in=successes,
             consecutive_successes_in=cons, object_pos=object_pos, object_rot=object_rot, target_pos=target_pos, target_rot=target_rot,
             actions=actions, rew=out[0], resets=out[1], goal_resets=out[2], progress=out[3], successes=out[4], cons_successes=out[5],
             scalar_max_consecutive_successes=mcs, scalar_ignore_z_rot=int(ignore_z), scalar_fall_penalty=fall_pen,
             **{"scalar_" + k_: v for k_, v in hp.items() if k_ not in ("max_consecutive_successes", "fall_penalty")})
    rand0, rand1 = torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 2 - 1
    ux, uy, uz = (torch.tensor(a).repeat(n, 1) for a in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]))
    rot = m.randomize_rotation_pen(rand0, rand1, torch.tensor(0.3), ux, uy, uz)   # a 0-dim tensor in the reference (allegro_hand.py:578)
    save("rotation_pen", rand0=rand0, rand1=rand1, x_unit=ux, y_unit=uy, z_unit=uz, out=rot, scalar_max_angle=0.3)

    # ---- Trifinger
    g = torch.Generator().manual_seed(16)
    m = imp("trifinger")
    x = torch.randn(n, generator=g) * 0.1
    save("lgsk", x=x, out_50=m.lgsk_kernel(x, 50.0, 2.0), out_30=m.lgsk_kernel(x, 30.0, 2.0))
    pose = torch.cat([torch.randn(n, 3, generator=g) * 0.1, unit(g, n)], dim=-1)
    save("keypoints", pose=pose, out=m.gen_keypoints(pose), size=np.array([0.065, 0.065, 0.065], np.float32))
    goal = torch.cat([torch.randn(n, 3, generator=g) * 0.1, unit(g, n)], dim=-1)
    obj = torch.cat([pose + torch.cat([torch.zeros(n, 3), torch.zeros(n, 4)], -1), torch.randn(n, 6, generator=g)], dim=-1)
    last_obj = obj + torch.cat([torch.randn(n, 3, generator=g) * 0.002, torch.zeros(n, 10)], -1)
    ft = torch.randn(n, 3, 13, generator=g) * 0.1
    last_ft = ft + torch.randn(n, 3, 13, generator=g) * 0.003
    progress = torch.randint(0, 752, (n,), generator=g); progress[:4] = torch.tensor([747, 748, 749, 750])
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    tp = dict(episode_length=750, dt=0.02, finger_move_penalty_weight=-0.5, finger_reach_object_weight=-250.0, object_dist_weight=2000.0,
              object_rot_weight=300.0)
    extra = {}
    for tag, use_kp, steps in (("kp", True, 1000), ("pose", False, 1000), ("late", True, 60000000)):
        rew, reset, info = m.compute_trifinger_reward(torch.zeros(n, 41), reset_in, progress, tp["episode_length"], tp["dt"],
                                                      tp["finger_move_penalty_weight"], tp["finger_reach_object_weight"], tp["object_dist_weight"],
                                                      tp["object_rot_weight"], steps, goal, obj, last_obj, ft, last_ft, use_kp)
        extra.update({"rew_" + tag: rew, "reset_" + tag: reset, "info_move_" + tag: info["finger_movement_penalty"],
                      "info_reach_" + tag: info["finger_reach_object_reward"], "scalar_steps_" + tag: steps})
    save("trifinger_reward", reset_in=reset_in, progress=progress, object_goal_poses=goal, object_state=obj, last_object_state=last_obj,
         fingertip_state=ft, last_fingertip_state=last_ft, **extra, **{"scalar_" + k_: v for k_, v in tp.items()})
    dof_pos, dof_vel, actions, tau = (torch.randn(n, 9, generator=g) for _ in range(4))
    wrench = torch.randn(n, 18, generator=g)
    obs_s, st_s = m.compute_trifinger_observations_states(False, dof_pos, dof_vel, obj, goal, actions, ft, tau, wrench)
    obs_a, st_a = m.compute_trifinger_observations_states(True, dof_pos, dof_vel, obj, goal, actions, ft, tau, wrench)
    save("trifinger_obs", dof_position=dof_pos, dof_velocity=dof_vel, object_state=obj, object_goal_poses=goal, actions=actions, fingertip_state=ft,
         joint_torques=tau, tip_wrenches=wrench, obs_sym=obs_s, states_sym=st_s, obs_asym=obs_a, states_asym=st_a)

    # the cuboid-pose samplers draw inside the jitted function: replay the global generator to capture the draws they consume
    def replay(seed, *shapes_kinds):
        torch.manual_seed(seed)
        return [torch.rand(s) if k == "u" else torch.randn(s) for s, k in shapes_kinds]
    u0, u1 = replay(101, ((n,), "u"), ((n,), "u"))
    torch.manual_seed(101); x, y = m.random_xy(n, 0.08, "cpu")
    (uz,) = replay(102, ((n,), "u"))
    torch.manual_seed(102); z = m.random_z(n, 0.0325, 0.1, "cpu")
    (g4,) = replay(103, ((n, 4), "n"))
    torch.manual_seed(103); qo = m.random_orientation(n, "cpu")
    (u3,) = replay(104, ((n, 3), "u"))
    base = unit(g, n)
    torch.manual_seed(104); qw = m.random_orientation_within_angle(n, "cpu", base, 0.6)
    ax, mg = replay(105, ((n, 3), "n"), ((n, 1), "n"))
    torch.manual_seed(105); av = m.random_angular_vel(n, "cpu", 0.5)
    (uy,) = replay(106, ((n,), "u"))
    torch.manual_seed(106); qy = m.random_yaw_orientation(n, "cpu")
    save("trifinger_samplers", rand_xy=torch.stack([u0, u1], -1), xy=torch.stack([x, y], -1), rand_z=uz, z=z, randn_orientation=g4, orientation=qo,
         rand_within=u3, base=base, within=qw, randn_angvel=torch.cat([ax, mg], -1), angvel=av, rand_yaw=uy, yaw=qy,
         default=m.default_orientation(n, "cpu"), scalar_max_dist=0.08, scalar_min_height=0.0325, scalar_max_height=0.1, scalar_max_angle=0.6,
         scalar_magnitude_stdev=0.5)

    # ---- HumanoidAMP
    g = torch.Generator().manual_seed(17)
    m = imp("amp.humanoid_amp_base")
    pose28 = torch.randn(n, 28, generator=g) * 0.8
    pose28[0, 0:3] = 0.0                      # zero exponential map -> default axis branch
    pose28[1, 0:3] = torch.tensor([2e-6, 0.0, 0.0])
    pose28[2, 3:6] = torch.tensor([3.5, 0.0, 0.0])   # |angle| > pi: normalize_angle wraps it negative -> default axis branch
    save("amp_dof_to_obs", pose=pose28, out=m.dof_to_obs(pose28))
    root = torch.zeros(n, 13)
    root[:, 0:3] = torch.randn(n, 3, generator=g) + torch.tensor([0.0, 0.0, 0.9])
    root[:, 3:7] = unit(g, n, [0.0, 0.0, 0.0, 1.0])
    root[:, 7:13] = torch.randn(n, 6, generator=g)
    dof_vel = torch.randn(n, 28, generator=g) * 3
    key = root[:, None, 0:3] + torch.randn(n, 4, 3, generator=g) * 0.5
    obs_l = m.compute_humanoid_observations(root, pose28, dof_vel, key, True)
    obs_g = m.compute_humanoid_observations(root, pose28, dof_vel, key, False)
    m2 = importlib.import_module("isaacgymenvs.tasks.humanoid_amp")
    amp_obs = m2.build_amp_observations(root, pose28, dof_vel, key, True)
    nb = 15
    contact = torch.relu(torch.randn(n, nb, 3, generator=g) - 1.0) * 2.0
    body_pos = torch.rand(n, nb, 3, generator=g) * 1.2
    ids = torch.tensor([5, 8, 11, 14])
    progress = torch.randint(0, 302, (n,), generator=g); progress[:6] = torch.tensor([0, 1, 2, 298, 299, 300])
    reset_in = torch.zeros(n, dtype=torch.long)
    r_e, t_e = m.compute_humanoid_reset(reset_in, progress, contact, ids, body_pos, 300.0, True, 0.25)
    r_n, t_n = m.compute_humanoid_reset(reset_in, progress, contact, ids, body_pos, 300.0, False, 0.25)
    save("amp_obs_reset", root_states=root, dof_pos=pose28, dof_vel=dof_vel, key_body_pos=key, obs_local=obs_l, obs_global=obs_g, amp_obs=amp_obs,
         contact_buf=contact, rigid_body_pos=body_pos, contact_body_ids=ids, progress=progress, reset_early=r_e, terminated_early=t_e,
         reset_noearly=r_n, terminated_noearly=t_n, scalar_max_episode_length=300.0, scalar_termination_height=0.25)

    # ---- DeXtreme
    g = torch.Generator().manual_seed(18)
    try:
        m = imp("dextreme.allegro_hand_dextreme")
    except Exception as e:  # noqa: BLE001
        print("dextreme import failed:", type(e).__name__, e)
        return
    dp = dict(max_episode_length=320.0, dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1, action_penalty_scale=-0.0001,
              action_delta_penalty_scale=-0.01, success_tolerance=0.4, reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=-50.0,
              max_consecutive_successes=50, av_factor=0.1, num_success_hold_steps=1)
    object_pos = torch.randn(n, 3, generator=g) * 0.12
    target_pos = torch.zeros(n, 3)
    target_rot, object_rot = unit(g, n), unit(g, n)
    object_rot[: n // 2] = torch.nn.functional.normalize(target_rot[: n // 2] + 0.15 * torch.randn(n // 2, 4, generator=g), dim=-1)
    actions = torch.rand(n, 16, generator=g) * 2 - 1
    cur_t, prev_t, dvel = (torch.randn(n, 16, generator=g) for _ in range(3))
    reset_in = (torch.rand(n, generator=g) < 0.05).long()
    reset_goal_in = torch.zeros(n, dtype=torch.long)
    hold = torch.randint(0, 3, (n,), generator=g)
    progress = torch.randint(0, 322, (n,), generator=g)
    successes = torch.randint(0, 52, (n,), generator=g).float()
    cons = torch.tensor([1.75])
    # reset_goal_buf is passed as bool: with this torch, torch.where (:1630) refuses the int64 condition the reference task would hand it
    # (reset_goal_buf = reset_buf.clone(), :1268); values are the same 0/1
    out = m.compute_hand_reward(torch.zeros(n), reset_in, reset_goal_in.bool(), progress.clone(), hold.clone(), cur_t, prev_t, dvel, successes.clone(),
                                cons.clone(), dp["max_episode_length"], object_pos, object_rot, target_pos, target_rot, dp["dist_reward_scale"],
                                dp["rot_reward_scale"], dp["rot_eps"], actions, dp["action_penalty_scale"], dp["action_delta_penalty_scale"],
                                dp["success_tolerance"], dp["reach_goal_bonus"], dp["fall_dist"], dp["fall_penalty"],
                                dp["max_consecutive_successes"], dp["av_factor"], dp["num_success_hold_steps"])
    names = ("rew", "resets", "goal_resets", "progress", "hold_count", "successes", "cons_successes", "dist_rew", "rot_rew", "action_penalty",
             "action_delta_penalty", "velocity_penalty", "reach_goal_rew", "fall_rew", "timeout_rew")
    save("dextreme_reward", reset_in=reset_in, reset_goal_in=reset_goal_in, progress_in=progress, hold_count_in=hold, cur_targets=cur_t,
         prev_targets=prev_t, hand_dof_vel=dvel, successes_in=successes, consecutive_successes_in=cons, object_pos=object_pos, object_rot=object_rot,
         target_pos=target_pos, target_rot=target_rot, actions=actions, **{k_: v.long() if v.dtype == torch.bool else v for k_, v in zip(names, out)},
         **{"scalar_" + k_: v for k_, v in dp.items()})


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Generate golden input/output vectors by running the REFERENCE's own @torch.jit.script functions.

Runs only in the development container (needs /root/reference); the .npz files it writes under tests/golden/
are committed so the GPU box (no /root/reference) can check the HIP kernels and the oracle against them.

Recipe: stub `isaacgym` and `gym` (absent here), pre-register the reference packages as
namespace modules so their __init__ files (which import hydra) are skipped, then import the task modules by name.
Functions captured:
  isaacgymenvs/tasks/ant.py:325        compute_ant_reward
  isaacgymenvs/tasks/ant.py:374        compute_ant_observations
  isaacgymenvs/tasks/humanoid.py:323   compute_humanoid_reward
  isaacgymenvs/tasks/humanoid.py:378   compute_humanoid_observations
  isaacgymenvs/tasks/cartpole.py:180   compute_cartpole_reward
  isaacgymenvs/tasks/anymal.py:311     compute_anymal_reward, :354 compute_anymal_observations
  isaacgymenvs/tasks/quadcopter.py:348 compute_quadcopter_reward
  isaacgymenvs/tasks/shadow_hand.py:746 compute_hand_reward, :803 randomize_rotation, :528 ShadowHand.compute_full_state (mock self)
  isaacgymenvs/tasks/anymal_terrain.py:294,302,315,515   AnymalTerrain.check_termination / compute_observations /
                                       compute_reward / get_heights (bound to a mock `self`), :676 quat_apply_yaw, :683 wrap_to_pi
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    tmp = tempfile.mkdtemp()
    stub = ("class _Dummy:\n    def __init__(self, *a, **k):\n        pass\n    def __call__(self, *a, **k):\n        return _Dummy()\n"
            "    def __getattr__(self, n):\n        return _Dummy()\n\ndef __getattr__(name):\n    return _Dummy()\n")
    os.makedirs(os.path.join(tmp, "isaacgym"))
    for m in ("gymapi", "gymutil"):
        open(os.path.join(tmp, "isaacgym", m + ".py"), "w").write(stub)
    # `from isaacgym.terrain_utils import *` (anymal_terrain.py:542): the module is absent from the reference tree
    open(os.path.join(tmp, "isaacgym", "terrain_utils.py"), "w").write("__all__ = []\n")
    open(os.path.join(tmp, "isaacgym", "__init__.py"), "w").write("from . import gymapi, gymutil, gymtorch, terrain_utils\n" + stub)
    open(os.path.join(tmp, "isaacgym", "gymtorch.py"), "w").write(
        "def wrap_tensor(x):\n    return x\n\ndef unwrap_tensor(x):\n    return x\n")
    os.makedirs(os.path.join(tmp, "gym"))
    open(os.path.join(tmp, "gym", "__init__.py"), "w").write("from . import spaces\n\nclass Space:\n    pass\n")
    open(os.path.join(tmp, "gym", "spaces.py"), "w").write("class Box:\n    def __init__(self, *a, **k):\n        pass\n")
    sys.path.insert(0, tmp)
    if not hasattr(np, "Inf"):
        np.Inf = np.inf  # vec_task.py:107 uses np.Inf (removed in NumPy 2)
    for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"),
                      ("isaacgymenvs.utils", "isaacgymenvs/utils"), ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = mod
    return {n: importlib.import_module("isaacgymenvs.tasks." + n) for n in ("ant", "humanoid", "cartpole", "anymal_terrain", "shadow_hand", "anymal", "quadcopter")}


def rand_quat(g, n):
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def locomotion_case(mod, name, nd, nsv, n, seed, hum, lo, up, gears, scal):
    g = torch.Generator().manual_seed(seed)
    root = torch.zeros(n, 13)
    root[:, 0:3] = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 3.0, 0.3]) + torch.tensor([0.0, 0.0, 0.6 if not hum else 1.2])
    root[:, 3:7] = rand_quat(g, n)
    # make half the batch near-upright so the thresholds (up_proj > 0.93, heading > 0.8) are exercised on both sides
    root[: n // 2, 3:7] = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 0.0, 1.0]) + 0.15 * torch.randn(n // 2, 4, generator=g), dim=-1)
    root[:, 7:13] = torch.randn(n, 6, generator=g) * 2.0
    lo_t, up_t = torch.tensor(lo, dtype=torch.float32), torch.tensor(up, dtype=torch.float32)
    u = torch.rand(n, nd, generator=g) * 1.1 - 0.05  # slightly beyond the limits too
    dof_pos = lo_t + u * (up_t - lo_t)
    dof_vel = torch.randn(n, nd, generator=g) * 5.0
    dof_force = torch.randn(n, nd, generator=g) * 50.0
    sensors = torch.randn(n, nsv, generator=g) * 30.0
    actions = torch.rand(n, nd, generator=g) * 2 - 1
    targets = torch.tensor([1000.0, 0.0, 0.0]).repeat(n, 1)
    potentials = -(targets[:, :2] - root[:, :2]).norm(dim=-1) / scal["dt"] + torch.randn(n, generator=g)
    inv_start_rot = torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(n, 1)
    b0 = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
    b1 = torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1)
    obs0 = torch.zeros(n, 12 + nd * (4 if hum else 3) + nsv)
    if hum:
        obs, pot, prev, upv, hv = mod.compute_humanoid_observations(
            obs0, root, targets, potentials.clone(), inv_start_rot, dof_pos, dof_vel, dof_force, lo_t, up_t,
            scal["dof_vel_scale"], sensors, actions, scal["dt"], scal["contact_force_scale"],
            scal["angular_velocity_scale"], b0, b1)
    else:
        obs, pot, prev, upv, hv = mod.compute_ant_observations(
            obs0, root, targets, potentials.clone(), inv_start_rot, dof_pos, dof_vel, lo_t, up_t,
            scal["dof_vel_scale"], sensors, actions, scal["dt"], scal["contact_force_scale"], b0, b1, 2)
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, int(scal["max_episode_length"]) + 2, (n,), generator=g)
    progress[:8] = torch.tensor([997, 998, 999, 1000, 0, 1, 998, 999])
    gears_t = torch.tensor(gears, dtype=torch.float32)
    if hum:
        rew, reset = mod.compute_humanoid_reward(
            obs, reset_in, progress, actions, scal["up_weight"], scal["heading_weight"], pot, prev,
            scal["actions_cost"], scal["energy_cost"], scal["joints_at_limit_cost"], float(max(gears)), gears_t,
            scal["termination_height"], scal["death_cost"], scal["max_episode_length"])
    else:
        rew, reset = mod.compute_ant_reward(
            obs, reset_in, progress, actions, scal["up_weight"], scal["heading_weight"], pot, prev,
            scal["actions_cost"], scal["energy_cost"], scal["joints_at_limit_cost"], scal["termination_height"],
            scal["death_cost"], scal["max_episode_length"])
    d = dict(root_states=root, targets=targets, potentials_in=potentials, inv_start_rot=inv_start_rot, dof_pos=dof_pos,
             dof_vel=dof_vel, dof_force=dof_force, dof_limits_lower=lo_t, dof_limits_upper=up_t, sensors=sensors,
             actions=actions, basis_vec0=b0, basis_vec1=b1, obs=obs, potentials=pot, prev_potentials=prev, up_vec=upv,
             heading_vec=hv, reset_in=reset_in, progress=progress, gears=gears_t, rew=rew, reset=reset)
    out = {k: v.numpy() for k, v in d.items()}
    out.update({"scalar_" + k: np.float64(v) for k, v in scal.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "obs", tuple(obs.shape), "rew mean", float(rew.mean()), "resets", int(reset.sum()))


def cartpole_case(mod, n, seed):
    g = torch.Generator().manual_seed(seed)
    pole_angle = torch.randn(n, generator=g) * 1.0
    pole_vel = torch.randn(n, generator=g) * 3
    cart_vel = torch.randn(n, generator=g) * 2
    cart_pos = torch.randn(n, generator=g) * 2
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, 502, (n,), generator=g)
    progress[:4] = torch.tensor([497, 498, 499, 500])
    rew, reset = mod.compute_cartpole_reward(pole_angle, pole_vel, cart_vel, cart_pos, 3.0, reset_in, progress, 500.0)
    np.savez_compressed(os.path.join(OUT, "cartpole_reward.npz"), pole_angle=pole_angle.numpy(), pole_vel=pole_vel.numpy(),
                        cart_vel=cart_vel.numpy(), cart_pos=cart_pos.numpy(), reset_in=reset_in.numpy(),
                        progress=progress.numpy(), rew=rew.numpy(), reset=reset.numpy(), scalar_reset_dist=3.0,
                        scalar_max_episode_length=500.0)
    print("cartpole_reward", n, "resets", int(reset.sum()))


def anymal_case(mod, n, seed):
    """Run the reference's own AnymalTerrain methods on a mock `self` holding random-but-plausible tensors."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from isaacgymenvs_amd.tasks.terrain import Terrain
    from isaacgymenvs_amd.utils.config import compose
    from isaacgymenvs_amd.registry import load_model
    cfg = compose(overrides=["task=AnymalTerrain"])["task"]
    learn = cfg["env"]["learn"]
    spec = load_model("anymal")
    terrain = Terrain(cfg["env"]["terrain"], num_robots=n, seed=7)
    g = torch.Generator().manual_seed(seed)
    A = mod.AnymalTerrain
    dt = 0.02
    m = types.SimpleNamespace()
    m.cfg = cfg
    m.device = "cpu"
    m.num_envs = n
    m.dt = dt
    m.terrain = terrain
    m.height_samples = torch.tensor(terrain.heightsamples).view(terrain.tot_rows, terrain.tot_cols)
    m.num_height_points = 140
    m.height_points = A.init_height_points(m)
    keys = {"termination": "terminalReward", "lin_vel_xy": "linearVelocityXYRewardScale", "lin_vel_z": "linearVelocityZRewardScale",
            "ang_vel_z": "angularVelocityZRewardScale", "ang_vel_xy": "angularVelocityXYRewardScale", "orient": "orientationRewardScale",
            "torque": "torqueRewardScale", "joint_acc": "jointAccRewardScale", "base_height": "baseHeightRewardScale",
            "air_time": "feetAirTimeRewardScale", "collision": "kneeCollisionRewardScale", "stumble": "feetStumbleRewardScale",
            "action_rate": "actionRateRewardScale", "hip": "hipRewardScale"}
    # non-zero values for the scales the shipped YAML sets to 0, so every term is exercised
    over = {"termination": -1.0, "orient": -1.0, "base_height": -5.0, "stumble": -2.0, "hip": -0.25}
    m.rew_scales = {k: float(over.get(k, learn[v])) * dt for k, v in keys.items()}
    m.lin_vel_scale, m.ang_vel_scale = learn["linearVelocityScale"], learn["angularVelocityScale"]
    m.dof_pos_scale, m.dof_vel_scale = learn["dofPositionScale"], learn["dofVelocityScale"]
    m.height_meas_scale = learn["heightMeasurementScale"]
    m.commands_scale = torch.tensor([m.lin_vel_scale, m.lin_vel_scale, m.ang_vel_scale])
    m.max_episode_length = 1000
    m.allow_knee_contacts = True
    m.base_index = 0
    m.knee_indices = torch.tensor([i for i, nm in enumerate(spec.body_names) if "THIGH" in nm])
    m.feet_indices = torch.tensor([i for i, nm in enumerate(spec.body_names) if "SHANK" in nm])
    root = torch.zeros(n, 13)
    root[:, 0] = torch.rand(n, generator=g) * 80.0
    root[:, 1] = torch.rand(n, generator=g) * 160.0
    root[:, 2] = 0.55 + 0.3 * torch.randn(n, generator=g)
    root[:, 3:7] = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 0.0, 1.0]) + 0.4 * torch.randn(n, 4, generator=g), dim=-1)
    root[:, 7:13] = torch.randn(n, 6, generator=g)
    m.root_states = root
    m.base_quat = root[:, 3:7]
    tj = importlib.import_module("isaacgymenvs.utils.torch_jit_utils")
    m.base_lin_vel = tj.quat_rotate_inverse(m.base_quat, root[:, 7:10])
    m.base_ang_vel = tj.quat_rotate_inverse(m.base_quat, root[:, 10:13])
    m.projected_gravity = tj.quat_rotate_inverse(m.base_quat, torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1))
    m.commands = torch.rand(n, 4, generator=g) * 2 - 1
    m.commands[: n // 8] *= 0.05
    heading = torch.rand(n, generator=g) * 20 - 10
    wrapped = mod.wrap_to_pi(heading.clone())
    m.dof_pos = torch.randn(n, 12, generator=g) * 0.5
    m.dof_vel = torch.randn(n, 12, generator=g) * 3
    m.last_dof_vel = m.dof_vel + torch.randn(n, 12, generator=g)
    m.default_dof_pos = torch.tensor([cfg["env"]["defaultJointAngles"][nm] for nm in spec.dof_names], dtype=torch.float32).repeat(n, 1)
    m.torques = torch.randn(n, 12, generator=g) * 30
    m.actions = torch.rand(n, 12, generator=g) * 2 - 1
    m.last_actions = torch.rand(n, 12, generator=g) * 2 - 1
    cf = torch.randn(n, 13, 3, generator=g) * 3.0
    cf[:, :, 2] = cf[:, :, 2].abs() * (torch.rand(n, 13, generator=g) < 0.5)
    cf[:, 0] *= (torch.rand(n, 1, generator=g) < 0.15)
    m.contact_forces = cf
    m.feet_air_time = torch.rand(n, 4, generator=g) * (torch.rand(n, 4, generator=g) < 0.7)
    m.progress_buf = torch.randint(0, 1002, (n,), generator=g)
    m.progress_buf[:4] = torch.tensor([997, 998, 999, 1000])
    m.timeout_buf = torch.rand(n, generator=g) < 0.05
    m.episode_sums = {k: torch.zeros(n) for k in ("lin_vel_xy", "lin_vel_z", "ang_vel_z", "ang_vel_xy", "orient", "torques",
                                                  "joint_acc", "base_height", "air_time", "collision", "stumble", "action_rate", "hip")}
    m.get_heights = lambda env_ids=None: A.get_heights(m, env_ids)
    inp = dict(root_states=root.clone(), commands=m.commands.clone(), dof_pos=m.dof_pos.clone(), dof_vel=m.dof_vel.clone(),
               last_dof_vel=m.last_dof_vel.clone(), torques=m.torques.clone(), actions=m.actions.clone(),
               last_actions=m.last_actions.clone(), contact_forces=cf.clone(), feet_air_time_in=m.feet_air_time.clone(),
               progress=m.progress_buf.clone(), timeout_in=m.timeout_buf.clone(), base_lin_vel=m.base_lin_vel.clone(),
               base_ang_vel=m.base_ang_vel.clone(), projected_gravity=m.projected_gravity.clone(), heading_in=heading, wrapped=wrapped,
               default_dof_pos=m.default_dof_pos.clone())
    A.check_termination(m)
    reset = m.reset_buf.clone()
    A.compute_reward(m)
    A.compute_observations(m)
    yaw_pts = mod.quat_apply_yaw(m.base_quat.repeat(1, 140), m.height_points)
    out = dict(inp, reset=reset, rew=m.rew_buf, feet_air_time_out=m.feet_air_time, obs=m.obs_buf, measured_heights=m.measured_heights,
               yaw_points=yaw_pts, height_points=m.height_points[0], knee_indices=m.knee_indices, feet_indices=m.feet_indices,
               **{"sum_" + k: v for k, v in m.episode_sums.items()})
    out = {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    out.update({"scale_" + k: np.float64(v) for k, v in m.rew_scales.items()})
    out["terrain_seed"] = np.int64(7)
    np.savez_compressed(os.path.join(OUT, "anymal_terrain.npz"), **out)
    print("anymal_terrain", n, "resets", int(reset.sum()), "rew mean", float(m.rew_buf.mean()), "obs", tuple(m.obs_buf.shape))


def shadow_hand_case(mod, n, seed):
    """compute_hand_reward / randomize_rotation (jitted) and compute_full_state (method on a mock self)."""
    g = torch.Generator().manual_seed(seed)
    nd, nf, na = 24, 5, 20
    object_pos = torch.randn(n, 3, generator=g) * 0.1 + torch.tensor([0.0, -0.39, 0.6])
    target_pos = torch.tensor([0.0, -0.39, 0.56]).repeat(n, 1) + torch.randn(n, 3, generator=g) * 0.02
    object_pos[: n // 8] += torch.tensor([0.3, 0.0, 0.0])          # some fall (goal_dist >= 0.24)
    object_rot, target_rot = rand_quat(g, n), rand_quat(g, n)
    near = torch.nn.functional.normalize(target_rot[: n // 4] + 0.03 * torch.randn(n // 4, 4, generator=g), dim=-1)
    object_rot[: n // 4] = near                                     # some successes (rot_dist <= 0.1)
    actions = torch.rand(n, na, generator=g) * 2 - 1
    reset_buf = (torch.rand(n, generator=g) < 0.05).long()
    reset_goal_buf = (torch.rand(n, generator=g) < 0.05).long()
    progress = torch.randint(0, 602, (n,), generator=g)
    progress[:4] = torch.tensor([597, 598, 599, 600])
    successes = torch.randint(0, 52, (n,), generator=g).float()
    cons = torch.tensor(3.25)
    out = {}
    for tag, mcs, igz in (("a", 0, False), ("b", 50, True)):
        sc = dict(max_episode_length=600.0, dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1, action_penalty_scale=-0.0002,
                  success_tolerance=0.1, reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=0.0 if tag == "a" else -50.0,
                  max_consecutive_successes=mcs, av_factor=0.1, ignore_z_rot=igz)
        r = mod.compute_hand_reward(torch.zeros(n), reset_buf.clone(), reset_goal_buf.clone(), progress.clone(), successes.clone(), cons.clone(),
                                    sc["max_episode_length"], object_pos, object_rot, target_pos, target_rot, sc["dist_reward_scale"],
                                    sc["rot_reward_scale"], sc["rot_eps"], actions, sc["action_penalty_scale"], sc["success_tolerance"],
                                    sc["reach_goal_bonus"], sc["fall_dist"], sc["fall_penalty"], sc["max_consecutive_successes"],
                                    sc["av_factor"], sc["ignore_z_rot"])
        for k, v in zip(("rew", "resets", "goal_resets", "progress_out", "successes_out", "cons_out"), r):
            out[f"{tag}_{k}"] = v.numpy()
        out.update({f"{tag}_scalar_{k}": np.float64(v) for k, v in sc.items()})
    rand0, rand1 = torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 2 - 1
    xu, yu = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1), torch.tensor([0.0, 1.0, 0.0]).repeat(n, 1)
    out["rand_rot"] = mod.randomize_rotation(rand0, rand1, xu, yu).numpy()
    # compute_full_state on a mock self
    m = types.SimpleNamespace()
    m.num_envs, m.num_shadow_hand_dofs, m.num_fingertips, m.num_actions = n, nd, nf, na
    lo = -torch.rand(nd, generator=g) - 0.1
    up = torch.rand(nd, generator=g) + 0.1
    m.shadow_hand_dof_lower_limits, m.shadow_hand_dof_upper_limits = lo, up
    m.shadow_hand_dof_pos = lo + torch.rand(n, nd, generator=g) * (up - lo)
    m.shadow_hand_dof_vel = torch.randn(n, nd, generator=g) * 3
    m.dof_force_tensor = torch.randn(n, nd, generator=g)
    m.vel_obs_scale, m.force_torque_obs_scale = 0.2, 10.0
    obj_state = torch.cat([object_pos, object_rot, torch.randn(n, 6, generator=g)], dim=-1)
    m.object_pose, m.object_linvel, m.object_angvel, m.object_rot = obj_state[:, 0:7], obj_state[:, 7:10], obj_state[:, 10:13], object_rot
    m.goal_pose = torch.cat([target_pos, target_rot], dim=-1)
    m.goal_rot = target_rot
    m.fingertip_state = torch.randn(n, nf, 13, generator=g)
    m.vec_sensor_tensor = torch.randn(n, 6 * nf, generator=g)
    m.actions = actions
    m.obs_buf = torch.zeros(n, 211)
    mod.ShadowHand.compute_full_state(m)
    out.update(dict(object_pos=object_pos, object_rot=object_rot, target_pos=target_pos, target_rot=target_rot, actions=actions,
                    reset_buf=reset_buf, reset_goal_buf=reset_goal_buf, progress=progress, successes=successes, cons=cons,
                    rand0=rand0, rand1=rand1, x_unit=xu, y_unit=yu, dof_lower=lo, dof_upper=up, dof_pos=m.shadow_hand_dof_pos,
                    dof_vel=m.shadow_hand_dof_vel, dof_force=m.dof_force_tensor, object_state=obj_state, goal_pose=m.goal_pose,
                    fingertip_state=m.fingertip_state, sensors=m.vec_sensor_tensor, full_state=m.obs_buf))
    out = {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, "shadow_hand.npz"), **out)
    print("shadow_hand", n, "resets a/b", int(out["a_resets"].sum()), int(out["b_resets"].sum()), "goal resets", int(out["a_goal_resets"].sum()))


def anymal_flat_case(mod, n, seed):
    """compute_anymal_observations / compute_anymal_reward (anymal.py:311-386) on random states; scales = cfg/task/Anymal.yaml."""
    g = torch.Generator().manual_seed(seed)
    root = torch.zeros(n, 13)
    root[:, :3] = torch.randn(n, 3, generator=g)
    root[:, 3:7] = rand_quat(g, n)
    root[:, 7:13] = torch.randn(n, 6, generator=g) * 1.5
    commands = torch.rand(n, 3, generator=g) * 4 - 2
    root[:192, 7:13] *= 0.15          # a share of envs tracks its command well enough for a non-zero (unclipped) reward
    commands[:192] *= 0.1
    dof_pos = torch.randn(n, 12, generator=g) * 0.5
    # defaultJointAngles of the reference YAML in the asset's dof order (LF, RF, LH, RH x HAA, HFE, KFE after collapsing)
    import yaml
    from isaacgymenvs_amd.registry import load_model
    angles = yaml.safe_load(open(os.path.join(REF, "isaacgymenvs/cfg/task/Anymal.yaml")))["env"]["defaultJointAngles"]
    default_dof_pos = torch.tensor([float(angles[nm]) for nm in load_model("anymal").dof_names]).repeat(n, 1)
    dof_vel = torch.randn(n, 12, generator=g) * 5
    gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
    actions = torch.rand(n, 12, generator=g) * 2 - 1
    torques = torch.randn(n, 12, generator=g) * 30
    torques[:96] *= 0.1
    contact = torch.zeros(n, 13, 3)
    contact[:, 0] = torch.randn(n, 3, generator=g) * (torch.rand(n, 1, generator=g) < 0.2) * 2
    knee_indices = torch.tensor([2, 5, 8, 11])
    contact[:, knee_indices] = torch.randn(n, 4, 3, generator=g) * (torch.rand(n, 4, 1, generator=g) < 0.1) * 2
    contact[:, [3, 6, 9, 12]] = torch.randn(n, 4, 3, generator=g) * 50
    episode_lengths = torch.randint(0, 2501, (n,), generator=g)
    episode_lengths[:4] = torch.tensor([2497, 2498, 2499, 2500])
    dt = 0.02
    rew_scales = {"lin_vel_xy": 1.0 * dt, "ang_vel_z": 0.5 * dt, "torque": -0.000025 * dt}
    sc = dict(lin_vel_scale=2.0, ang_vel_scale=0.25, dof_pos_scale=1.0, dof_vel_scale=0.05)
    obs = mod.compute_anymal_observations(root, commands, dof_pos, default_dof_pos, dof_vel, gravity_vec, actions, sc["lin_vel_scale"],
                                          sc["ang_vel_scale"], sc["dof_pos_scale"], sc["dof_vel_scale"])
    rew, reset = mod.compute_anymal_reward(root, commands, torques, contact, knee_indices, episode_lengths, rew_scales, 0, 2500)
    out = dict(root_states=root.numpy(), commands=commands.numpy(), dof_pos=dof_pos.numpy(), default_dof_pos=default_dof_pos.numpy(),
               dof_vel=dof_vel.numpy(), gravity_vec=gravity_vec.numpy(), actions=actions.numpy(), torques=torques.numpy(),
               contact_forces=contact.numpy(), knee_indices=knee_indices.numpy(), episode_lengths=episode_lengths.numpy(),
               obs=obs.numpy(), rew=rew.numpy(), reset=reset.numpy(), scalar_max_episode_length=2500, scalar_base_index=0)
    out.update({"scalar_" + k: v for k, v in sc.items()})
    out.update({"scalar_rew_" + k: v for k, v in rew_scales.items()})
    np.savez_compressed(os.path.join(OUT, "anymal_flat.npz"), **out)
    print("anymal_flat", n, "resets", int(reset.sum()), "rew mean", float(rew.mean()), "obs", tuple(obs.shape))


def quadcopter_case(mod, n, seed):
    """compute_quadcopter_reward (quadcopter.py:348-386) on random root states around the hover target."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(n, 3, generator=g) * 1.5 + torch.tensor([0.0, 0.0, 1.0])
    pos[:8, 2] = torch.tensor([0.29, 0.3, 0.31, 1.0, 1.0, 1.0, 1.0, 1.0])
    pos[8:12] = torch.tensor([[3.0001, 0, 1.0], [2.9999, 0, 1.0], [0, 0, 1.0], [0, 0, 4.1]])
    quat = torch.randn(n, 4, generator=g); quat[:, 3] += 2.0
    quat = quat / quat.norm(dim=-1, keepdim=True)
    linvel = torch.randn(n, 3, generator=g)
    angvel = torch.randn(n, 3, generator=g) * 3
    reset_in = (torch.rand(n, generator=g) < 0.1).long()
    progress = torch.randint(0, 501, (n,), generator=g)
    progress[:4] = torch.tensor([497, 498, 499, 500])
    rew, reset = mod.compute_quadcopter_reward(pos, quat, linvel, angvel, reset_in, progress, 500.0)
    np.savez_compressed(os.path.join(OUT, "quadcopter_reward.npz"), root_positions=pos.numpy(), root_quats=quat.numpy(),
                        root_linvels=linvel.numpy(), root_angvels=angvel.numpy(), reset_in=reset_in.numpy(), progress=progress.numpy(),
                        rew=rew.numpy(), reset=reset.numpy(), scalar_max_episode_length=500.0)
    print("quadcopter_reward", n, "resets", int(reset.sum()), "rew mean", float(rew.mean()))


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from isaacgymenvs_amd.registry import load_model
    torch.set_num_threads(1)
    mods = import_reference()
    os.makedirs(OUT, exist_ok=True)
    ant, hum = load_model("ant"), load_model("humanoid")
    a_lo, a_up = np.minimum(ant.dof_lower, ant.dof_upper), np.maximum(ant.dof_lower, ant.dof_upper)
    h_lo, h_up = np.minimum(hum.dof_lower, hum.dof_upper), np.maximum(hum.dof_lower, hum.dof_upper)
    ant_s = dict(dt=0.0166, dof_vel_scale=0.2, contact_force_scale=0.1, angular_velocity_scale=1.0, up_weight=0.1,
                 heading_weight=0.5, actions_cost=0.005, energy_cost=0.05, joints_at_limit_cost=0.1,
                 termination_height=0.31, death_cost=-2.0, max_episode_length=1000.0)
    hum_s = dict(dt=0.0166, dof_vel_scale=0.1, contact_force_scale=0.01, angular_velocity_scale=0.25, up_weight=0.1,
                 heading_weight=0.5, actions_cost=0.01, energy_cost=0.05, joints_at_limit_cost=0.25,
                 termination_height=0.8, death_cost=-1.0, max_episode_length=1000.0)
    locomotion_case(mods["ant"], "ant_obs_reward", 8, 24, 512, 1, False, a_lo, a_up, list(ant.act_gear), ant_s)
    locomotion_case(mods["humanoid"], "humanoid_obs_reward", 21, 12, 512, 2, True, h_lo, h_up, list(hum.act_gear), hum_s)
    cartpole_case(mods["cartpole"], 512, 3)
    anymal_case(mods["anymal_terrain"], 256, 4)
    shadow_hand_case(mods["shadow_hand"], 512, 5)
    anymal_flat_case(mods["anymal"], 512, 6)
    quadcopter_case(mods["quadcopter"], 512, 7)


if __name__ == "__main__":
    main()

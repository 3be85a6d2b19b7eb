"""GPU parity of the stand-alone HIP replacements of the reference's remaining jitted task functions (csrc/kernels_jit_twins.hip,
through the C ABI) against the outputs of the REFERENCE's own functions (tests/golden/jit_twins_*.npz, tools/gen_golden_jit_twins.py).
Stated tolerance: flags / counters bit-exact, floats <= a few fp32 ulps of the elementwise maths (device sin / tanh / exp vs ATen's)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from isaacgymenvs_amd import native  # noqa: E402

DEV = "cuda:0"


def _t(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV).contiguous()


def _i(a):
    return _t(a, torch.int64)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, "jit_twins_" + name + ".npz")))


def _np(x):
    torch.cuda.synchronize()
    return x.cpu().numpy()


def test_bbot_and_ingenuity_reward_kernels(golden_dir):
    L = native.lib()
    g = _load(golden_dir, "bbot")
    n = len(g["rew"])
    bp, bv, ri, pr = _t(g["ball_positions"]), _t(g["ball_velocities"]), _i(g["reset_in"]), _i(g["progress"])
    rew, reset = torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    native.check(L.mi_compute_bbot_reward(n, None, bp.data_ptr(), bv.data_ptr(), float(g["scalar_ball_radius"]), ri.data_ptr(), pr.data_ptr(),
                                          float(g["scalar_max_episode_length"]), rew.data_ptr(), reset.data_ptr(), _stream()))
    np.testing.assert_array_equal(_np(reset), g["reset"])
    np.testing.assert_allclose(_np(rew), g["rew"], rtol=2e-6)
    g = _load(golden_dir, "ingenuity")
    n = len(g["rew"])
    a = [_t(g[k]) for k in ("root_positions", "target_root_positions", "root_quats", "root_linvels", "root_angvels")]
    pr = _i(g["progress"])
    rew, reset = torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    native.check(L.mi_compute_ingenuity_reward(n, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(), None,
                                               pr.data_ptr(), float(g["scalar_max_episode_length"]), rew.data_ptr(), reset.data_ptr(), _stream()))
    np.testing.assert_array_equal(_np(reset), g["reset"])
    np.testing.assert_allclose(_np(rew), g["rew"], rtol=3e-6, atol=1e-7)


def test_franka_cabinet_kernels(golden_dir):
    L = native.lib()
    g = _load(golden_dir, "franka_cabinet")
    n = len(g["rew"])
    p = native.MiFrankaCabinetRewardParams(*[float(g["scalar_" + f[0]]) for f in native.MiFrankaCabinetRewardParams._fields_])
    names = ("franka_grasp_pos", "drawer_grasp_pos", "franka_grasp_rot", "drawer_grasp_rot", "franka_lfinger_pos", "franka_rfinger_pos",
             "gripper_forward_axis", "drawer_inward_axis", "gripper_up_axis", "drawer_up_axis")
    t = [_t(g[k]) for k in names]
    ri, pr, act, cab = _i(g["reset_in"]), _i(g["progress"]), _t(g["actions"]), _t(g["cabinet_dof_pos"])
    rew, reset = torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    native.check(L.mi_compute_franka_cabinet_reward(n, C.byref(p), ri.data_ptr(), pr.data_ptr(), act.data_ptr(), act.shape[1], cab.data_ptr(),
                                                    cab.shape[1], *[x.data_ptr() for x in t], rew.data_ptr(), reset.data_ptr(), _stream()))
    np.testing.assert_array_equal(_np(reset), g["reset"])
    np.testing.assert_allclose(_np(rew), g["rew"], rtol=3e-6, atol=2e-6)
    # argument validation: the drawer column must exist
    assert L.mi_compute_franka_cabinet_reward(n, C.byref(p), ri.data_ptr(), pr.data_ptr(), act.data_ptr(), act.shape[1], cab.data_ptr(), 3,
                                              *[x.data_ptr() for x in t], rew.data_ptr(), reset.data_ptr(), _stream()) == -1
    assert b"drawer_top_joint" in L.mi_last_error()
    g = _load(golden_dir, "grasp_transforms")
    ins = [_t(g[k]) for k in ("hand_rot", "hand_pos", "franka_local_grasp_rot", "franka_local_grasp_pos", "drawer_rot", "drawer_pos",
                              "drawer_local_grasp_rot", "drawer_local_grasp_pos")]
    outs = [torch.empty(n, k, device=DEV) for k in (4, 3, 4, 3)]
    native.check(L.mi_compute_grasp_transforms(n, *[x.data_ptr() for x in ins], *[x.data_ptr() for x in outs], _stream()))
    for o, k in zip(outs, ("global_franka_rot", "global_franka_pos", "global_drawer_rot", "global_drawer_pos")):
        np.testing.assert_allclose(_np(o), g[k], atol=1e-6)


def test_franka_cube_stack_kernels(golden_dir):
    L = native.lib()
    a = _load(golden_dir, "axisangle2quat")
    n = len(a["vec"])
    vec, quat = _t(a["vec"]), torch.empty(n, 4, device=DEV)
    native.check(L.mi_axisangle2quat(n, vec.data_ptr(), float(a["scalar_eps"]), quat.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(quat), a["quat"], atol=3e-7)
    g = _load(golden_dir, "franka_cube_stack")
    n = len(g["rew"])
    p = native.MiFrankaCubeStackRewardParams(*[float(g["scalar_" + f[0]]) for f in native.MiFrankaCubeStackRewardParams._fields_])
    t = [_t(g[k]) for k in ("cubeA_size", "cubeB_size", "cubeA_pos", "cubeA_pos_relative", "eef_lf_pos", "eef_rf_pos", "cubeA_to_cubeB_pos")]
    ri, pr = _i(g["reset_in"]), _i(g["progress"])
    rew, reset = torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    native.check(L.mi_compute_franka_cube_stack_reward(n, C.byref(p), ri.data_ptr(), pr.data_ptr(), *[x.data_ptr() for x in t], rew.data_ptr(),
                                                       reset.data_ptr(), _stream()))
    np.testing.assert_array_equal(_np(reset), g["reset"])
    np.testing.assert_allclose(_np(rew), g["rew"], rtol=1e-5, atol=3e-6)


def test_allegro_hand_reward_and_pen_rotation_kernels(golden_dir):
    L = native.lib()
    for tag in ("a", "b"):
        g = _load(golden_dir, "allegro_hand_reward_" + tag)
        n = len(g["rew"])
        p = native.MiHandRewardParams(float(g["scalar_max_episode_length"]), float(g["scalar_dist_reward_scale"]), float(g["scalar_rot_reward_scale"]),
                                      float(g["scalar_rot_eps"]), float(g["scalar_action_penalty_scale"]), float(g["scalar_success_tolerance"]),
                                      float(g["scalar_reach_goal_bonus"]), float(g["scalar_fall_dist"]), float(g["scalar_fall_penalty"]),
                                      int(g["scalar_max_consecutive_successes"]), float(g["scalar_av_factor"]), int(g["scalar_ignore_z_rot"]))
        rew = torch.zeros(n, device=DEV)
        rs, gr, pr, su = _i(g["reset_in"]), _i(g["reset_goal_in"]), _i(g["progress_in"]), _t(g["successes_in"])
        cs, ws = _t(g["consecutive_successes_in"]), torch.zeros(2, device=DEV)
        ins = [_t(g[k]) for k in ("object_pos", "object_rot", "target_pos", "target_rot", "actions")]
        native.check(L.mi_compute_hand_reward(n, C.byref(p), rew.data_ptr(), rs.data_ptr(), gr.data_ptr(), pr.data_ptr(), su.data_ptr(), cs.data_ptr(),
                                              *[x.data_ptr() for x in ins], ins[4].shape[1], ws.data_ptr(), _stream()))
        np.testing.assert_allclose(_np(rew), g["rew"], rtol=2e-5, atol=1e-5)
        for o, k in zip((rs, gr, pr, su), ("resets", "goal_resets", "progress", "successes")):
            np.testing.assert_array_equal(_np(o), g[k])
        np.testing.assert_allclose(_np(cs), g["cons_successes"], rtol=2e-6)
    g = _load(golden_dir, "rotation_pen")
    n = len(g["rand0"])
    ins = [_t(g[k]) for k in ("rand0", "rand1", "x_unit", "y_unit", "z_unit")]
    out = torch.empty(n, 4, device=DEV)
    native.check(L.mi_randomize_rotation_pen(n, ins[0].data_ptr(), ins[1].data_ptr(), float(g["scalar_max_angle"]), ins[2].data_ptr(),
                                             ins[3].data_ptr(), ins[4].data_ptr(), out.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out), g["out"], atol=4e-7)


def test_trifinger_kernels(golden_dir):
    L = native.lib()
    k = _load(golden_dir, "lgsk")
    n = len(k["x"])
    x, out = _t(k["x"]), torch.empty(n, device=DEV)
    for scale, key in ((50.0, "out_50"), (30.0, "out_30")):
        native.check(L.mi_lgsk_kernel(n, x.data_ptr(), scale, 2.0, out.data_ptr(), _stream()))
        np.testing.assert_allclose(_np(out), k[key], rtol=3e-6)
    kp = _load(golden_dir, "keypoints")
    pose, pts = _t(kp["pose"]), torch.empty(n, 8, 3, device=DEV)
    size = (C.c_float * 3)(*[float(v) for v in kp["size"]])
    native.check(L.mi_gen_keypoints(n, pose.data_ptr(), 7, size, pts.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(pts), kp["out"], atol=2e-7)
    g = _load(golden_dir, "trifinger_reward")
    t = [_t(g[k_]) for k_ in ("object_goal_poses", "object_state", "last_object_state", "fingertip_state", "last_fingertip_state")]
    pr = _i(g["progress"])
    for tag, use_kp in (("kp", 1), ("pose", 0), ("late", 1)):
        p = native.MiTrifingerRewardParams(int(g["scalar_episode_length"]), float(g["scalar_dt"]), float(g["scalar_finger_move_penalty_weight"]),
                                           float(g["scalar_finger_reach_object_weight"]), float(g["scalar_object_dist_weight"]),
                                           float(g["scalar_object_rot_weight"]), int(g["scalar_steps_" + tag]), use_kp, (C.c_float * 3)(0.065, 0.065, 0.065))
        rew, reset = torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
        mv, rc = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        native.check(L.mi_compute_trifinger_reward(n, C.byref(p), pr.data_ptr(), *[x.data_ptr() for x in t], rew.data_ptr(), reset.data_ptr(),
                                                   mv.data_ptr(), rc.data_ptr(), _stream()))
        np.testing.assert_array_equal(_np(reset), g["reset_" + tag])
        np.testing.assert_allclose(_np(rew), g["rew_" + tag], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(_np(mv), g["info_move_" + tag], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(rc), g["info_reach_" + tag], rtol=1e-5, atol=2e-4)
    o = _load(golden_dir, "trifinger_obs")
    ins = [_t(o[k_]) for k_ in ("dof_position", "dof_velocity", "object_state", "object_goal_poses", "actions", "fingertip_state", "joint_torques",
                                "tip_wrenches")]
    for asym, tag in ((0, "sym"), (1, "asym")):
        obs = torch.empty(n, o["obs_" + tag].shape[1], device=DEV)
        st = torch.empty(n, o["states_" + tag].shape[1], device=DEV)
        native.check(L.mi_compute_trifinger_observations_states(n, asym, 9, 9, 39, 18, *[x.data_ptr() for x in ins], obs.data_ptr(), st.data_ptr(),
                                                                _stream()))
        np.testing.assert_array_equal(_np(obs), o["obs_" + tag])
        np.testing.assert_array_equal(_np(st), o["states_" + tag])


def test_trifinger_sampler_kernels(golden_dir):
    L = native.lib()
    g = _load(golden_dir, "trifinger_samplers")
    n = len(g["z"])
    out2, out1, out4, out3 = torch.empty(n, 2, device=DEV), torch.empty(n, device=DEV), torch.empty(n, 4, device=DEV), torch.empty(n, 3, device=DEV)
    u = _t(g["rand_xy"])
    native.check(L.mi_trifinger_random_xy(n, u.data_ptr(), float(g["scalar_max_dist"]), out2.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out2), g["xy"], atol=3e-8)
    u = _t(g["rand_z"])
    native.check(L.mi_trifinger_random_z(n, u.data_ptr(), float(g["scalar_min_height"]), float(g["scalar_max_height"]), out1.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out1), g["z"], atol=1e-8)
    native.check(L.mi_trifinger_default_orientation(n, out4.data_ptr(), _stream()))
    np.testing.assert_array_equal(_np(out4), g["default"])
    u = _t(g["randn_orientation"])
    native.check(L.mi_trifinger_random_orientation(n, u.data_ptr(), out4.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out4), g["orientation"], atol=2e-7)
    u, b = _t(g["rand_within"]), _t(g["base"])
    native.check(L.mi_trifinger_random_orientation_within_angle(n, u.data_ptr(), b.data_ptr(), float(g["scalar_max_angle"]), out4.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out4), g["within"], atol=3e-6)     # sqrt((1 - cos)/2) amplifies the last bit of cos
    u = _t(g["randn_angvel"])
    native.check(L.mi_trifinger_random_angular_vel(n, u.data_ptr(), float(g["scalar_magnitude_stdev"]), out3.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out3), g["angvel"], atol=4e-7)
    u = _t(g["rand_yaw"])
    native.check(L.mi_trifinger_random_yaw_orientation(n, u.data_ptr(), out4.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out4), g["yaw"], atol=3e-7)
    assert L.mi_trifinger_random_xy(n, None, 0.1, out2.data_ptr(), _stream()) == -1 and b"null argument" in L.mi_last_error()


def test_humanoid_amp_kernels(golden_dir):
    L = native.lib()
    d = _load(golden_dir, "amp_dof_to_obs")
    n = len(d["pose"])
    pose, out = _t(d["pose"]), torch.empty(n, 52, device=DEV)
    native.check(L.mi_amp_dof_to_obs(n, pose.data_ptr(), out.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(out), d["out"], atol=1e-6)
    g = _load(golden_dir, "amp_obs_reset")
    root, q, qd, key = _t(g["root_states"]), _t(g["dof_pos"]), _t(g["dof_vel"]), _t(g["key_body_pos"])
    nk = g["key_body_pos"].shape[1]
    for local, tag in ((1, "obs_local"), (0, "obs_global")):
        obs = torch.empty(n, g[tag].shape[1], device=DEV)
        native.check(L.mi_compute_humanoid_amp_observations(n, root.data_ptr(), q.data_ptr(), qd.data_ptr(), key.data_ptr(), nk, local,
                                                            obs.data_ptr(), _stream()))
        np.testing.assert_allclose(_np(obs), g[tag], atol=3e-6)
    rew = torch.zeros(n, device=DEV)
    native.check(L.mi_compute_humanoid_amp_reward(n, None, rew.data_ptr(), _stream()))      # humanoid_amp_base.py:530-534
    assert bool((_np(rew) == 1.0).all())
    contact, pos, pr = _t(g["contact_buf"]), _t(g["rigid_body_pos"]), _i(g["progress"])
    ids = (C.c_int64 * len(g["contact_body_ids"]))(*[int(v) for v in g["contact_body_ids"]])
    for early, tag in ((1, "early"), (0, "noearly")):
        reset, term = torch.empty(n, device=DEV, dtype=torch.int64), torch.empty(n, device=DEV, dtype=torch.int64)
        native.check(L.mi_compute_humanoid_amp_reset(n, None, pr.data_ptr(), contact.data_ptr(), ids, len(ids), pos.data_ptr(), pos.shape[1],
                                                     float(g["scalar_max_episode_length"]), early, float(g["scalar_termination_height"]),
                                                     reset.data_ptr(), term.data_ptr(), _stream()))
        np.testing.assert_array_equal(_np(reset), g["reset_" + tag])
        np.testing.assert_array_equal(_np(term), g["terminated_" + tag])


def test_dextreme_hand_reward_kernel(golden_dir):
    L = native.lib()
    g = _load(golden_dir, "dextreme_reward")
    n = len(g["rew"])
    P = native.MiDextremeRewardParams
    p = P(*[(int if f[1] is C.c_int32 else float)(g["scalar_" + f[0]]) for f in P._fields_])
    rew = torch.zeros(n, device=DEV)
    rs, gr, pr, hold = _i(g["reset_in"]), _i(g["reset_goal_in"]), _i(g["progress_in"]), _i(g["hold_count_in"])
    su, cs, ws = _t(g["successes_in"]), _t(g["consecutive_successes_in"]), torch.zeros(2, device=DEV)
    ct, pt, dv = _t(g["cur_targets"]), _t(g["prev_targets"]), _t(g["hand_dof_vel"])
    ins = [_t(g[k]) for k in ("object_pos", "object_rot", "target_pos", "target_rot", "actions")]
    terms = torch.empty(8, n, device=DEV)
    native.check(L.mi_compute_hand_reward_dextreme(n, C.byref(p), rew.data_ptr(), rs.data_ptr(), gr.data_ptr(), pr.data_ptr(), hold.data_ptr(),
                                                   ct.data_ptr(), pt.data_ptr(), dv.data_ptr(), ct.shape[1], su.data_ptr(), cs.data_ptr(),
                                                   *[x.data_ptr() for x in ins], ins[4].shape[1], terms.data_ptr(), ws.data_ptr(), _stream()))
    np.testing.assert_allclose(_np(rew), g["rew"], rtol=2e-5, atol=2e-5)
    for o, k in zip((rs, gr, pr, hold, su), ("resets", "goal_resets", "progress", "hold_count", "successes")):
        np.testing.assert_array_equal(_np(o), g[k], err_msg=k)
    np.testing.assert_allclose(_np(cs), np.atleast_1d(g["cons_successes"]), rtol=2e-6)
    for i, k in enumerate(("dist_rew", "rot_rew", "action_penalty", "action_delta_penalty", "velocity_penalty", "reach_goal_rew", "fall_rew",
                           "timeout_rew")):
        np.testing.assert_allclose(_np(terms[i]), g[k], rtol=2e-5, atol=1e-5, err_msg=k)

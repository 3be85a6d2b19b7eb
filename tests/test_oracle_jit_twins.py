"""Pin oracle/jit_twins.py (numpy fp32) against the outputs of the REFERENCE's own @torch.jit.script functions
(tools/gen_golden_jit_twins.py -> tests/golden/jit_twins_*.npz).  Reset / flag outputs bit-exact; floats to fp32 round-off of
the elementwise maths (libm sin / tanh / exp differ from ATen's by an ulp or two)."""
import os

import numpy as np

from oracle import jit_twins as J
from oracle import tasks as T


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, "jit_twins_" + name + ".npz")))


def test_bbot_reward(golden_dir):
    g = _load(golden_dir, "bbot")
    rew, reset = J.compute_bbot_reward(g["tray_positions"], g["ball_positions"], g["ball_velocities"], float(g["scalar_ball_radius"]),
                                       g["reset_in"], g["progress"], float(g["scalar_max_episode_length"]))
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_allclose(rew, g["rew"], rtol=1e-6)
    assert g["reset"][0] == 1 and g["reset"][3] == 1          # ball below 1.5 radii


def test_ingenuity_reward(golden_dir):
    g = _load(golden_dir, "ingenuity")
    rew, reset = J.compute_ingenuity_reward(g["root_positions"], g["target_root_positions"], g["root_quats"], g["root_linvels"],
                                            g["root_angvels"], g["reset_in"], g["progress"], float(g["scalar_max_episode_length"]))
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_allclose(rew, g["rew"], rtol=2e-6, atol=1e-7)
    assert g["reset"][0] == 1 and g["reset"][4] == 1 and g["reset"].sum() < len(g["reset"])


def _cabinet_scalars(g):
    return [float(g["scalar_" + k]) for k in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale",
                                               "finger_dist_reward_scale", "action_penalty_scale", "distX_offset", "max_episode_length")]


def test_franka_cabinet_reward_and_grasp_transforms(golden_dir):
    g = _load(golden_dir, "franka_cabinet")
    rew, reset = J.compute_franka_cabinet_reward(
        g["reset_in"], g["progress"], g["actions"], g["cabinet_dof_pos"], g["franka_grasp_pos"], g["drawer_grasp_pos"], g["franka_grasp_rot"],
        g["drawer_grasp_rot"], g["franka_lfinger_pos"], g["franka_rfinger_pos"], g["gripper_forward_axis"], g["drawer_inward_axis"],
        g["gripper_up_axis"], g["drawer_up_axis"], len(g["rew"]), *_cabinet_scalars(g))
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_allclose(rew, g["rew"], rtol=3e-6, atol=2e-6)
    assert (g["rew"] == -1).any() and (g["rew"] > 0).any()    # both the "bad style" override and regular rewards are exercised
    t = _load(golden_dir, "grasp_transforms")
    out = J.compute_grasp_transforms(t["hand_rot"], t["hand_pos"], t["franka_local_grasp_rot"], t["franka_local_grasp_pos"], t["drawer_rot"],
                                     t["drawer_pos"], t["drawer_local_grasp_rot"], t["drawer_local_grasp_pos"])
    for o, k in zip(out, ("global_franka_rot", "global_franka_pos", "global_drawer_rot", "global_drawer_pos")):
        np.testing.assert_allclose(o, t[k], atol=1e-6)


def test_franka_cube_stack_reward_and_axisangle2quat(golden_dir):
    a = _load(golden_dir, "axisangle2quat")
    np.testing.assert_allclose(J.axisangle2quat(a["vec"], float(a["scalar_eps"])), a["quat"], atol=2e-7)
    assert list(a["quat"][0]) == [0, 0, 0, 1] and list(a["quat"][1]) == [0, 0, 0, 1] and a["quat"][2][0] > 0
    g = _load(golden_dir, "franka_cube_stack")
    states = {k: g[k] for k in ("cubeA_size", "cubeB_size", "cubeA_pos", "cubeA_pos_relative", "eef_lf_pos", "eef_rf_pos", "cubeA_to_cubeB_pos")}
    rs = {k: float(g["scalar_" + k]) for k in ("r_dist_scale", "r_lift_scale", "r_align_scale", "r_stack_scale", "table_height")}
    rew, reset = J.compute_franka_cube_stack_reward(g["reset_in"], g["progress"], g["actions"], states, rs, float(g["scalar_max_episode_length"]))
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_allclose(rew, g["rew"], rtol=1e-5, atol=2e-6)
    assert (g["rew"] == 16.0).sum() >= 3 and (g["rew"] < 16.0).any()   # stacked and not-stacked cases


def test_allegro_hand_reward_is_the_shadow_hand_function(golden_dir):
    for tag in ("a", "b"):
        g = _load(golden_dir, "allegro_hand_reward_" + tag)
        out = T.compute_hand_reward(None, g["reset_in"], g["reset_goal_in"], g["progress_in"], g["successes_in"], float(g["consecutive_successes_in"][0]),
                                    float(g["scalar_max_episode_length"]), g["object_pos"], g["object_rot"], g["target_pos"], g["target_rot"],
                                    float(g["scalar_dist_reward_scale"]), float(g["scalar_rot_reward_scale"]), float(g["scalar_rot_eps"]), g["actions"],
                                    float(g["scalar_action_penalty_scale"]), float(g["scalar_success_tolerance"]), float(g["scalar_reach_goal_bonus"]),
                                    float(g["scalar_fall_dist"]), float(g["scalar_fall_penalty"]), int(g["scalar_max_consecutive_successes"]),
                                    float(g["scalar_av_factor"]), bool(g["scalar_ignore_z_rot"]))
        np.testing.assert_allclose(out[0], g["rew"], rtol=2e-5, atol=1e-5)
        for o, k in zip(out[1:5], ("resets", "goal_resets", "progress", "successes")):
            np.testing.assert_array_equal(o, g[k])
        np.testing.assert_allclose(out[5], g["cons_successes"][0], rtol=1e-6)
        assert g["goal_resets"].sum() > 0 and g["resets"].sum() > 0


def test_randomize_rotation_pen(golden_dir):
    g = _load(golden_dir, "rotation_pen")
    out = J.randomize_rotation_pen(g["rand0"], g["rand1"], float(g["scalar_max_angle"]), g["x_unit"], g["y_unit"], g["z_unit"])
    np.testing.assert_allclose(out, g["out"], atol=3e-7)


def test_trifinger_functions(golden_dir):
    k = _load(golden_dir, "lgsk")
    np.testing.assert_allclose(J.lgsk_kernel(k["x"], 50.0, 2.0), k["out_50"], rtol=2e-6)
    np.testing.assert_allclose(J.lgsk_kernel(k["x"], 30.0, 2.0), k["out_30"], rtol=2e-6)
    kp = _load(golden_dir, "keypoints")
    np.testing.assert_allclose(J.gen_keypoints(kp["pose"]), kp["out"], atol=2e-7)
    g = _load(golden_dir, "trifinger_reward")
    for tag, use_kp in (("kp", True), ("pose", False), ("late", True)):
        rew, reset, info = J.compute_trifinger_reward(
            None, g["reset_in"], g["progress"], int(g["scalar_episode_length"]), float(g["scalar_dt"]), float(g["scalar_finger_move_penalty_weight"]),
            float(g["scalar_finger_reach_object_weight"]), float(g["scalar_object_dist_weight"]), float(g["scalar_object_rot_weight"]),
            int(g["scalar_steps_" + tag]), g["object_goal_poses"], g["object_state"], g["last_object_state"], g["fingertip_state"],
            g["last_fingertip_state"], use_kp)
        np.testing.assert_array_equal(reset, g["reset_" + tag])
        # the reach term is a small difference of norms times 250: its absolute round-off is what bounds the sum
        np.testing.assert_allclose(rew, g["rew_" + tag], rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(info["finger_movement_penalty"], g["info_move_" + tag], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(info["finger_reach_object_reward"], g["info_reach_" + tag], rtol=1e-5, atol=2e-4)
    assert np.all(g["info_reach_late"] == 0) and np.any(g["info_reach_kp"] != 0)      # the 5e7-step schedule switches the term off
    o = _load(golden_dir, "trifinger_obs")
    for asym, tag in ((False, "sym"), (True, "asym")):
        obs, st = J.compute_trifinger_observations_states(asym, o["dof_position"], o["dof_velocity"], o["object_state"], o["object_goal_poses"],
                                                          o["actions"], o["fingertip_state"], o["joint_torques"], o["tip_wrenches"])
        np.testing.assert_array_equal(obs, o["obs_" + tag])
        np.testing.assert_array_equal(st, o["states_" + tag])


def test_humanoid_amp_functions(golden_dir):
    d = _load(golden_dir, "amp_dof_to_obs")
    np.testing.assert_allclose(J.amp_dof_to_obs(d["pose"]), d["out"], atol=5e-7)
    g = _load(golden_dir, "amp_obs_reset")
    np.testing.assert_allclose(J.compute_humanoid_amp_observations(g["root_states"], g["dof_pos"], g["dof_vel"], g["key_body_pos"], True),
                               g["obs_local"], atol=2e-6)
    np.testing.assert_allclose(J.compute_humanoid_amp_observations(g["root_states"], g["dof_pos"], g["dof_vel"], g["key_body_pos"], False),
                               g["obs_global"], atol=2e-6)
    np.testing.assert_array_equal(g["amp_obs"], g["obs_local"])     # build_amp_observations is the same function in the reference
    for early, tag in ((True, "early"), (False, "noearly")):
        reset, term = J.compute_humanoid_amp_reset(np.zeros_like(g["progress"]), g["progress"], g["contact_buf"], g["contact_body_ids"],
                                                   g["rigid_body_pos"], float(g["scalar_max_episode_length"]), early,
                                                   float(g["scalar_termination_height"]))
        np.testing.assert_array_equal(reset, g["reset_" + tag])
        np.testing.assert_array_equal(term, g["terminated_" + tag])
    assert 0 < g["terminated_early"].sum() < len(g["progress"]) and g["terminated_noearly"].sum() == 0


def test_dextreme_hand_reward(golden_dir):
    path = os.path.join(golden_dir, "jit_twins_dextreme_reward.npz")
    g = dict(np.load(path))
    out = J.compute_hand_reward_dextreme(
        None, g["reset_in"], g["reset_goal_in"], g["progress_in"], g["hold_count_in"], g["cur_targets"], g["prev_targets"], g["hand_dof_vel"],
        g["successes_in"], float(g["consecutive_successes_in"][0]), float(g["scalar_max_episode_length"]), g["object_pos"], g["object_rot"],
        g["target_pos"], g["target_rot"], float(g["scalar_dist_reward_scale"]), float(g["scalar_rot_reward_scale"]), float(g["scalar_rot_eps"]),
        g["actions"], float(g["scalar_action_penalty_scale"]), float(g["scalar_action_delta_penalty_scale"]), float(g["scalar_success_tolerance"]),
        float(g["scalar_reach_goal_bonus"]), float(g["scalar_fall_dist"]), float(g["scalar_fall_penalty"]), int(g["scalar_max_consecutive_successes"]),
        float(g["scalar_av_factor"]), int(g["scalar_num_success_hold_steps"]))
    names = ("rew", "resets", "goal_resets", "progress", "hold_count", "successes", "cons_successes", "dist_rew", "rot_rew", "action_penalty",
             "action_delta_penalty", "velocity_penalty", "reach_goal_rew", "fall_rew", "timeout_rew")
    for o, k in zip(out, names):
        if k in ("resets", "goal_resets", "progress", "hold_count", "successes"):
            np.testing.assert_array_equal(o, g[k], err_msg=k)
        else:
            np.testing.assert_allclose(o, np.squeeze(g[k]), rtol=2e-5, atol=1e-5, err_msg=k)
    assert g["goal_resets"].sum() > 0 and g["resets"].sum() > 0 and (g["hold_count"] > 0).any()


def test_trifinger_samplers_on_the_reference_draws(golden_dir):
    """trifinger.py:1427-1512 draw inside the jitted function; the golden file holds the draws the global generator handed them
    (replayed with the same seed) next to their outputs."""
    g = _load(golden_dir, "trifinger_samplers")
    x, y = J.tri_random_xy(g["rand_xy"], float(g["scalar_max_dist"]))
    np.testing.assert_allclose(np.stack([x, y], -1), g["xy"], atol=2e-8)
    np.testing.assert_allclose(J.tri_random_z(g["rand_z"], float(g["scalar_min_height"]), float(g["scalar_max_height"])), g["z"], atol=1e-8)
    np.testing.assert_allclose(J.tri_random_orientation(g["randn_orientation"]), g["orientation"], atol=2e-7)
    np.testing.assert_allclose(J.tri_random_orientation_within_angle(g["rand_within"], g["base"], float(g["scalar_max_angle"])), g["within"], atol=3e-6)   # sqrt((1 - cos)/2) amplifies the last bit of cos for small angles
    np.testing.assert_allclose(J.tri_random_angular_vel(g["randn_angvel"], float(g["scalar_magnitude_stdev"])), g["angvel"], atol=3e-7)
    np.testing.assert_allclose(J.tri_random_yaw_orientation(g["rand_yaw"]), g["yaw"], atol=2e-7)
    assert np.array_equal(g["default"], np.tile(np.array([0, 0, 0, 1], np.float32), (len(g["default"]), 1)))
    # reference quirk kept as is: the construction uses sqrt(1 - z^2) with the already n-scaled z (trifinger.py:1487-1488), so the vector
    # part is longer than sin(theta / 2) before the re-normalisation and the result can exceed max_angle (0.78 rad here for 0.6)
    d = J.quat_diff_rad(g["within"], g["base"])
    assert float(g["scalar_max_angle"]) < float(d.max()) < 1.5 * float(g["scalar_max_angle"])

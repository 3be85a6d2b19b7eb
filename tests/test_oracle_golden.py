"""Pin the numpy oracle (oracle/tasks.py) against golden vectors produced by the REFERENCE's own jitted functions
(tools/gen_golden.py -> tests/golden/*.npz).  Tolerances: fp32 round-off of elementwise maths; `potentials` is
~6e4 so one ulp is 4e-3, which also bounds the progress term of the reward."""
import os

import numpy as np
import pytest

from oracle import tasks as T


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.mark.parametrize("name,hum", [("ant_obs_reward.npz", False), ("humanoid_obs_reward.npz", True)])
def test_locomotion_observations_match_reference(golden_dir, name, hum):
    g = _load(golden_dir, name)
    obs, pot, prev, upv, hv = T.compute_locomotion_observations(
        hum, g["root_states"], g["targets"], g["potentials_in"], g["inv_start_rot"], g["dof_pos"], g["dof_vel"],
        g["dof_force"], g["dof_limits_lower"], g["dof_limits_upper"], g["scalar_dof_vel_scale"], g["sensors"],
        g["actions"], g["scalar_dt"], g["scalar_contact_force_scale"], g["scalar_angular_velocity_scale"],
        g["basis_vec0"], g["basis_vec1"])
    assert obs.shape == g["obs"].shape
    # angles near the +-pi / 0|2pi wrap may legitimately land on either side by one ulp: compare on the circle
    ang = [7, 8, 9]
    d = np.abs(obs - g["obs"])
    d[:, ang] = np.minimum(d[:, ang], np.abs(d[:, ang] - 2 * np.pi))
    assert d.max() < 2e-5, d.max()
    np.testing.assert_allclose(pot, g["potentials"], rtol=2e-7)
    np.testing.assert_array_equal(prev, g["prev_potentials"])
    np.testing.assert_allclose(upv, g["up_vec"], atol=2e-6)
    np.testing.assert_allclose(hv, g["heading_vec"], atol=2e-6)


@pytest.mark.parametrize("name,hum", [("ant_obs_reward.npz", False), ("humanoid_obs_reward.npz", True)])
def test_locomotion_reward_matches_reference(golden_dir, name, hum):
    g = _load(golden_dir, name)
    rew, reset = T.compute_locomotion_reward(
        hum, g["obs"], g["reset_in"], g["progress"], g["actions"], g["scalar_up_weight"], g["scalar_heading_weight"],
        g["potentials"], g["prev_potentials"], g["scalar_actions_cost"], g["scalar_energy_cost"],
        g["scalar_joints_at_limit_cost"], g["scalar_termination_height"], g["scalar_death_cost"],
        g["scalar_max_episode_length"], g["gears"], float(g["gears"].max()))
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_allclose(rew, g["rew"], rtol=1e-5, atol=2e-5)


def test_cartpole_reward_matches_reference(golden_dir):
    g = _load(golden_dir, "cartpole_reward.npz")
    rew, reset = T.compute_cartpole_reward(g["pole_angle"], g["pole_vel"], g["cart_vel"], g["cart_pos"],
                                           g["scalar_reset_dist"], g["reset_in"], g["progress"],
                                           g["scalar_max_episode_length"])
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_allclose(rew, g["rew"], rtol=1e-6, atol=1e-6)


def test_reset_rng_is_uniform_and_deterministic():
    u = T.mi_uniform(1234, np.arange(4096, dtype=np.uint32)[:, None], np.uint32(3), np.arange(16, dtype=np.uint32)[None, :])
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    u2 = T.mi_uniform(1234, np.arange(4096, dtype=np.uint32)[:, None], np.uint32(3), np.arange(16, dtype=np.uint32)[None, :])
    np.testing.assert_array_equal(u, u2)
    # sharding invariance: env 100 of a 2-rank job == env 100 of a 1-rank job
    a = T.mi_uniform(7, np.uint32(64 + 36), np.uint32(0), np.arange(8, dtype=np.uint32))
    b = T.mi_uniform(7, np.uint32(100), np.uint32(0), np.arange(8, dtype=np.uint32))
    np.testing.assert_array_equal(a, b)

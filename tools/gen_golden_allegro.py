#!/usr/bin/env python
"""Golden vectors of the AllegroHand task's observation assembly (reference isaacgymenvs/tasks/allegro_hand.py:441-507): the METHOD BODIES
compute_full_observations (no_vel False / True) and compute_full_state run on a mock `self` with random inputs -> tests/golden/allegro_hand.npz.
(compute_hand_reward / randomize_rotation_pen of the same file: tools/gen_golden_jit_twins.py.)  Needs /root/reference; run in the development
container, commit the file."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "gen_golden.py"))
gg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gg)


def main():
    import importlib
    gg.import_reference()
    mod = importlib.import_module("isaacgymenvs.tasks.allegro_hand")
    n, nd, na = 256, 16, 16
    g = torch.Generator().manual_seed(11)
    m = types.SimpleNamespace()
    m.num_envs, m.num_shadow_hand_dofs, m.num_actions = n, nd, na
    lo = -torch.rand(nd, generator=g) - 0.1
    up = torch.rand(nd, generator=g) + 0.1
    m.shadow_hand_dof_lower_limits, m.shadow_hand_dof_upper_limits = lo, up
    m.shadow_hand_dof_pos = lo + torch.rand(n, nd, generator=g) * (up - lo)
    m.shadow_hand_dof_vel = torch.randn(n, nd, generator=g) * 3
    m.dof_force_tensor = torch.randn(n, nd, generator=g)
    m.vel_obs_scale, m.force_torque_obs_scale = 0.2, 10.0
    obj_state = torch.cat([torch.randn(n, 3, generator=g) * 0.1, gg.rand_quat(g, n), torch.randn(n, 6, generator=g)], dim=-1)
    m.object_pose, m.object_linvel, m.object_angvel, m.object_rot = obj_state[:, 0:7], obj_state[:, 7:10], obj_state[:, 10:13], obj_state[:, 3:7]
    m.goal_pose = torch.cat([torch.randn(n, 3, generator=g) * 0.1, gg.rand_quat(g, n)], dim=-1)
    m.goal_rot = m.goal_pose[:, 3:7]
    m.actions = torch.rand(n, na, generator=g) * 2 - 1
    out = dict(dof_lower=lo, dof_upper=up, dof_pos=m.shadow_hand_dof_pos, dof_vel=m.shadow_hand_dof_vel, dof_force=m.dof_force_tensor,
               object_state=obj_state, goal_pose=m.goal_pose, actions=m.actions)
    m.obs_buf = torch.zeros(n, 88)
    mod.AllegroHand.compute_full_state(m)
    out["full_state"] = m.obs_buf.clone()
    m.states_buf = torch.zeros(n, 88)
    mod.AllegroHand.compute_full_state(m, True)
    out["states"] = m.states_buf.clone()
    m.obs_buf = torch.zeros(n, 72)
    mod.AllegroHand.compute_full_observations(m)
    out["full"] = m.obs_buf.clone()
    m.obs_buf = torch.zeros(n, 50)
    mod.AllegroHand.compute_full_observations(m, True)
    out["full_no_vel"] = m.obs_buf.clone()
    out = {k: v.numpy() for k, v in out.items()}
    np.savez_compressed(os.path.join(gg.OUT, "allegro_hand.npz"), **out)
    print("allegro_hand", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

"""Soak: every task under a full-range uniform random policy for many steps -- everything finite, no speed beyond what the mechanics
allow.  (The Ingenuity's locked rotor joints once pumped a yaw oscillation that only showed after ~230 steps of full-range thrusts and
never in the near-hover parity test: tools/soak.py found it, this keeps it found.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# task, envs, steps, largest admissible |root velocity / joint speed|
CASES = [("Cartpole", 512, 800, 100.0), ("Ant", 1024, 800, 150.0), ("Humanoid", 1024, 800, 150.0), ("Anymal", 1024, 800, 150.0),
         ("AnymalTerrain", 1024, 600, 150.0), ("ShadowHand", 2048, 600, 60.0), ("Quadcopter", 1024, 800, 60.0), ("Ingenuity", 1024, 1200, 60.0),
         ("BallBalance", 1024, 800, 80.0)]


@pytest.mark.parametrize("task,n,steps,vmax", CASES)
def test_random_policy_soak(task, n, steps, vmax):
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=123, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    g = torch.Generator(device=DEV).manual_seed(7)
    t = env.engine.tensors
    worst = 0.0
    for i in range(steps):
        obs, rew, reset, _ = env.step(torch.rand((n, env.num_actions), device=DEV, generator=g) * 2 - 1)
        if i % 50 == 49:
            assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all(), (task, i)
            assert torch.isfinite(t["root_states"]).all() and torch.isfinite(t["dof_state"]).all(), (task, i)
            worst = max(worst, float(t["root_states"][:, 7:13].abs().max()), float(t["dof_state"][..., 1].abs().max()))
            assert worst < vmax, (task, i, worst)

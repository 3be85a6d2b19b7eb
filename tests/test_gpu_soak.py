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


def test_shadow_hand_soak_at_the_corners_of_its_actor_params_ranges():
    """ShadowHand with every env at a corner of the `actor_params` ranges of cfg/task/ShadowHand.yaml (light hand + stiff drives + weak
    damping, heavy object on a light hand, ...) and joint limits shifted by three standard deviations, under full-range random actions:
    the implicit drives and the tendon rows stay stable, nothing blows up."""
    import isaacgymenvs_amd
    n, steps = 2048, 500
    env = isaacgymenvs_amd.make(seed=321, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    t = env.engine.tensors
    g = torch.Generator(device=DEV).manual_seed(11)
    ranges = torch.tensor([[0.5, 1.5], [0.3, 3.0], [0.75, 1.5], [0.75, 1.5], [0.3, 3.0], [0.5, 1.5], [0.95, 1.05]], device=DEV)
    pick = torch.randint(0, 2, (n, 7), device=DEV, generator=g)
    t["actor_scale"][:, :7] = torch.where(pick == 0, ranges[:, 0], ranges[:, 1])
    t["dof_limit_shift"][:] = 0.03 * (torch.randint(0, 2, (n, 48), device=DEV, generator=g) * 2 - 1).float()
    worst = 0.0
    for i in range(steps):
        obs, rew, reset, _ = env.step(torch.rand((n, 20), device=DEV, generator=g) * 2 - 1)
        if i % 50 == 49:
            assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all(), i
            assert torch.isfinite(t["dof_state"]).all() and torch.isfinite(t["object_state"]).all(), i
            worst = max(worst, float(t["dof_state"][..., 1].abs().max()))
            assert worst < 80.0, (i, worst)
            lo = env.shadow_hand_dof_lower_limits - 0.03; up = env.shadow_hand_dof_upper_limits + 0.03
            viol = torch.maximum(lo - env.shadow_hand_dof_pos, env.shadow_hand_dof_pos - up).max()
            assert float(viol) < 0.35, (i, float(viol))
    assert int(t["object_contact_count"].sum()) > 0


@pytest.mark.parametrize("mw", [0, 16])
def test_periodic_full_amplitude_policy_soak(mw):
    """A crude periodic gait at full amplitude flings some Ants into the air spinning at > 100 rad/s; before the simulator's velocity
    clamp (AssetOptions.max_angular_velocity = 64 rad/s by default) was in the engine, the single-wave kernel ended with a NaN state
    for one env in 4096 after ~660 steps of this (tools/debug/mw_policy_check.py)."""
    import isaacgymenvs_amd
    n, steps = 4096, 800
    env = isaacgymenvs_amd.make(seed=3, task="Ant", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    env.engine.set_option("multi_wave", mw)
    g = torch.Generator(device=DEV).manual_seed(1)
    phase = torch.rand((n, 8), device=DEV, generator=g) * 6.283
    freq = 0.15 + 0.1 * torch.rand((n, 1), device=DEV, generator=g)
    t = env.engine.tensors
    for k in range(steps):
        obs, rew, reset, _ = env.step(torch.sin(freq * k + phase))
        if k % 40 == 39:
            assert torch.isfinite(rew).all() and torch.isfinite(t["root_states"]).all() and torch.isfinite(t["dof_state"]).all(), (mw, k)
            assert float(t["root_states"][:, 10:13].norm(dim=1).max()) <= 64.0 * (1 + 1e-3), (mw, k)


PERIODIC = [("Cartpole", 1024, 600), ("Humanoid", 4096, 600), ("Anymal", 2048, 600), ("AnymalTerrain", 2048, 600), ("ShadowHand", 4096, 500),
            ("Quadcopter", 2048, 600), ("Ingenuity", 2048, 800), ("BallBalance", 2048, 600)]


@pytest.mark.parametrize("task,n,steps", PERIODIC)
def test_periodic_policy_soak_of_the_other_tasks(task, n, steps):
    """The periodic full-amplitude policy (per-env random phases and frequencies) that found the missing velocity clamp on the Ant, on
    every other task: everything stays finite, the root's angular speed stays inside the simulator's clamp (tools/debug/periodic_policy_soak.py)."""
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    g = torch.Generator(device=DEV).manual_seed(1)
    phase = torch.rand((n, env.num_actions), device=DEV, generator=g) * 6.283
    freq = 0.1 + 0.3 * torch.rand((n, 1), device=DEV, generator=g)
    t = env.engine.tensors
    for k in range(steps):
        obs, rew, reset, _ = env.step(torch.sin(freq * k + phase))
        if k % 40 == 39:
            assert torch.isfinite(rew).all() and torch.isfinite(obs["obs"]).all(), (task, k)
            assert torch.isfinite(t["root_states"]).all() and torch.isfinite(t["dof_state"]).all(), (task, k)
            assert float(t["root_states"][:, 10:13].norm(dim=1).max()) <= 64.0 * (1 + 1e-3), (task, k)
            assert float(t["dof_state"][..., 1].abs().max()) < 400.0, (task, k)

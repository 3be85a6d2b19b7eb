"""A coarse speed check of every task's VecTask.step(): the other GPU tests compare results, so a kernel that computes the right thing
twenty times too slowly passes them all -- which is how the AllegroHand sub-step ran at 5.3 ms per launch for a while (constexpr table
look-ups left as run-time loops, profiles/r3z_allegro_hand_fix.txt).  Ceilings are 5-10 x the measured step time on a fast box (the pool's
slow boxes are 1.2-1.5 x slower): they catch pathologies, not regressions of a few per cent -- those are the A/B files' business."""
import time

import pytest
import torch

DEV = "cuda:0"

# task, num_envs, ceiling in ms per step (measured, round 3: see the comment on each line)
CASES = [("Cartpole", 4096, 0.15),        # 0.016
         ("Ant", 4096, 0.3),              # 0.040
         ("Humanoid", 8192, 1.2),         # 0.180
         ("AnymalTerrain", 4096, 0.9),    # 0.123
         ("ShadowHand", 16384, 1.6),      # 0.207
         ("Anymal", 4096, 0.4),           # 0.044 (leg waves, both sub-steps in one launch; one wave: 0.082)
         ("Quadcopter", 8192, 0.35),      # 0.042
         ("Ingenuity", 4096, 0.25),       # 0.021
         ("BallBalance", 4096, 0.3),      # 0.031
         ("AllegroHand", 4096, 2.0)]      # 0.26 (four finger-per-wave sub-step launches per control step)


@pytest.mark.gpu
@pytest.mark.parametrize("task,n,ceiling_ms", CASES)
def test_step_time_is_in_the_expected_order_of_magnitude(task, n, ceiling_ms):
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=42, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    g = torch.Generator(device=DEV).manual_seed(1)
    acts = [torch.rand((n, env.num_actions), device=DEV, generator=g) * 2 - 1 for _ in range(4)]
    for i in range(40):
        env.step(acts[i % 4])
    torch.cuda.synchronize()
    k = 100
    t0 = time.perf_counter()
    for i in range(k):
        env.step(acts[i % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    assert ms < ceiling_ms, f"{task}@{n}: {ms:.3f} ms per step (ceiling {ceiling_ms})"

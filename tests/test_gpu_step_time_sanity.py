"""A coarse speed check of every task's VecTask.step(): the other GPU tests compare results, so a kernel that computes the right thing
twenty times too slowly passes them all -- which is how the AllegroHand sub-step ran at 5.3 ms per launch for a while (constexpr table
look-ups left as run-time loops, profiles/r3z_allegro_hand_fix.txt).  Ceilings are 5-10 x the measured step time on a fast box (the pool's
slow boxes are 1.2-1.5 x slower): they catch pathologies, not regressions of a few per cent -- those are the A/B files' business."""
import os
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


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["Cartpole", "Ant", "Humanoid", "AnymalTerrain", "ShadowHand", "Anymal", "AllegroHand", "Quadcopter", "Ingenuity", "BallBalance"])
def test_init_state_writes_everything_the_step_reads(task, monkeypatch):
    """The arena is caller-owned memory (include/mi_engine.h); the Python layer hands it over zeroed, a C client need not.  With the arena full
    of NaN bit patterns before mi_engine_create / mi_engine_init_state (MI_ARENA_POISON) a rollout must give the same observations, rewards and
    resets, bit for bit, as from a zeroed arena: nothing the step reads may be left to what the memory held before."""
    import isaacgymenvs_amd
    n = 256
    ref = isaacgymenvs_amd.make(seed=7, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    monkeypatch.setenv("MI_ARENA_POISON", "1")
    env = isaacgymenvs_amd.make(seed=7, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    monkeypatch.delenv("MI_ARENA_POISON")
    # (the poison was there: the rigid-body state tensor is only ever written on demand)
    assert torch.isnan(env.engine.tensors["rigid_body_state"]).all() and not torch.isnan(ref.engine.tensors["rigid_body_state"]).any()
    g = torch.Generator(device=DEV).manual_seed(2)
    for step in range(12):
        a = torch.rand((n, env.num_actions), device=DEV, generator=g) * 2 - 1
        o1, r1, d1, _ = env.step(a)
        o2, r2, d2, _ = ref.step(a)
        assert torch.isfinite(o1["obs"]).all() and torch.isfinite(r1).all(), (task, step)
        assert torch.equal(o1["obs"], o2["obs"]) and torch.equal(r1, r2) and torch.equal(d1, d2), (task, step)


@pytest.mark.gpu
def test_bench_line_is_the_last_stdout_line_when_a_process_group_is_up():
    """`bench.py --gpus N` under torchrun prints ONE JSON line on rank 0 (the driver parses it).  RCCL prints a version banner to the C-level
    stdout when its first communicator comes up; through a pipe that banner used to land AFTER the JSON line.  Exercised here with a one-rank
    process group over RCCL (MI_FORCE_DIST=1: the same code path as N > 1 -- barriers, MAX all-reduce of the region time, the settle loop's
    broadcast, the statistics reducers)."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, MI_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    d = json.loads(lines[-1])                       # the LAST line is the JSON line
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1

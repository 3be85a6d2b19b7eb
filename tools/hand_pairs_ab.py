#!/usr/bin/env python3
"""What the Shadow Hand's hand-to-hand contact pairs cost (round 5): ShadowHand at the benchmark size, random-action rollout, the SAME engine
stepped in alternating blocks with option hand_pair_stiffness = 0 (pairs off: no pair capsules exchanged, no barrier B0) and 2e4 (the default).
Prints ms per control step per block, the pair sides pushed per env-sub-step, and the contact refusal rate.  Usage: tools/hand_pairs_ab.py [envs] [steps]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import isaacgymenvs_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
g = torch.Generator(device="cuda:0").manual_seed(0)
acts = [2.0 * torch.rand((n, 20), device="cuda:0", generator=g) - 1.0 for _ in range(8)]
for i in range(200):
    env.step(acts[i % 8])
torch.cuda.synchronize()
print(f"ShadowHand@{n} multi_wave={int(env.engine.get_option('multi_wave'))}, {steps} steps per block")
for rep in range(3):
    for k in (0.0, 2.0e4):
        env.engine.set_option("hand_pair_stiffness", k)
        for i in range(50):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        sides, t0 = 0, time.perf_counter()
        for i in range(steps):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pc = env.engine.tensors["hand_pair_count"]
        print(f"rep {rep} pair_k {k:8.0f}: {1e3 * dt / steps:.4f} ms/step  {n * steps / dt / 1e6:.2f} M env-steps/s   pair sides per env (last sub-step) {float(pc.sum()) / n:.3f}, "
              f"envs with a pushed pair {float((pc.sum(1) > 0).float().mean()):.3f}")

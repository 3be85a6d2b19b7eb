#!/usr/bin/env python3
"""Contact-slot use at the benchmark sizes under the benchmark's random policy (the acceptance bounds of tests/test_gpu_fullsize.py should sit
close to measured rates): ShadowHand@16384 -- contacts taken / refused for want of a slot, in the finger-per-wave form (per-limb caps, model table limb_kcap)
and the one-wave form (one pool of 12); Humanoid@8192 -- ground contacts refused per env-sub-step (per-role caps wave_kcap), self contacts refused."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isaacgymenvs_amd  # noqa: E402

DEV = "cuda:0"


def hand(mw, n=16384, steps=150):
    env = isaacgymenvs_amd.make(seed=42, task="ShadowHand", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    env.engine.set_option("multi_wave", mw)
    g = torch.Generator(device=DEV).manual_seed(0)
    taken = 0
    hist = torch.zeros(32, dtype=torch.long, device=DEV)
    for _ in range(steps):
        env.step(torch.rand((n, 20), device=DEV, generator=g) * 2 - 1)
        c = env.engine.tensors["object_contact_count"]
        taken += int(c.sum())
        hist += torch.bincount(c.clamp(max=31), minlength=32)
    dropped = int(env.engine.tensors["object_contact_dropped"].sum())
    envs_dropping = int((env.engine.tensors["object_contact_dropped"] > 0).sum())
    print(f"ShadowHand@{n} multi_wave={mw}: contacts taken (last sub-step of each step) {taken}, refused (both sub-steps) {dropped}: "
          f"{dropped / (2.0 * taken):.5f} per taken contact; envs that ever refused one {envs_dropping} of {n}; "
          f"contacts per env histogram {[int(x) for x in hist.tolist()[:24]]}")


def humanoid(n=8192, steps=200):
    env = isaacgymenvs_amd.make(seed=42, task="Humanoid", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    g = torch.Generator(device=DEV).manual_seed(0)
    for _ in range(steps):
        env.step(torch.rand((n, 21), device=DEV, generator=g) * 2 - 1)
    d = env.engine.tensors["contact_dropped"]
    print(f"Humanoid@{n}: ground contacts refused {int(d[:, 0].sum())} = {int(d[:, 0].sum()) / (n * steps * 2.0):.2e} per env-sub-step "
          f"(envs: {int((d[:, 0] > 0).sum())}); self contacts refused {int(d[:, 1].sum())} = {int(d[:, 1].sum()) / (n * steps * 2.0):.2e} per env-sub-step")


if __name__ == "__main__":
    hand(64)
    hand(0)
    humanoid()

#!/usr/bin/env python3
"""What the `isaacgym` stand-in costs: the reference's UNMODIFIED ant.py / humanoid.py (tasks/ant.py:281-297: set_dof_actuation_force_tensor ->
simulate x controlFrequencyInv -> refresh_* -> jitted compute_*_observations / compute_*_reward as torch ops) stepping on the HIP engine
through isaacgymenvs_amd/shims, against the native task class (one fused launch group per step) on the same box.  The stand-in's
acquire_* tensors are AoS copies of the engine's SoA arena (a transpose per refresh / set call) and its step is the reference's own
Python: this measures both together.  Needs the reference tree (MI_REFERENCE_ROOT, /root/reference or ab/ref_stage).
Usage: tools/shim_overhead.py [num_envs]"""
import importlib
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("MI_REFERENCE_ROOT") or next((p for p in ("/root/reference", os.path.join(ROOT, "ab", "ref_stage"))
                                                    if os.path.isdir(os.path.join(p, "isaacgymenvs", "tasks"))), None)
if REF is None:
    sys.exit("reference tree not reachable")
DEV = "cuda:0"


def timed(env, n, na, k):
    g = torch.Generator(device=DEV).manual_seed(1)
    acts = [torch.rand((n, na), device=DEV, generator=g) * 2 - 1 for _ in range(8)]
    for i in range(100):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(k):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / k)
    return best


def main():
    import isaacgymenvs_amd
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    n_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    rows = []
    for task, mod, cls, na, n_def in (("Ant", "ant", "Ant", 8, 4096), ("Humanoid", "humanoid", "Humanoid", 21, 8192)):
        n = n_arg or n_def
        native = isaacgymenvs_amd.make(seed=42, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
        t_native = timed(native, n, na, 500)
        del native
        rows.append((task, n, "native task class (fused kernels)", t_native))
    import isaacgymenvs_amd.shims as shims
    for k in [k for k in sys.modules if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")]:
        del sys.modules[k]
    shims.install(force=True)
    for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"), ("isaacgymenvs.utils", "isaacgymenvs/utils"),
                      ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = m
    vt = importlib.import_module("isaacgymenvs.tasks.base.vec_task")
    for task, mod, cls, na, n_def in (("Ant", "ant", "Ant", 8, 4096), ("Humanoid", "humanoid", "Humanoid", 21, 8192)):
        n = n_arg or n_def
        cfg = omegaconf_to_dict(compose("config", overrides=[f"task={task}"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
        cfg["env"]["numEnvs"] = n
        cfg["sim"]["use_gpu_pipeline"] = True
        vt.EXISTING_SIM = None
        env = getattr(importlib.import_module("isaacgymenvs.tasks." + mod), cls)(cfg, rl_device=DEV, sim_device=DEV, graphics_device_id=-1, headless=True,
                                                                                  virtual_screen_capture=False, force_render=False)
        rows.append((task, n, "reference task file through the isaacgym stand-in", timed(env, n, na, 200)))
        del env
    for task, n, what, t in rows:
        print(f"{task}@{n}: {what}: {t * 1e3:.4f} ms/step, {n / t / 1e6:.2f} M env-steps/s", flush=True)


if __name__ == "__main__":
    main()

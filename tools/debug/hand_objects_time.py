import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import isaacgymenvs_amd
from isaacgymenvs_amd.utils.config import compose
for task, n, na in (("ShadowHand", 16384, 20), ("AllegroHand", 16384, 16)):
    for obj in ("egg", "pen"):
        cfg = compose(overrides=[f"task={task}"]); cfg["task"]["env"]["numEnvs"] = n; cfg["task"]["env"]["objectType"] = obj
        env = isaacgymenvs_amd.make(seed=1, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True, cfg=cfg)
        g = torch.Generator(device="cuda:0").manual_seed(1)
        acts = [torch.rand((n, na), device="cuda:0", generator=g) * 2 - 1 for _ in range(4)]
        for i in range(60): env.step(acts[i % 4])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200): env.step(acts[i % 4])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"{task}@{n} {obj}: {dt*1e3:.4f} ms/step, {n/dt/1e6:.1f} M env-steps/s, multi_wave {int(env.engine.get_option('multi_wave'))}", flush=True)
        del env

import time, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import isaacgymenvs_amd
for task, n in (("Cartpole", 64), ("Ant", 64), ("Ant", 4096)):
    env = isaacgymenvs_amd.make(seed=1, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    a = torch.zeros((n, env.num_actions), device="cuda:0")
    for _ in range(200): env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): env.step(a)
    t1 = time.perf_counter()           # launch (CPU) time only
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t3 = time.perf_counter()
    for _ in range(3000): env.engine.step(a)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{task}@{n}: env.step CPU {1e6*(t1-t0)/3000:.1f} us/step (drain {1e3*(t2-t1):.1f} ms), engine.step CPU {1e6*(t4-t3)/3000:.1f} us/step")

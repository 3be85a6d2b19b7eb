"""Which outputs of a Humanoid simulate() depend on what the LDS held before (tests/test_gpu_fullsize.py poisons it with NaNs)."""
import sys; import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import ctypes as C, numpy as np, torch
from isaacgymenvs_amd import native
from isaacgymenvs_amd.registry import load_model
import isaacgymenvs_amd
from test_gpu_parity import _random_state, _t
for task, n, zl, zh, gear, mw in (("Humanoid", 8192, 0.9, 1.4, 60.0, 32), ("Humanoid", 8192, 0.9, 1.4, 60.0, 2), ("Humanoid", 8192, 0.9, 1.4, 60.0, 0)):
    env = isaacgymenvs_amd.make(seed=5, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    env.engine.set_option("multi_wave", mw)
    L = native.lib(); L.mi_debug_poison_lds.argtypes = [C.c_uint, C.c_void_p]
    L.mi_debug_poison_lds(0x7FC00000, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    spec = load_model(task.lower()); rng = np.random.default_rng(7)
    root, q, qd = _random_state(spec, n, rng, zl, zh); tau = rng.uniform(-gear, gear, (n, spec.nd))
    t = env.engine.tensors
    t["root_states"][:] = _t(root); env.dof_pos[:] = _t(q); env.dof_vel[:] = _t(qd)
    for k in ("contact_impulse", "limit_impulse", "self_contact_impulse"):
        if k in t: t[k].zero_()
    t["dof_actuation_force"][:] = _t(tau)
    for it in range(2):
        env.engine.simulate(); torch.cuda.synchronize()
        vs = env.vec_sensor_tensor.cpu().numpy()
        bad = ~np.isfinite(vs)
        print(task, "mw", mw, "it", it, "vec_sensor_tensor", vs.shape, "non-finite envs", int(bad.any(1).sum()), "columns", np.nonzero(bad.any(0))[0], flush=True)
        for k in t:
            a = t[k]
            if a.dtype == torch.float32:
                a = a.cpu().numpy().reshape(-1)
                if not np.isfinite(a).all(): print("   tensor", k, "non-finite count", int((~np.isfinite(a)).sum()), flush=True)

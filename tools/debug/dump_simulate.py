"""GPU debug: run mi_engine_simulate from the random states of tests/test_gpu_parity.py and dump results."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import isaacgymenvs_amd
from isaacgymenvs_amd.registry import load_model
task = sys.argv[1] if len(sys.argv) > 1 else "Ant"
n = 256
env = isaacgymenvs_amd.make(seed=5, task=task, num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
spec = load_model(task.lower())
rng = np.random.default_rng(0)
nd = spec.nd
lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
z_lo, z_hi, gear = (0.3, 0.6, 15.0) if task == "Ant" else (0.9, 1.4, 60.0)
root = np.zeros((n, 13)); root[:, 0:2] = rng.normal(size=(n, 2)); root[:, 2] = rng.uniform(z_lo, z_hi, n)
q = rng.normal(size=(n, 4)); q[:, 3] += 3; q /= np.linalg.norm(q, axis=1, keepdims=True); root[:, 3:7] = q
root[:, 7:13] = rng.normal(size=(n, 6))
qq = rng.uniform(lo, up, (n, nd)); qd = rng.normal(size=(n, nd)) * 2
tau = rng.uniform(-gear, gear, (n, nd))
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
t = env.engine.tensors
t["root_states"][:] = T(root); env.dof_pos[:] = T(qq); env.dof_vel[:] = T(qd)
t["contact_impulse"].zero_(); t["limit_impulse"].zero_(); t["dof_actuation_force"][:] = T(tau)
torch.cuda.synchronize()
pre = dict(root0=t["root_states"].cpu().numpy(), q0=env.dof_pos.cpu().numpy(), qd0=env.dof_vel.cpu().numpy(), tau0=t["dof_actuation_force"].cpu().numpy())
env.engine.simulate()
torch.cuda.synchronize()
np.savez(os.path.join(ROOT, "gpurun_out", f"dump_{task}.npz"), root=t["root_states"].cpu().numpy(), q=env.dof_pos.cpu().numpy(),
         qd=env.dof_vel.cpu().numpy(), lamc=t["contact_impulse"].cpu().numpy(), laml=t["limit_impulse"].cpu().numpy(),
         sensor=t["force_sensor"].cpu().numpy(), dof_force=t["dof_force"].cpu().numpy(), **pre)
print("dumped", task)

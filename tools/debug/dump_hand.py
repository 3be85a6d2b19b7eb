import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import isaacgymenvs_amd
np.set_printoptions(precision=4, suppress=True, linewidth=220)
n = 64
env = isaacgymenvs_amd.make(seed=13, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
g = torch.Generator(device="cpu").manual_seed(7)
for step in range(4):
    a = torch.rand((n, 20), generator=g) * 2 - 1
    env.step(a.to("cuda:0"))
    torch.cuda.synchronize()
    q = env.shadow_hand_dof_pos.cpu().numpy(); qd = env.shadow_hand_dof_vel.cpu().numpy()
    print("step", step, "q0", q[0], "\n  qd0", qd[0][:8], "\n  obj0", env.object_state[0].cpu().numpy(), "ncon", env.engine.tensors["object_contact_count"][:8].cpu().numpy(),
          "\n  tgt0", env.cur_targets[0].cpu().numpy()[:8], "nan envs", int(torch.isnan(env.obs_buf).any(dim=1).sum()), "rew nan", int(torch.isnan(env.rew_buf).sum()),
          "\n  laml0", env.engine.tensors["limit_impulse"][0].cpu().numpy()[:8], "dof_force0", env.dof_force_tensor[0].cpu().numpy()[:6])

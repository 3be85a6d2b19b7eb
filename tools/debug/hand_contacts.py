import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import isaacgymenvs_amd
n = 4096
env = isaacgymenvs_amd.make(seed=3, task="ShadowHand", num_envs=n, sim_device="cuda:0", rl_device="cuda:0", headless=True)
nc = env.engine.tensors["object_contact_count"]
hist = torch.zeros(20, device="cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(1)
for mode in ("random", "zero"):
    hist.zero_()
    for step in range(400):
        a = (torch.rand((n, 20), device="cuda:0", generator=g) * 2 - 1) if mode == "random" else torch.zeros((n, 20), device="cuda:0")
        env.step(a)
        if step >= 50:
            hist += torch.bincount(nc.clamp(0, 19).long(), minlength=20).float()
    h = (hist / hist.sum()).cpu().tolist()
    cum = 0; out = []
    for k, p in enumerate(h):
        cum += p; out.append(f"{k}:{p:.3f}")
    print(mode, " ".join(out[:17]), "P(>8)=%.4f P(>12)=%.4f P(>=16)=%.4f" % (sum(h[9:]), sum(h[13:]), sum(h[16:])))

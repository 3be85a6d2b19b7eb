#!/usr/bin/env python3
"""Minimal PPO on top of the VecTask API -- an end-to-end plausibility check of the engine, not part of the product path.

The reference trains through rl_games (`python train.py task=Ant`, isaacgymenvs/train.py + cfg/train/AntPPO.yaml); rl_games
is not installable here, so this file restates the few pieces of that recipe that matter for a sanity run: horizon 16,
gamma 0.99, lambda 0.95, 4 mini-epochs, minibatch 32768, clip 0.2, lr 3e-4 with the adaptive-KL schedule (kl_threshold
0.008), reward scale 0.01, running-mean observation / value normalisation, MLP 256-128-64 ELU, fixed-sigma init 0, time-out
bootstrapping from extras["time_outs"] (cfg/train/AntPPO.yaml:1-70).

    python examples/train_ppo.py --task Ant --num-envs 4096 --iters 300

A physics engine that is wrong in a way that matters (no traction, energy leak, exploding contacts) shows up here as a
policy that does not learn to move forward; `mean_episode_return` is the same quantity the reference's observer logs."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn as nn


class RunningMeanStd:
    def __init__(self, shape, device):
        self.mean = torch.zeros(shape, device=device, dtype=torch.float64)
        self.var = torch.ones(shape, device=device, dtype=torch.float64)
        self.count = 1e-4

    def update(self, x):
        x = x.to(torch.float64).reshape(-1, *self.mean.shape) if self.mean.ndim else x.to(torch.float64).reshape(-1)
        bm, bv, bc = x.mean(0), x.var(0, unbiased=False), x.shape[0]
        d = bm - self.mean
        tot = self.count + bc
        self.mean = self.mean + d * bc / tot
        self.var = (self.var * self.count + bv * bc + d * d * self.count * bc / tot) / tot
        self.count = tot

    def norm(self, x, clip=5.0):
        return torch.clamp((x - self.mean.float()) / torch.sqrt(self.var.float() + 1e-5), -clip, clip)

    def denorm(self, x):
        return x * torch.sqrt(self.var.float() + 1e-5) + self.mean.float()


class ActorCritic(nn.Module):
    def __init__(self, nobs, nact, units=(256, 128, 64)):
        super().__init__()
        layers, d = [], nobs
        for u in units:
            layers += [nn.Linear(d, u), nn.ELU()]
            d = u
        self.trunk = nn.Sequential(*layers)
        self.mu = nn.Linear(d, nact)
        self.value = nn.Linear(d, 1)
        self.logstd = nn.Parameter(torch.zeros(nact))

    def forward(self, obs):
        h = self.trunk(obs)
        return self.mu(h), self.logstd.expand(obs.shape[0], -1), self.value(h).squeeze(-1)


def logp(mu, logstd, a):
    return (-0.5 * ((a - mu) / logstd.exp()) ** 2 - logstd - 0.9189385332046727).sum(-1)


def reference_env(spec, task, num_envs, dev):
    """one of the reference's own task classes, its file imported unmodified, on this engine through the `isaacgym` stand-in"""
    import importlib
    import types
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.environ.get("MI_REFERENCE_ROOT") or next((p for p in ("/root/reference", os.path.join(root, "ab", "ref_stage"))
                                                        if os.path.isdir(os.path.join(p, "isaacgymenvs", "tasks"))), "/root/reference")
    shims.install(force=True)
    for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"), ("isaacgymenvs.utils", "isaacgymenvs/utils"),
                      ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(ref, rel)]
        sys.modules[name] = mod
    module, cls = spec.split(":")
    m = importlib.import_module("isaacgymenvs.tasks." + module)
    cfg = omegaconf_to_dict(compose("config", overrides=[f"task={task}"], cfg_dir=os.path.join(ref, "isaacgymenvs", "cfg"))["task"])
    cfg["env"]["numEnvs"], cfg["sim"]["use_gpu_pipeline"] = num_envs, True
    return getattr(m, cls)(cfg, rl_device=dev, sim_device=dev, graphics_device_id=-1, headless=True, virtual_screen_capture=False, force_render=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Ant")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--minibatch", type=int, default=32768)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--reward-scale", type=float, default=0.01)
    ap.add_argument("--log", default="")
    ap.add_argument("--units", default="256,128,64", help="hidden layer sizes of the MLP (HumanoidPPO.yaml: 400,200,100)")
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--reference-task", default="", help="module:Class of one of the REFERENCE's own task files, run unmodified through the isaacgym stand-in "
                    "(e.g. franka_cube_stack:FrankaCubeStack; the reference tree at MI_REFERENCE_ROOT, /root/reference or ab/ref_stage)")
    args = ap.parse_args()

    import isaacgymenvs_amd
    dev = "cuda:0"
    torch.manual_seed(args.seed)
    if args.reference_task:
        env = reference_env(args.reference_task, args.task, args.num_envs, dev)
    else:
        env = isaacgymenvs_amd.make(seed=args.seed, task=args.task, num_envs=args.num_envs, sim_device=dev, rl_device=dev, headless=True)
    N, T, A, O = args.num_envs, args.horizon, env.num_actions, env.num_obs
    net = ActorCritic(O, A, tuple(int(u) for u in args.units.split(","))).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, eps=1e-8)
    lr, gamma, lam, clip = args.lr, 0.99, 0.95, 0.2
    obs_rms, val_rms = RunningMeanStd((O,), dev), RunningMeanStd((), dev)

    obs = env.reset()["obs"].clone()
    ep_ret = torch.zeros(N, device=dev)
    ep_len = torch.zeros(N, device=dev)
    hist = []
    t0 = time.time()
    for it in range(args.iters):
        B = dict(obs=torch.zeros(T, N, O, device=dev), act=torch.zeros(T, N, A, device=dev), logp=torch.zeros(T, N, device=dev),
                 val=torch.zeros(T, N, device=dev), rew=torch.zeros(T, N, device=dev), done=torch.zeros(T, N, device=dev),
                 mu=torch.zeros(T, N, A, device=dev))
        fin_ret, fin_len, fin_n = 0.0, 0.0, 0
        with torch.no_grad():
            for t in range(T):
                obs_rms.update(obs)
                on = obs_rms.norm(obs)
                mu, ls, v = net(on)
                a = mu + ls.exp() * torch.randn_like(mu)
                B["obs"][t], B["act"][t], B["logp"][t], B["mu"][t] = on, a, logp(mu, ls, a), mu
                B["val"][t] = val_rms.denorm(v)
                od, rew, done, extras = env.step(torch.clamp(a, -1.0, 1.0))
                nobs, rew = od["obs"].clone(), rew.detach()
                r = rew * args.reward_scale
                # time-out bootstrapping (rl_games value_bootstrap): add gamma * V(s_T) for envs cut by the time limit
                to = extras["time_outs"].float()
                if to.any():
                    _, _, vt = net(obs_rms.norm(nobs))
                    r = r + gamma * val_rms.denorm(vt) * to
                B["rew"][t], B["done"][t] = r, done.float()
                ep_ret += rew
                ep_len += 1
                d = done.bool()
                if d.any():
                    fin_ret += float(ep_ret[d].sum()); fin_len += float(ep_len[d].sum()); fin_n += int(d.sum())
                    ep_ret[d] = 0; ep_len[d] = 0
                obs = nobs
            _, _, vl = net(obs_rms.norm(obs))
            last_v = val_rms.denorm(vl)
            adv = torch.zeros(T, N, device=dev)
            g = torch.zeros(N, device=dev)
            for t in reversed(range(T)):
                nv = last_v if t == T - 1 else B["val"][t + 1]
                nd = 1.0 - B["done"][t]
                delta = B["rew"][t] + gamma * nv * nd - B["val"][t]
                g = delta + gamma * lam * nd * g
                adv[t] = g
            ret = adv + B["val"]
            val_rms.update(ret)
        flat = {k: v.reshape(T * N, *v.shape[2:]) for k, v in B.items()}
        fadv, fret = adv.reshape(-1), val_rms.norm(ret.reshape(-1), clip=1e9)
        fadv = (fadv - fadv.mean()) / (fadv.std() + 1e-8)
        kls = []
        for _ in range(args.epochs):
            perm = torch.randperm(T * N, device=dev)
            for s in range(0, T * N, args.minibatch):
                idx = perm[s:s + args.minibatch]
                mu, ls, v = net(flat["obs"][idx])
                lp = logp(mu, ls, flat["act"][idx])
                ratio = (lp - flat["logp"][idx]).exp()
                a_ = fadv[idx]
                pl = torch.max(-a_ * ratio, -a_ * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
                vl_ = ((v - fret[idx]) ** 2).mean()
                bl = (torch.clamp(mu - 1.1, min=0) ** 2 + torch.clamp(-1.1 - mu, min=0) ** 2).sum(-1).mean()   # bounds loss
                loss = pl + 2.0 * vl_ + 1e-4 * bl
                opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(net.parameters(), 1.0)
                opt.step()
                with torch.no_grad():
                    omu = flat["mu"][idx]
                    kl = (0.5 * ((omu - mu) / ls.exp()) ** 2).sum(-1).mean()   # fixed-sigma gaussians (sigma changes slowly)
                    kls.append(float(kl))
            mkl = sum(kls[-max(1, (T * N) // args.minibatch):]) / max(1, (T * N) // args.minibatch)
            if mkl > 2.0 * 0.008:
                lr = max(lr / 1.5, 1e-6)
            elif mkl < 0.5 * 0.008:
                lr = min(lr * 1.5, 1e-2)
            for gparam in opt.param_groups:
                gparam["lr"] = lr
        rec = dict(iter=it, env_steps=(it + 1) * T * N, mean_episode_return=fin_ret / max(fin_n, 1), mean_episode_length=fin_len / max(fin_n, 1),
                   episodes=fin_n, mean_step_reward=float(B["rew"].mean()) / args.reward_scale, lr=lr, kl=mkl, wall_s=time.time() - t0)
        hist.append(rec)
        if it % 10 == 0 or it == args.iters - 1:
            print(json.dumps(rec), flush=True)
    if args.log:
        with open(args.log, "w") as f:
            json.dump(hist, f)


if __name__ == "__main__":
    main()

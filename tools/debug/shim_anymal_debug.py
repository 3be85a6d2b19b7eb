import importlib, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = next(p for p in ("/root/reference", os.path.join(ROOT, "ab", "ref_stage")) if os.path.isdir(os.path.join(p, "isaacgymenvs", "tasks")))
import isaacgymenvs_amd, isaacgymenvs_amd.shims as shims
from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
shims.install(force=True)
for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"), ("isaacgymenvs.utils", "isaacgymenvs/utils"), ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
    mod = types.ModuleType(name); mod.__path__ = [os.path.join(REF, rel)]; sys.modules[name] = mod
DEV = "cuda:0"
n, seed = 128, 7
cfg = omegaconf_to_dict(compose("config", overrides=["task=AnymalTerrain"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
cfg["env"]["numEnvs"] = n; cfg["sim"]["use_gpu_pipeline"] = True
cfg["env"]["terrain"].update(numLevels=3, numTerrains=4, curriculum=True); cfg["env"]["learn"]["addNoise"] = False; cfg["env"]["learn"]["pushRobots"] = False
np.random.seed(seed); torch.manual_seed(seed)
mod = importlib.import_module("isaacgymenvs.tasks.anymal_terrain")
ref = mod.AnymalTerrain(cfg, rl_device=DEV, sim_device=DEV, graphics_device_id=-1, headless=True, virtual_screen_capture=False, force_render=False)
g = torch.Generator().manual_seed(1)
for step in range(40):
    ref.step((torch.rand((n, 12), generator=g) * 2 - 1).to(DEV))
ncfg = compose(overrides=["task=AnymalTerrain"]); ncfg["task"]["env"]["numEnvs"] = n
ncfg["task"]["env"]["terrain"].update(numLevels=3, numTerrains=4, curriculum=True); ncfg["task"]["env"]["learn"]["addNoise"] = False; ncfg["task"]["env"]["learn"]["pushRobots"] = False
nat = isaacgymenvs_amd.make(seed=seed, task="AnymalTerrain", num_envs=n, sim_device=DEV, rl_device=DEV, headless=True, cfg=ncfg)
nat.step(torch.zeros((n, 12), device=DEV))
eng = ref.sim.engine
et, nt = eng.tensors, nat.engine.tensors
print("options", {k: (eng.get_option(k), nat.engine.get_option(k)) for k in ("multi_wave", "control_freq_inv")})
for k in ("root_states", "dof_state", "contact_impulse", "limit_impulse", "friction", "net_contact_force"):
    nt[k].copy_(et[k])
nt["commands"].copy_(ref.commands); nt["last_actions"].copy_(ref.last_actions); nt["last_dof_vel"].copy_(ref.last_dof_vel)
nt["feet_air_time"].copy_(ref.feet_air_time); nt["env_origins"].copy_(ref.env_origins)
nat.progress_buf.copy_(ref.progress_buf); nat.reset_buf.copy_(ref.reset_buf.long())
# --- isolate the physics: same state, same stored efforts, one simulate() on each engine
tau = (torch.rand((n, 12), generator=g) * 40 - 20).to(DEV)
st = {k: et[k].clone() for k in ("root_states", "dof_state", "contact_impulse", "limit_impulse")}
et["dof_actuation_force"].copy_(tau); nt["dof_actuation_force"].copy_(tau)
eng.simulate(); nat.engine.simulate()
print("one simulate: root diff", float((et["root_states"] - nt["root_states"]).abs().max()), "dof diff", float((et["dof_state"] - nt["dof_state"]).abs().max()),
      "netf diff", float((et["net_contact_force"] - nt["net_contact_force"]).abs().max()))
for k, v_ in st.items():
    et[k].copy_(v_); nt[k].copy_(v_)
# --- the PD loop by hand on the native engine, as the reference does it: 4 x (torques from the current dof state, simulate), + 1 simulate
a0 = (torch.rand((n, 12), generator=g) * 2 - 1).to(DEV)
for eng_ in (eng, nat.engine):
    t_ = eng_.tensors
    for i in range(4):
        tq = torch.clip(ref.Kp * (ref.action_scale * a0 + ref.default_dof_pos - t_["dof_state"][..., 0]) - ref.Kd * t_["dof_state"][..., 1], -80., 80.)
        t_["dof_actuation_force"].copy_(tq); eng_.simulate()
    eng_.simulate()
print("manual PD loop on both engines: dof diff", float((et["dof_state"] - nt["dof_state"]).abs().max()))
manual = et["dof_state"].clone()
for k, v_ in st.items():
    et[k].copy_(v_); nt[k].copy_(v_)
nt["last_actions"].copy_(ref.last_actions)
nat.step(a0.clone())
print("native fused step vs manual loop: dof diff", float((manual - nt["dof_state"]).abs().max()), "resets", int(nat.reset_buf.sum()))
for k, v_ in st.items():
    et[k].copy_(v_); nt[k].copy_(v_)
# --- the reference's own pre_physics_step from the same state, sub-step by sub-step
ref.gym.refresh_dof_state_tensor(ref.sim); ref.gym.refresh_actor_root_state_tensor(ref.sim)
print("buffer vs engine before:", float((ref.dof_pos - et["dof_state"][..., 0]).abs().max()), float((ref.dof_vel - et["dof_state"][..., 1]).abs().max()))
ref.actions = a0.clone()
for i in range(4):
    tq_ref = torch.clip(ref.Kp * (ref.action_scale * ref.actions + ref.default_dof_pos - ref.dof_pos) - ref.Kd * ref.dof_vel, -80., 80.)
    tq_man = torch.clip(ref.Kp * (ref.action_scale * a0 + ref.default_dof_pos - nt["dof_state"][..., 0]) - ref.Kd * nt["dof_state"][..., 1], -80., 80.)
    ref.gym.set_dof_actuation_force_tensor(ref.sim, tq_ref); nt["dof_actuation_force"].copy_(tq_man)
    print(i, "torque diff", float((tq_ref - tq_man).abs().max()), "tau tensor diff", float((et["dof_actuation_force"] - nt["dof_actuation_force"]).abs().max()))
    ref.gym.simulate(ref.sim); nat.engine.simulate()
    ref.gym.refresh_dof_state_tensor(ref.sim)
    print(i, "dof diff after simulate", float((et["dof_state"] - nt["dof_state"]).abs().max()), "buffer vs engine", float((ref.dof_pos - et["dof_state"][..., 0]).abs().max()))
for k, v_ in st.items():
    et[k].copy_(v_); nt[k].copy_(v_)
ref.gym.refresh_dof_state_tensor(ref.sim); ref.gym.refresh_actor_root_state_tensor(ref.sim)
a = (torch.rand((n, 12), generator=g) * 2 - 1).to(DEV)
r_obs, r_rew, r_reset, _ = ref.step(a.clone())
n_obs, n_rew, n_reset, _ = nat.step(a.clone())
keep = ~r_reset.bool() & ~n_reset.bool()
print("keep", int(keep.sum()))
print("root diff", float((et["root_states"] - nt["root_states"]).abs()[keep].max()), "dof pos diff", float((et["dof_state"][..., 0] - nt["dof_state"][..., 0]).abs()[keep].max()),
      "dof vel diff", float((et["dof_state"][..., 1] - nt["dof_state"][..., 1]).abs()[keep].max()))
d = (r_obs["obs"] - n_obs["obs"]).abs()[keep]
for lo, hi, nm in ((0, 3, "linvel"), (3, 6, "angvel"), (6, 9, "gravity"), (9, 12, "commands"), (12, 24, "dof pos"), (24, 36, "dof vel"), (36, 176, "heights"), (176, 188, "actions")):
    print(nm, float(d[:, lo:hi].max()), float(d[:, lo:hi].mean()))
print("torques ref vs nat", float((ref.torques - nt["dof_actuation_force"]).abs()[keep].max()), float(ref.torques.abs().max()))
print("rew diff", float((r_rew - n_rew).abs()[keep].max()))

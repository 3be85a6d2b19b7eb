#!/usr/bin/env python3
"""How many ground / self contacts does a Humanoid env carry under random actions?  (CPU oracle, fp32, self-collision on.)  Basis of the
engine's contact-store sizes: 12 ground + 3 self contacts per env.  Output: profiles/r2_contact_counts.txt"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from isaacgymenvs_amd.registry import load_model, sensor_bodies, load_selfcol
from isaacgymenvs_amd.utils.config import compose
from isaacgymenvs_amd.tasks.locomotion import loco_params_from_cfg
from oracle.tasks import OracleLocomotionEnv
cfg = compose(overrides=["task=Humanoid"])["task"]
p = loco_params_from_cfg(cfg, "humanoid", 1.34)
sim = dict(dt=cfg["sim"]["dt"], substeps=2, iters=4, gravity=(0,0,-9.81), contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0, erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)
n=1024
env = OracleLocomotionEnv(True, load_model("humanoid"), sensor_bodies("humanoid"), sim, p, n, seed=1, precision="f32", selfcol=load_selfcol("humanoid"))
rng=np.random.default_rng(0)
hg=np.zeros(40); hp=np.zeros(14); grp=np.zeros(13); ep_len=[]
for it in range(300):
    env.step(rng.uniform(-1,1,(n,21)))
    ng=(env.eng.lam[:, :3*35].reshape(n,35,3)[:,:,0]>0).sum(1)   # loaded ground contacts
    sph_on=(np.abs(env.eng.sph_force).sum(2)>0).sum(1)
    pi=env.eng.pair_info
    np_=(pi[:,:,3]>=0).sum(1)
    if it>=30:
        hg+=np.bincount(np.minimum(sph_on,39),minlength=40); hp+=np.bincount(np_,minlength=14); grp+=(pi[:,:,3]>=0).sum(0)
hg/=hg.sum(); hp/=hp.sum()
print("ground contacts:", " ".join(f"{k}:{v:.4f}" for k,v in enumerate(hg[:20])), "P(>12)=%.5f P(>16)=%.5f"%(hg[13:].sum(), hg[17:].sum()))
print("pair contacts:", " ".join(f"{k}:{v:.4f}" for k,v in enumerate(hp[:8])), "P(>3)=%.5f"%hp[4:].sum())
print("per group share:", np.round(grp/grp.sum(),3))

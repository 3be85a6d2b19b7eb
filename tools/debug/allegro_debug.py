import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_allegro_hand import _make, _oracle
n, seed = 64, 13
env = _make(n, seed)
orc = _oracle(env, n, seed)
g = torch.Generator(device="cpu").manual_seed(7)
a = torch.rand((n, 16), generator=g) * 2 - 1
env.step(a.to("cuda:0")); orc.step(a.numpy())
nc = env.engine.tensors["object_contact_count"].cpu().numpy()
q = env.shadow_hand_dof_pos.cpu().numpy(); qd = env.shadow_hand_dof_vel.cpu().numpy()
ob = env.object_state.cpu().numpy()
print("nc gpu", nc[:24]); print("nc orc", orc.eng.ncontacts[:24])
print("q diff", np.abs(q - orc.eng.q).max(axis=1)[:12])
print("qd diff", np.abs(qd - orc.eng.qd).max(axis=1)[:12])
print("obj diff", np.abs(ob - orc.eng.obj).max(axis=1)[:12])
print("targets diff", np.abs(env.cur_targets.cpu().numpy() - orc.cur_targets).max())
bad = np.nonzero((nc > 0) != (orc.eng.ncontacts > 0))[0]
print("bad", bad)
e = int(bad[0])
print("env", e, "q gpu", q[e].round(4)); print("q orc", orc.eng.q[e].round(4)); print("obj gpu", ob[e].round(4)); print("obj orc", orc.eng.obj[e].round(4))
env.engine.refresh_rigid_body_states()
bs = env.engine.tensors["rigid_body_state"].cpu().numpy()
orc.eng.eng.q[:] = q; orc.eng.eng.qd[:] = qd
bp = orc.eng._poses(e)
print("body pos diff", np.abs(bs[e, :, 0:3] - bp[:, 0:3]).max(axis=1).round(5))

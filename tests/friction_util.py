"""First-principles friction known answers for the locomotion engines, shared by the oracle tests (tests/test_oracle_physics.py), the
CPU-backend tests and the `-m gpu` tests (tests/test_friction.py).

What they pin: a contact's two tangent rows are solved from the SAME velocity, then projected onto the friction disc.  The
order every form used until round 6 -- t1 solved and applied before t2 is looked at -- let a fast-sliding contact's friction point off the
sliding direction (mu_eff 0.468 for mu 0.5 in the scene engine's ramp test).  The known answers, for a robot resting on its
contact spheres (an Ant on its four foot spheres; a Humanoid lying on the end spheres of its capsules):

  * gravity tilted by theta < atan(mu): it stays put;
  * gravity tilted by theta > atan(mu): every contact slides downhill, so the centre of mass accelerates with a = g (sin theta - mu cos theta);
  * on level ground, pushed with v0 along a DIAGONAL of the tangent axes: it stops after v0^2 / (2 mu g), without leaving the line of the push.

A `rig` is anything with set_state(root, q, qd), set_gravity(g3), step(k), root() -> [n, 13] and joints() -> [n, nd] (oracle: OracleRig; engines: EngineRig).
"""
import numpy as np

G = 9.81


class OracleRig:
    def __init__(self, eng, env_mu=None):
        self.e, self.mu = eng, env_mu
        self.tau = np.zeros((eng.N, eng.nd))
        self.pd = None

    def hold(self, q_target, kp, kd):
        """joint PD servo evaluated before every simulate() (the Ant's legs are passive otherwise)"""
        self.pd = (np.array(q_target, dtype=np.float64), float(kp), float(kd))

    def set_state(self, root, q, qd):
        self.e.root[:] = root; self.e.q[:] = q; self.e.qd[:] = qd
        self.e.lam[:] = 0

    def set_gravity(self, g):
        self.e.set_params(**dict(self.e.params_dict, gravity=tuple(float(x) for x in g)))

    def step(self, k):
        for _ in range(k):
            if self.pd is not None:
                self.tau = self.pd[1] * (self.pd[0] - self.e.q) - self.pd[2] * self.e.qd
            self.e.step(self.tau, env_mu=self.mu)

    def root(self):
        return np.array(self.e.root, dtype=np.float64)

    def joints(self):
        return np.array(self.e.q, dtype=np.float64)


class EngineRig:
    """isaacgymenvs_amd env (either backend): state through the engine's own tensors, gym.simulate() through the C ABI"""

    def __init__(self, env):
        import torch
        self.env, self.t, self.torch = env, env.engine.tensors, torch
        self.dev = self.t["root_states"].device
        self.pd = None

    def hold(self, q_target, kp, kd):
        self.pd = (self._put(q_target), float(kp), float(kd))

    def _put(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a), dtype=self.torch.float32, device=self.dev)

    def set_state(self, root, q, qd):
        t = self.t
        t["root_states"][:] = self._put(root); self.env.dof_pos[:] = self._put(q); self.env.dof_vel[:] = self._put(qd)
        for k in ("contact_impulse", "limit_impulse", "dof_actuation_force", "self_contact_impulse"):
            if k in t:
                t[k].zero_()

    def set_gravity(self, g):
        for ax, v in zip("xyz", g):
            self.env.engine.set_option("gravity_" + ax, float(v))

    def step(self, k):
        for _ in range(k):
            if self.pd is not None:
                self.t["dof_actuation_force"][:] = self.pd[1] * (self.pd[0] - self.env.dof_pos) - self.pd[2] * self.env.dof_vel
            self.env.engine.simulate()

    def root(self):
        if self.dev.type == "cuda":
            self.torch.cuda.synchronize()
        return self.t["root_states"].detach().cpu().numpy().astype(np.float64)

    def joints(self):
        return self.env.dof_pos.detach().cpu().numpy().astype(np.float64)


def rest_pose(spec, n, robot):
    """a pose from which the robot settles onto its contact spheres without toppling"""
    lo, up = np.minimum(spec.dof_lower, spec.dof_upper), np.maximum(spec.dof_lower, spec.dof_upper)
    root = np.zeros((n, 13)); root[:, 6] = 1.0
    if robot == "ant":
        root[:, 2] = 0.62          # above the ground with the legs at mid range: it drops a few centimetres onto its feet
        q = np.tile(0.5 * (lo + up), (n, 1))
    else:           # humanoid: lying on its back (root frame pitched by -90 degrees), limbs straight
        root[:, 2] = 0.20
        root[:, 3:7] = [0.0, -np.sin(np.pi / 4), 0.0, np.cos(np.pi / 4)]
        q = np.tile(np.where(lo > 0, lo, np.where(up < 0, up, 0.0)), (n, 1))
    return root, q, np.zeros_like(q)


def friction_known_answers(rig, spec, robot, mu, dt, settle=150):
    """runs the three scenarios on one rig and returns the measured numbers next to the closed forms (the caller asserts)"""
    n = rig.root().shape[0]
    out = {}
    root0, q0, qd0 = rest_pose(spec, n, robot)
    if robot == "ant":
        rig.hold(q0, 8.0, 0.3)             # legs held (explicit PD, gains inside its stability bound for the 0.01 armature): the body slides as one piece
    rig.set_gravity((0.0, 0.0, -G)); rig.set_state(root0, q0, qd0); rig.step(settle)
    rest = rig.root()
    out["settled_speed"] = float(np.abs(rest[:, 7:10]).max())
    # ---- below the friction angle: static friction holds for 2 s
    th = np.arctan(mu) * 0.6
    rig.set_gravity((G * np.sin(th), 0.0, -G * np.cos(th))); rig.step(int(round(2.0 / dt)))
    r = rig.root()
    out["stick_shift"], out["stick_speed"] = float(np.abs(r[:, 0] - rest[:, 0]).max()), float(np.abs(r[:, 7:10]).max())
    # ---- above it: a = g (sin - mu cos), measured on the root velocity over a 0.5 s window after 0.25 s of sliding
    rig.set_gravity((0.0, 0.0, -G)); rig.set_state(root0, q0, qd0); rig.step(settle)
    th = np.arctan(mu) + np.deg2rad(13.0)
    # downhill along a diagonal of the tangent axes: the artefact needs both tangent rows to carry the sliding velocity
    c = np.sqrt(0.5)
    rig.set_gravity((G * np.sin(th) * c, G * np.sin(th) * c, -G * np.cos(th)))
    # (a short window: on this slope the lying Humanoid starts to roll after about a second -- sooner where the limb-wave kernels' contact caps
    # leave its arms without contacts -- and a rolling body's root is no longer its centre of mass)
    k0, k1 = int(round(0.25 / dt)), int(round(0.5 / dt))
    rig.step(k0); ra = rig.root(); rig.step(k1); rb = rig.root()
    dv = (rb[:, 7:9] - ra[:, 7:9]) / (k1 * dt)
    out["slide_acc"] = (dv[:, 0] + dv[:, 1]) * c                      # along the fall line
    out["slide_acc_lateral"] = (dv[:, 0] - dv[:, 1]) * c              # across it: zero
    out["slide_acc_expected"] = G * (np.sin(th) - mu * np.cos(th))
    out["slide_acc_frictionless"] = G * np.sin(th)
    out["slide_theta"] = th
    # ---- level ground, pushed along a diagonal: stops after v0^2 / (2 mu g) on the line of the push
    rig.set_gravity((0.0, 0.0, -G)); rig.set_state(root0, q0, qd0); rig.step(settle)
    r = rig.root()
    v0 = 1.0
    root1 = r.copy(); root1[:, 7:13] = 0.0; root1[:, 7] = v0 * c; root1[:, 8] = v0 * c
    rig.set_state(root1, rig.joints(), qd0)           # (the warm-start impulses are zeroed with it: the first sub-step finds the weight again)
    rig.step(int(round(1.5 / dt)))
    e = rig.root()
    d = e[:, 0:2] - r[:, 0:2]
    out["stop_dist"] = (d[:, 0] + d[:, 1]) * c
    out["stop_lateral"] = (d[:, 0] - d[:, 1]) * c
    out["stop_dist_expected"] = v0 * v0 / (2.0 * mu * G)
    out["stop_speed"] = float(np.abs(e[:, 7:9]).max())
    return out

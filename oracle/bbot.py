"""CPU oracle of the BallBalance physics step (free tray on three two-joint legs + one free ball).  TEST INFRASTRUCTURE ONLY.

Same stated algorithm as csrc/core/bbot_engine.hpp, deliberately written differently: dense generalized-coordinate matrices from
oracle/physics.c (or_dynamics / or_point_jac / or_energy), numpy linear algebra in fp64, PGS in generalized-velocity space.  What
it replaces in the reference: gym.simulate() for the BallBalance task (reference isaacgymenvs/tasks/ball_balance.py; closed PhysX
=> PARITY UNPINNED).

  bot     : floating tray + 6 hinges (joint-space dynamics), implicit PD position drives on the three lower-leg joints
            (DOF_MODE_POS, stiffness 4000, damping 100, ball_balance.py:273-281), joint-limit rows
  pins    : the reference's rigid-body attractors (stiffness 5e7, damping 5e3, AXIS_TRANSLATION, :285-300) hold the far end of
            each lower leg at a fixed world point.  Implicit spring-damper = soft equality rows: with k, c and step h
                J v+ + gamma lam = -beta x,   gamma = 1 / (h (h k + c)),   beta = k / (h k + c),   lam = h * force
            three rows (world x, y, z) per foot, warm started
  ball    : free sphere (radius 0.1, density 200, :262-266), gravity on
  contact : ball against the tray's solid cylinder (closest point, exact), one contact of 3 rows (normal + friction disc, mu 1)
            over [tray's 6 root dofs | 6 ball dofs], no warm start.  The legs and the ground are never reached: the episode ends
            when the ball centre drops below 1.5 radii (:473).
  sensors : the reference puts three force sensors on the tray (:254-260).  Reported here: the net non-gravity wrench on the tray
            body over the step (leg joints + ball contact), from its momentum balance, in the tray frame about each sensor origin.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .engine import OracleEngine, _ptr
from .hand import contact_frame, friction_step, quat2mat

ATT_K, ATT_C = 5e7, 5e3                 # ball_balance.py:287-288
DRIVE_KP, DRIVE_KD = 4000.0, 100.0      # :276-277
ACTUATED = (1, 3, 5)                    # :271
BALL_RADIUS, BALL_DENSITY = 0.1, 200.0  # :263-265
BALL_MASS = BALL_DENSITY * 4.0 / 3.0 * math.pi * BALL_RADIUS ** 3
BALL_INERTIA = 0.4 * BALL_MASS * BALL_RADIUS ** 2
MU = 1.0                                # default shape friction on both sides


def sphere_cylinder(c_local, r, radius, half):
    """signed distance of a sphere (centre in the cylinder's frame, axis z) to a solid cylinder, outward normal (cylinder frame)."""
    rho = math.hypot(c_local[0], c_local[1])
    s = min(1.0, radius / rho) if rho > 0 else 1.0
    qc = np.array([c_local[0] * s, c_local[1] * s, min(max(c_local[2], -half), half)])
    d = c_local - qc
    nd = np.linalg.norm(d)
    if nd > 1e-12:
        return nd - r, d / nd
    # centre inside the solid: leave through the nearer flat face
    pen = half - abs(c_local[2])
    return -pen - r, np.array([0.0, 0.0, 1.0 if c_local[2] >= 0 else -1.0])


class OracleBbotEngine:
    def __init__(self, spec, dims, num_envs, sim: dict, foot_bodies):
        self.spec, self.dims, self.N, self.sim = spec, dims, num_envs, sim
        self.eng = OracleEngine(spec, num_envs, params=sim, sensor_bodies=list(foot_bodies), precision="f64")
        self.feet = list(foot_bodies)
        N, nd = num_envs, spec.nd
        self.nd = nd
        self.eng.root[:, 2] = dims["tray_height"]
        self.targets = np.zeros((N, nd))
        self.ball = np.zeros((N, 13)); self.ball[:, 6] = 1.0
        self.lam_pin = np.zeros((N, 9))
        self.sensor = np.zeros((N, 18))
        self.ncontacts = np.zeros(N, int)
        self.lo = np.minimum(spec.dof_lower, spec.dof_upper); self.up = np.maximum(spec.dof_lower, spec.dof_upper)
        a = dims["leg_angles"]
        self.pin_target = np.array([[dims["leg_outer_offset"] * math.cos(x), dims["leg_outer_offset"] * math.sin(x), dims["leg_radius"]] for x in a])
        self.pin_offset = np.array([0.0, 0.0, 0.5 * dims["leg_length"]])                 # :299, in the lower leg's frame
        self.sensor_pos = np.array([[dims["leg_outer_offset"] * math.cos(x), dims["leg_outer_offset"] * math.sin(x), 0.0] for x in a])   # :256-259

    @property
    def q(self): return self.eng.q
    @property
    def qd(self): return self.eng.qd
    @property
    def root(self): return self.eng.root
    @property
    def laml(self): return self.eng.lam[:, :]     # nsph = 0 => lam is just the limit impulses

    def step(self):
        P = self.sim
        h = P["dt"] / P["substeps"]
        for _ in range(P["substeps"]):
            for e in range(self.N):
                self._substep_env(e, h)

    def _substep_env(self, e, h):
        P, nd, spec, dm = self.sim, self.nd, self.spec, self.dims
        nv = nd + 6
        root = self.eng.root[e]
        q, qd, tgt = self.q[e].copy(), self.qd[e].copy(), self.targets[e]
        v0 = np.concatenate([root[7:13], qd])
        M, bias = self.eng.dynamics(e)
        kp = np.zeros(nd); kd = np.zeros(nd)
        kp[list(ACTUATED)] = DRIVE_KP; kd[list(ACTUATED)] = DRIVE_KD
        D = np.array(spec.dof_damping, float)
        Mh = M.copy()
        Mh[6:, 6:] += np.diag(np.array(spec.dof_armature, float) + h * (D + kd) + h * h * kp)
        rhs = -bias
        rhs[6:] += kp * (tgt - q) - (D + kd + h * kp) * qd
        Minv = np.linalg.inv(Mh)
        v = v0 + h * (Minv @ rhs)
        g = np.array(P["gravity"], float)
        xb, qb = self.ball[e, 0:3].copy(), self.ball[e, 3:7].copy()
        vball = np.concatenate([self.ball[e, 7:10] + h * g, self.ball[e, 10:13]])
        Mbinv = np.diag([1 / BALL_MASS] * 3 + [1 / BALL_INERTIA] * 3)
        _, _, bp = self.eng.energy(e, poses=True)
        O = root[0:3].copy()
        s_state = np.ascontiguousarray(self.eng.state[e])
        J3 = np.zeros((3, nv))
        rows = []
        # ---- joint limits (as oracle/physics.c)
        for d in range(nd):
            if not spec.dof_limited[d]:
                self.laml[e, d] = 0.0
                continue
            dl, du = q[d] - self.lo[d], self.up[d] - q[d]
            Cc, s = (dl, 1.0) if dl < du else (du, -1.0)
            lw = self.laml[e, d]
            l0 = (0.0 if lw * s < 0 else abs(lw)) * P["warm"]
            Jt = np.zeros(nv); Jt[6 + d] = s
            vt = -Cc / h if Cc >= 0 else min(-Cc * P["erp"] / h, P["max_depen_vel"])
            rows.append(dict(Jt=Jt, Jb=np.zeros(6), vt=vt, lam=l0, gamma=0.0, kind="lim", d=d, s=s))
        # ---- attractors
        gamma = 1.0 / (h * (h * ATT_K + ATT_C))
        beta = ATT_K / (h * ATT_K + ATT_C)
        for j, b in enumerate(self.feet):
            p = bp[b, 0:3] + bp[b, 3:12].reshape(3, 3) @ self.pin_offset
            self.eng.lib.or_point_jac(C.byref(self.eng.model), _ptr(s_state), b, _ptr(np.ascontiguousarray(p - O)), _ptr(J3))
            x = p - self.pin_target[j]
            for k in range(3):
                rows.append(dict(Jt=J3[k].copy(), Jb=np.zeros(6), vt=-beta * x[k], lam=self.lam_pin[e, 3 * j + k] * P["warm"], gamma=gamma,
                                 kind="pin", idx=3 * j + k))
        # ---- ball against the tray
        Rt = quat2mat(root[3:7])
        dist, nl = sphere_cylinder(Rt.T @ (xb - O), BALL_RADIUS, dm["tray_radius"], 0.5 * dm["tray_thickness"])
        ncon = 0
        if dist < P["contact_offset"]:
            n = Rt @ nl                                                   # from the tray towards the ball
            t1, t2 = contact_frame(n)
            pc = xb - BALL_RADIUS * n
            self.eng.lib.or_point_jac(C.byref(self.eng.model), _ptr(s_state), 0, _ptr(np.ascontiguousarray(pc - O)), _ptr(J3))
            rc = pc - xb
            gap = dist - P["rest_offset"]
            vtn = -gap / h if gap >= 0 else min(-gap * P["erp"] / h, P["max_depen_vel"])
            for k, u in enumerate((n, t1, t2)):
                rows.append(dict(Jt=-(u @ J3), Jb=np.concatenate([u, np.cross(rc, u)]), vt=vtn if k == 0 else 0.0, lam=0.0, gamma=0.0, kind="con", k=k))
            ncon = 1
        self.ncontacts[e] = ncon
        for r in rows:
            r["Bt"] = Minv @ r["Jt"]; r["Bb"] = Mbinv @ r["Jb"]
            r["Ainv"] = 1.0 / (P["cfm"] + r["gamma"] + r["Jt"] @ r["Bt"] + r["Jb"] @ r["Bb"])
            if r["lam"] != 0.0:
                v += r["Bt"] * r["lam"]
        for _ in range(P["iters"]):
            i = 0
            while i < len(rows):
                r = rows[i]
                if r["kind"] == "lim":
                    nl_ = max(r["lam"] - (r["Jt"] @ v - r["vt"]) * r["Ainv"], 0.0)
                    dl = nl_ - r["lam"]; r["lam"] = nl_
                    v += r["Bt"] * dl
                    i += 1
                elif r["kind"] == "pin":
                    dl = -(r["Jt"] @ v - r["vt"] + r["gamma"] * r["lam"]) * r["Ainv"]
                    r["lam"] += dl
                    v += r["Bt"] * dl
                    i += 1
                else:
                    rn, ra, rb = rows[i], rows[i + 1], rows[i + 2]
                    vn = rn["Jt"] @ v + rn["Jb"] @ vball
                    ln = max(rn["lam"] - (vn - rn["vt"]) * rn["Ainv"], 0.0)
                    dl = ln - rn["lam"]; rn["lam"] = ln
                    v += rn["Bt"] * dl; vball += rn["Bb"] * dl
                    # both tangent rows from the SAME velocity, then the disc projection, one application (round 6: solving and applying t1
                    # before t2 is looked at let a fast-sliding contact's friction point off the sliding direction)
                    vtan = [rt["Jt"] @ v + rt["Jb"] @ vball for rt in (ra, rb)]
                    lt = friction_step([rt["lam"] - vt_ * rt["Ainv"] for rt, vt_ in zip((ra, rb), vtan)], (ra["lam"], rb["lam"]), vtan,
                                       (ra["Ainv"], rb["Ainv"]), MU * ln)
                    for rt, nl_ in zip((ra, rb), lt):
                        dl = nl_ - rt["lam"]; rt["lam"] = nl_
                        v += rt["Bt"] * dl; vball += rt["Bb"] * dl
                    i += 3
        # ---- outputs
        ll = np.zeros(nd)
        for r in rows:
            if r["kind"] == "lim":
                ll[r["d"]] = r["lam"] * r["s"]
            elif r["kind"] == "pin":
                self.lam_pin[e, r["idx"]] = r["lam"]
        self.laml[e] = ll
        # net non-gravity wrench on the tray (its centre of mass is the body origin): F = m (dv / h - g), T = I dw / h + w x I w
        mt = float(spec.mass[0])
        ixx, iyy, izz = spec.inertia[0][0:3]
        Iw = Rt @ np.diag([ixx, iyy, izz]) @ Rt.T
        F = mt * ((v[0:3] - v0[0:3]) / h - g)
        T = Iw @ ((v[3:6] - v0[3:6]) / h) + np.cross(v0[3:6], Iw @ v0[3:6])
        for i in range(3):
            Ti = T - np.cross(Rt @ self.sensor_pos[i], F)
            self.sensor[e, 6 * i:6 * i + 3] = Rt.T @ F
            self.sensor[e, 6 * i + 3:6 * i + 6] = Rt.T @ Ti
        # ---- integrate (tray as oracle/physics.c, ball as oracle/hand.py)
        self.qd[e] = v[6:]; self.q[e] = q + h * v[6:]
        root[7:13] = v[0:6]
        root[0:3] += h * v[0:3]
        root[3:7] = _integrate_quat(root[3:7], v[3:6], h)
        self.ball[e, 7:13] = vball
        self.ball[e, 0:3] = xb + h * vball[:3]
        self.ball[e, 3:7] = _integrate_quat(qb, vball[3:], h)


def _integrate_quat(Q, om, h):
    an = np.linalg.norm(om); th = an * h
    if th > 1e-12:
        dq = np.concatenate([om * np.sin(th / 2) / an, [np.cos(th / 2)]])
    else:
        dq = np.concatenate([om * h / 2, [1.0]])
    x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1]
    y = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0]
    z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3]
    w = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2]
    qn = np.array([x, y, z, w])
    return qn / np.linalg.norm(qn)

"""CPU oracle of the SCENE physics step: a fixed-base articulated actor beside free and static boxes.  TEST INFRASTRUCTURE ONLY.

Same stated algorithm as csrc/core/scene_engine.hpp, deliberately written differently: dense generalized-coordinate matrices from
oracle/physics.c (or_dynamics / or_point_jac / or_energy), numpy linear algebra in fp64, PGS in generalized-velocity space, the boxes'
6 x 6 inverse mass matrices explicit.  What it replaces in the reference: gym.simulate() for envs that hold more than one actor -- reference
isaacgymenvs/tasks/franka_cube_stack.py:204-233,323-339 (arm + table + stand + two cubes; closed PhysX => PARITY UNPINNED).

  actor  : joint-space dynamics (its own gravity per asset option disable_gravity), efforts, passive spring / damper, implicit per-dof
           position drives, joint-limit rows with warm start
  boxes  : free rigid boxes (principal inertias along the box axes, gravity on, velocity clamps 1000 m/s / 64 rad/s), static boxes
  contact: actor collision spheres vs free / static boxes (at most KARM, sphere order, free boxes before static ones); the corners of every
           free box vs ground plane, static boxes, the other free boxes, then the static boxes' corners vs the free boxes, then one EDGE-EDGE
           contact per box pair whose least-penetration axis is edge x edge (at most KBOX together); 3 rows each (normal + friction disc); friction = mean of
           the two sides'; warm start by FEATURE (sphere x target / corner x target): a contact starts its first sweep from the impulses the
           same feature ended the last sub-step with (applied when the sweep reaches it)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import OracleEngine, _ptr
from .hand import contact_frame, quat2mat, sphere_box3

KARM, KBOX = 24, 24                   # csrc/core/scene_engine.hpp SceneSim::KARM / KBOX
MAX_W, MAX_V = 64.0, 1000.0           # csrc/core/engine.hpp kMaxAngularVelocity / kMaxLinearVelocity


def box_edge_contact(Ra, xa, ha, Rb, xb, hb):
    """Edge-edge contact of two boxes (rotation, centre, half sizes each) by the separating-axis test: among the 15 axes (3 + 3 face normals, 9
    edge x edge) the one of LEAST penetration says which features touch.  Returns None unless that axis is an edge x edge one -- by a margin
    (5 % + 0.1 mm), so that the ties of face contacts (a yawed cube flat on a table: a_x x b_y is the table's normal again) stay face contacts,
    which the corner-in-box tests cover --, else (separation, unit normal from B towards A, contact point, edge axis of A, edge axis of B): the
    point is the middle of the closest points of the two supporting edges.  (csrc/core/scene_engine.hpp scene_box_edge; crossed edges have no
    corner inside the other box, so without this a cube pushed edge-on across another's edge passes through it: VERDICT r5 #5 / #9.)"""
    ha, hb = np.asarray(ha, float), np.asarray(hb, float)
    d = xa - xb

    def sep(ax):
        return abs(ax @ d) - sum(ha[k] * abs(ax @ Ra[:, k]) for k in range(3)) - sum(hb[k] * abs(ax @ Rb[:, k]) for k in range(3))

    s_face = max([sep(Ra[:, k]) for k in range(3)] + [sep(Rb[:, k]) for k in range(3)])
    best = None
    for i in range(3):
        for j in range(3):
            c = np.cross(Ra[:, i], Rb[:, j])
            l = np.linalg.norm(c)
            if l < 1e-3:                       # (nearly) parallel edges: a face axis covers the pair
                continue
            s = sep(c / l)
            if best is None or s > best[0]:
                best = (s, i, j, c / l)
    if best is None or not best[0] > s_face + 0.05 * abs(s_face) + 1e-4:
        return None
    s, i, j, ax = best
    n = ax if ax @ d > 0 else -ax
    pa, pb = xa.copy(), xb.copy()
    for k in range(3):
        if k != i:
            pa = pa - (1.0 if n @ Ra[:, k] >= 0 else -1.0) * ha[k] * Ra[:, k]        # the edge of A farthest along -n
        if k != j:
            pb = pb + (1.0 if n @ Rb[:, k] >= 0 else -1.0) * hb[k] * Rb[:, k]        # the edge of B farthest along +n
    u, w = Ra[:, i], Rb[:, j]
    r, uw = pb - pa, float(Ra[:, i] @ Rb[:, j])
    den = 1.0 - uw * uw
    al = min(max(((r @ u) - (r @ w) * uw) / den, -ha[i]), ha[i])
    be = min(max(((r @ u) * uw - (r @ w)) / den, -hb[j]), hb[j])
    return s, n, 0.5 * ((pa + al * u) + (pb + be * w)), i, j


def box_face_crossings(Ra, xa, ha, Rb, xb, hb):
    """The part of a FACE contact of two boxes that no corner-in-box test sees: where the outline of the incident face crosses the outline of the
    reference face (two planks lying crossed on each other touch in a rectangle none of whose corners is a corner of either; a plank on a knife
    edge).  Reference face: the face axis of least penetration, B's unless one of A's is better by the margin of box_edge_contact (5 % + 0.1 mm;
    nothing when an edge x edge axis beats the faces by that margin -- box_edge_contact's case); incident face: the other box's face most
    anti-parallel to it.  Each of its four edges is clipped to the reference face's rectangle (in the reference box's frame); a clip point that is
    not an end of the edge is a contact: (separation from the reference face, unit normal from B towards A, point, id = 2 * edge + end).
    csrc/core/scene_engine.hpp scene_face_crossings."""
    ha, hb = np.asarray(ha, float), np.asarray(hb, float)
    d = xa - xb

    def sep(ax):
        return abs(ax @ d) - sum(ha[k] * abs(ax @ Ra[:, k]) for k in range(3)) - sum(hb[k] * abs(ax @ Rb[:, k]) for k in range(3))

    sA, sB = [sep(Ra[:, k]) for k in range(3)], [sep(Rb[:, k]) for k in range(3)]
    kA, kB = int(np.argmax(sA)), int(np.argmax(sB))
    s_face = max(sA[kA], sB[kB])
    for i in range(3):
        for j in range(3):
            c = np.cross(Ra[:, i], Rb[:, j])
            l = np.linalg.norm(c)
            if l >= 1e-3 and sep(c / l) > s_face + 0.05 * abs(s_face) + 1e-4:
                return []
    ref_a = sA[kA] > sB[kB] + 0.05 * abs(sB[kB]) + 1e-4
    if ref_a:
        Rr, xr, hr, k, Ri, xi, hi = Ra, xa, ha, kA, Rb, xb, hb
    else:
        Rr, xr, hr, k, Ri, xi, hi = Rb, xb, hb, kB, Ra, xa, ha
    nr = Rr[:, k] * (1.0 if Rr[:, k] @ (xi - xr) >= 0 else -1.0)          # the reference face's outward normal (towards the incident box)
    m = int(np.argmax(np.abs(Ri.T @ nr)))
    sI = -1.0 if Ri[:, m] @ nr >= 0 else 1.0
    m1, m2 = (m + 1) % 3, (m + 2) % 3
    cI = xi + Ri[:, m] * sI * hi[m]
    verts = [cI + s1 * hi[m1] * Ri[:, m1] + s2 * hi[m2] * Ri[:, m2] for s1, s2 in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
    k1, k2 = (k + 1) % 3, (k + 2) % 3
    out = []
    for e in range(4):
        p0, p1 = verts[e], verts[(e + 1) % 4]
        q0, q1 = Rr.T @ (p0 - xr), Rr.T @ (p1 - xr)
        t0, t1 = 0.0, 1.0
        for c_ in (k1, k2):
            dq = q1[c_] - q0[c_]
            for bound, sg in ((hr[c_], 1.0), (hr[c_], -1.0)):             # sg * q <= bound
                num, den = bound - sg * q0[c_], sg * dq
                if abs(den) < 1e-12:
                    if num < 0:
                        t0, t1 = 1.0, 0.0
                elif den > 0:
                    t1 = min(t1, num / den)
                else:
                    t0 = max(t0, num / den)
        if t0 > t1:
            continue
        for end, t in ((0, t0), (1, t1)):
            if (end == 0 and t > 1e-6) or (end == 1 and t < 1.0 - 1e-6):
                p = p0 + t * (p1 - p0)
                out.append((float(nr @ (p - xr) - hr[k]), (nr if not ref_a else -nr), p, 2 * e + end))
    return out


class OracleSceneEngine:
    def __init__(self, spec, num_envs, sim: dict, scene: dict, kp, kd, drive_vmax=None):
        """scene: dict(arm_gravity=bool, arm_mu=float, free=[dict(half=[3], mass=, inertia=[3], mu=, pose=[7])...],
        static=[dict(pos=[3], quat=[4], half=[3], mu=)...]); kp / kd: per-dof position-drive gains"""
        self.spec, self.N, self.sim, self.scene = spec, num_envs, sim, scene
        g = tuple(sim["gravity"]) if scene.get("arm_gravity", True) else (0.0, 0.0, 0.0)
        self.eng = OracleEngine(spec, num_envs, params=dict(sim, gravity=g), sensor_bodies=[0], precision="f64")
        self.nd = spec.nd
        self.kp, self.kd = np.asarray(kp, float), np.asarray(kd, float)
        # velocity limits of the position drives (<= 0: none): the drive's position error is clamped to vmax kd / kp (csrc/tasks/articulation.hpp)
        self.drive_vmax = np.zeros(spec.nd) if drive_vmax is None else np.asarray(drive_vmax, float)
        self.free, self.static = list(scene.get("free", [])), list(scene.get("static", []))
        self.box = np.zeros((num_envs, max(len(self.free), 1), 13))
        self.box[:, :, 6] = 1.0
        for i, f in enumerate(self.free):
            self.box[:, i, :7] = f["pose"]
        self.targets = np.zeros((num_envs, self.nd))
        self.laml = np.zeros((num_envs, self.nd))              # warm-start impulses of the joint-limit rows
        self.warm = [dict() for _ in range(num_envs)]          # feature id -> impulses (normal, two tangents) of the last sub-step's contacts
        self.dof_force = np.zeros((num_envs, self.nd))
        self.ncontacts = np.zeros(num_envs, int)
        self.refused = np.zeros(num_envs, int)
        self.contact_forces = [[] for _ in range(num_envs)]     # last sub-step: (ia, ib, world force on side A)
        self.netf = np.zeros((num_envs, spec.nb, 3))            # last sub-step: net contact force on every actor body (world frame)
        self.lo = np.minimum(spec.dof_lower, spec.dof_upper); self.up = np.maximum(spec.dof_lower, spec.dof_upper)
        sb = np.asarray(spec.sph_body)
        assert np.all(np.diff(sb) >= 0), "collision spheres are listed body by body"

    @property
    def q(self): return self.eng.q
    @property
    def qd(self): return self.eng.qd
    @property
    def root(self): return self.eng.root

    def step(self, tau):
        h = self.sim["dt"] / self.sim["substeps"]
        tau = np.asarray(tau, float).reshape(self.N, self.nd)
        for _ in range(self.sim["substeps"]):
            for e in range(self.N):
                self._substep_env(e, h, tau[e])

    def _substep_env(self, e, h, tau):
        P, nd, spec = self.sim, self.nd, self.spec
        q, qd, tgt = self.q[e].copy(), self.qd[e].copy(), self.targets[e].copy()
        for d in range(nd):
            if self.drive_vmax[d] > 0 and self.kp[d] > 0 and self.kd[d] > 0:      # (a drive without a damper is bounded by the velocity clamp alone)
                emax = self.drive_vmax[d] * self.kd[d] / self.kp[d]
                tgt[d] = q[d] + min(max(tgt[d] - q[d], -emax), emax)
        M, bias = self.eng.dynamics(e)
        K, D = np.array(spec.dof_stiffness, float), np.array(spec.dof_damping, float)
        kp, kd = self.kp, self.kd
        Mh = M + np.diag(np.array(spec.dof_armature, float) + h * (D + kd) + h * h * (K + kp))
        rhs = tau - bias - K * (q - np.array(spec.dof_springref, float)) - (D + h * K) * qd + kp * (tgt - q) - (kd + h * kp) * qd
        Minv = np.linalg.inv(Mh)
        v = qd + h * (Minv @ rhs)
        g = np.array(P["gravity"], float)
        nf = len(self.free)
        vb = [np.concatenate([self.box[e, i, 7:10] + h * g, self.box[e, i, 10:13]]) for i in range(nf)]
        Rf = [quat2mat(self.box[e, i, 3:7]) for i in range(nf)]
        xf = [self.box[e, i, 0:3].copy() for i in range(nf)]
        Binv = []
        for i, f in enumerate(self.free):
            W = np.zeros((6, 6))
            W[:3, :3] = np.eye(3) / f["mass"]
            W[3:, 3:] = Rf[i] @ np.diag(1.0 / np.asarray(f["inertia"], float)) @ Rf[i].T
            Binv.append(W)
        Rs = [quat2mat(np.asarray(s["quat"], float)) for s in self.static]
        _, _, bp = self.eng.energy(e, poses=True)
        O = self.eng.root[e, :3]
        s_state = np.ascontiguousarray(self.eng.state[e])
        J3 = np.zeros((3, nd))

        def vtarget(dist):
            gap = dist - P["rest_offset"]
            return -gap / h if gap >= 0 else min(-gap * P["erp"] / h, P["max_depen_vel"])

        rows = []
        for d in range(nd):
            if not spec.dof_limited[d]:
                self.laml[e, d] = 0.0
                continue
            dl, du = q[d] - self.lo[d], self.up[d] - q[d]
            Cc, s = (dl, 1.0) if dl < du else (du, -1.0)
            lw = self.laml[e, d]
            l0 = (0.0 if lw * s < 0 else abs(lw)) * P["warm"]
            Jh = np.zeros(nd); Jh[d] = s
            vt = -Cc / h if Cc >= 0 else min(-Cc * P["erp"] / h, P["max_depen_vel"])
            rows.append(dict(kind="lim", Jh=Jh, vt=vt, lam=l0, d=d, s=s))
        contacts = []      # dict(Jh[3][nd] or None, ia, ib, n, t1, t2, pc, vtn, mu)
        moved = [False] * spec.nb                    # does any dof sit on the body or on one of its ancestors
        for b_ in range(spec.nb):
            a = b_
            while a >= 0 and not moved[b_]:
                moved[b_] = bool((np.asarray(spec.dof_body) == a).any())
                a = int(spec.parent[a])
        refused = 0
        narm = 0
        for si in range(len(spec.sph_body)):
            b = int(spec.sph_body[si])
            cs = bp[b, 0:3] + bp[b, 3:12].reshape(3, 3) @ np.asarray(spec.sph_pos[si], float)
            rad = float(spec.sph_rad[si])
            for t in range(nf + len(self.static)):
                if t >= nf and not moved[b]:
                    continue               # a body no dof moves (the fixed base link) against a static box: the row would act on nothing
                if t < nf:
                    Rb, xb, hb, mub, ib = Rf[t], xf[t], self.free[t]["half"], self.free[t]["mu"], t
                else:
                    s_ = self.static[t - nf]
                    Rb, xb, hb, mub, ib = Rs[t - nf], np.asarray(s_["pos"], float), s_["half"], s_["mu"], -1
                dist, nl = sphere_box3(Rb.T @ (cs - xb), rad, hb)
                if not dist < P["contact_offset"]:
                    continue
                if narm >= KARM:
                    refused += 1
                    continue
                n = Rb @ nl
                t1, t2 = contact_frame(n)
                pc = cs - rad * n
                self.eng.lib.or_point_jac(C.byref(self.eng.model), _ptr(s_state), b, _ptr(np.ascontiguousarray(pc - O)), _ptr(J3))
                contacts.append(dict(Jh=[u @ J3 for u in (n, t1, t2)], ia=-1, ib=ib, body=b, fr=(n, t1, t2), pc=pc, vtn=vtarget(dist),
                                     mu=0.5 * (self.scene.get("arm_mu", 1.0) + mub), fid=1 + si * 8 + (t if t < nf else 4 + (t - nf))))
                narm += 1
        nbox = 0
        for i in range(nf):
            hi = np.asarray(self.free[i]["half"], float)
            for cr in range(8):
                pl = np.array([hi[0] if cr & 1 else -hi[0], hi[1] if cr & 2 else -hi[1], hi[2] if cr & 4 else -hi[2]])
                pc = xf[i] + Rf[i] @ pl
                for t in range(-1, len(self.static) + nf):
                    if t >= len(self.static) and t - len(self.static) == i:
                        continue
                    if t < 0:
                        n, dist, mub, ib = np.array([0.0, 0.0, 1.0]), pc[2] - P["ground_z"], P["plane_mu"], -1
                    else:
                        if t < len(self.static):
                            s_ = self.static[t]
                            Rb, xb, hb, mub, ib = Rs[t], np.asarray(s_["pos"], float), s_["half"], s_["mu"], -1
                        else:
                            j = t - len(self.static)
                            Rb, xb, hb, mub, ib = Rf[j], xf[j], self.free[j]["half"], self.free[j]["mu"], j
                        dist, nl = sphere_box3(Rb.T @ (pc - xb), 0.0, hb)
                        n = Rb @ nl
                    if not dist < P["contact_offset"]:
                        continue
                    if nbox >= KBOX:
                        refused += 1
                        continue
                    t1, t2 = contact_frame(n)
                    contacts.append(dict(Jh=None, ia=i, ib=ib, fr=(n, t1, t2), pc=pc, vtn=vtarget(dist), mu=0.5 * (self.free[i]["mu"] + mub),
                                         fid=1 + len(spec.sph_body) * 8 + (i * 8 + cr) * 9 + (t + 1)))
                    nbox += 1
        # the corners of the STATIC boxes inside free boxes (a plate lying on a stand smaller than itself has no corner of its own in the stand): the
        # free box is pushed back along the inward normal of the face the corner is nearest to
        for t, s_ in enumerate(self.static):
            hs = np.asarray(s_["half"], float)
            for cr in range(8):
                pl = np.array([hs[0] if cr & 1 else -hs[0], hs[1] if cr & 2 else -hs[1], hs[2] if cr & 4 else -hs[2]])
                pc = np.asarray(s_["pos"], float) + Rs[t] @ pl
                for j in range(nf):
                    dist, nl = sphere_box3(Rf[j].T @ (pc - xf[j]), 0.0, self.free[j]["half"])
                    if not dist < P["contact_offset"]:
                        continue
                    if nbox >= KBOX:
                        refused += 1
                        continue
                    n = -(Rf[j] @ nl)
                    t1, t2 = contact_frame(n)
                    contacts.append(dict(Jh=None, ia=j, ib=-1, fr=(n, t1, t2), pc=pc, vtn=vtarget(dist), mu=0.5 * (self.free[j]["mu"] + s_["mu"]),
                                         fid=("sc", t, cr, j)))
                    nbox += 1
        # per box pair: the EDGE-EDGE contact when the least-penetration axis is the cross product of an edge of each (box_edge_contact), else -- a face
        # axis wins -- the points where the incident face's outline crosses the reference face's (box_face_crossings)
        for i in range(nf):
            others = [(("st", t), Rs[t], np.asarray(s_["pos"], float), s_["half"], s_["mu"], -1) for t, s_ in enumerate(self.static)]
            others += [(("fr", j), Rf[j], xf[j], self.free[j]["half"], self.free[j]["mu"], j) for j in range(i + 1, nf)]
            for tag, Rb, xb, hb, mub, ib in others:
                hit = box_edge_contact(Rf[i], xf[i], self.free[i]["half"], Rb, xb, hb)
                found = [] if hit is None else [(hit[0], hit[1], hit[2], ("ee", i, tag, hit[3], hit[4]))]
                found += [(d_, n_, p_, ("fc", i, tag, cid)) for d_, n_, p_, cid in box_face_crossings(Rf[i], xf[i], self.free[i]["half"], Rb, xb, hb)]
                for dist, n, pc, fid in found:
                    if not dist < P["contact_offset"]:
                        continue
                    if nbox >= KBOX:
                        refused += 1
                        continue
                    t1, t2 = contact_frame(n)
                    contacts.append(dict(Jh=None, ia=i, ib=ib, fr=(n, t1, t2), pc=pc, vtn=vtarget(dist), mu=0.5 * (self.free[i]["mu"] + mub), fid=fid))
                    nbox += 1
        self.ncontacts[e] = narm + nbox
        self.refused[e] += refused
        # ---- per row: Jacobians of the actor and the (at most two) boxes, responses, diagonal
        for r in rows:
            r["Bh"] = Minv @ r["Jh"]
            r["Ainv"] = 1.0 / (P["cfm"] + r["Jh"] @ r["Bh"])
            if r["lam"] != 0.0:
                v += r["Bh"] * r["lam"]
        for cdat in contacts:
            cdat["rows"] = []
            l0 = self.warm[e].get(cdat["fid"], (0.0, 0.0, 0.0))
            for k, u in enumerate(cdat["fr"]):
                r = dict(Jh=None if cdat["Jh"] is None else cdat["Jh"][k], lam=l0[k] * P["warm"])
                a = P["cfm"]
                if r["Jh"] is not None:
                    r["Bh"] = Minv @ r["Jh"]
                    a += r["Jh"] @ r["Bh"]
                for side, sgn in (("ia", 1.0), ("ib", -1.0)):
                    ix = cdat[side]
                    if ix >= 0:
                        J = sgn * np.concatenate([u, np.cross(cdat["pc"] - xf[ix], u)])
                        r["J" + side], r["B" + side] = J, Binv[ix] @ J
                        a += J @ r["B" + side]
                r["Ainv"] = 1.0 / a
                cdat["rows"].append(r)

        def rowvel(cdat, r):
            x = 0.0 if r["Jh"] is None else r["Jh"] @ v
            for side in ("ia", "ib"):
                if cdat[side] >= 0:
                    x += r["J" + side] @ vb[cdat[side]]
            return x

        def apply(cdat, r, dl):
            nonlocal v
            if r["Jh"] is not None:
                v = v + r["Bh"] * dl
            for side in ("ia", "ib"):
                if cdat[side] >= 0:
                    vb[cdat[side]] = vb[cdat[side]] + r["B" + side] * dl

        for it in range(P["iters"]):
            for r in rows:
                vn = r["Jh"] @ v
                nl_ = max(r["lam"] - (vn - r["vt"]) * r["Ainv"], 0.0)
                v = v + r["Bh"] * (nl_ - r["lam"])
                r["lam"] = nl_
            for cdat in contacts:
                rn, ra, rb = cdat["rows"]
                if it == 0:                      # last sub-step's impulses have not acted yet
                    for r in (rn, ra, rb):
                        apply(cdat, r, r["lam"])
                ln = max(rn["lam"] - (rowvel(cdat, rn) - cdat["vtn"]) * rn["Ainv"], 0.0)
                apply(cdat, rn, ln - rn["lam"]); rn["lam"] = ln
                # both tangent rows from the same velocity, each with its own step size, radial projection onto the disc (csrc/core/engine.hpp
                # friction_disc<false>: the scene keeps round 5's rule -- the isotropic step of the other engine forms loses a held cube within nine sweeps)
                lt = [rt["lam"] - rowvel(cdat, rt) * rt["Ainv"] for rt in (ra, rb)]
                lim = cdat["mu"] * ln
                nrm = np.hypot(lt[0], lt[1])
                sc = lim / max(nrm, 1e-30) if nrm > lim else 1.0
                for rt, l in zip((ra, rb), lt):
                    apply(cdat, rt, l * sc - rt["lam"]); rt["lam"] = l * sc
        for d in range(nd):                      # the asset's joint velocity limits: clamp of the solved velocities
            if self.drive_vmax[d] > 0:
                v[d] = min(max(v[d], -self.drive_vmax[d]), self.drive_vmax[d])
        # ---- outputs
        ll = np.zeros(nd)
        for r in rows:
            ll[r["d"]] = r["lam"] * r["s"]
        self.laml[e] = ll
        self.dof_force[e] = tau - K * (q - np.array(spec.dof_springref, float)) - D * v + ll / h + kp * (tgt - q) - kd * v
        self.warm[e] = {c_["fid"]: tuple(r["lam"] for r in c_["rows"]) for c_ in contacts}
        self.contact_forces[e] = [(c_["ia"], c_["ib"], sum(u * r["lam"] for u, r in zip(c_["fr"], c_["rows"])) / h) for c_ in contacts]
        # gym's net contact force tensor, the actor's rows: per actor body the sum of its contacts' forces (world frame, this sub-step)
        self.netf[e] = 0.0
        for c_, (_, _, f) in zip(contacts, self.contact_forces[e]):
            if "body" in c_:
                self.netf[e, c_["body"]] += f
        # ---- integrate
        self.qd[e] = v; self.q[e] = q + h * v
        for i in range(nf):
            vv = vb[i].copy()
            wn, ln_ = np.linalg.norm(vv[3:]), np.linalg.norm(vv[:3])
            if wn > MAX_W:
                vv[3:] *= MAX_W / wn
            if ln_ > MAX_V:
                vv[:3] *= MAX_V / ln_
            self.box[e, i, 7:13] = vv
            self.box[e, i, 0:3] = xf[i] + h * vv[:3]
            om = vv[3:]; an = np.linalg.norm(om); th = an * h
            dq = np.concatenate([om * np.sin(th / 2) / an, [np.cos(th / 2)]]) if th > 1e-12 else np.concatenate([om * h / 2, [1.0]])
            Q = self.box[e, i, 3:7]
            x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1]
            y = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0]
            z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3]
            w = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2]
            qn = np.array([x, y, z, w]); self.box[e, i, 3:7] = qn / np.linalg.norm(qn)

"""CPU oracle of the Shadow-Hand physics step (fixed-base 24-DoF hand + one free cube).  TEST INFRASTRUCTURE ONLY.

Same stated algorithm as csrc/core/hand_engine.hpp, deliberately written differently: dense generalized-coordinate
matrices from oracle/physics.c (or_dynamics / or_point_jac / or_energy), numpy linear algebra in fp64, PGS in
generalized-velocity space.  What it replaces in the reference: gym.simulate() for the ShadowHand task
(reference isaacgymenvs/tasks/shadow_hand.py; closed PhysX => PARITY UNPINNED).

  hand   : joint-space dynamics, gravity disabled (shadow_hand.py:239), implicit PD position drives
           tau = kp (target - q) - D qd (kp from the MJCF position actuators, shared.xml:250-269), joint-limit rows,
           4 fixed tendons as soft two-sided limits (limit_stiffness 30, damping 0.1, shadow_hand.py:256-266)
  object : free rigid cube (5 cm, density 567 => isotropic inertia), gravity on
  contact: hand collision geometry sampled by spheres (models/shadow_hand_extras.json) against the exact box; at most
           KMAX active contacts per env (taken in sphere order), 3 rows each (normal + friction disc), no warm start
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import OracleEngine, _ptr

KMAX = 12            # csrc/core/hand_engine.hpp MI_HAND_KMAX (16 until round 2: the 611-float store holds 12 slots)
BODY_CAP = 4        # contacts admitted per hand body, in the body's (farthest-point ordered) sphere order -- csrc/core/hand_engine.hpp
CUBE_HALF = 0.025
CUBE_MASS = 567.0 * 0.05 ** 3                      # cube_multicolor.urdf: box 0.05, density 567
CUBE_INERTIA = CUBE_MASS * 0.05 ** 2 / 6.0         # isotropic


def friction_step(lt, lam_t, vtan, ainv, lim):
    """oracle/physics.c friction_step: the per-row step `lt` stands inside the friction disc; a contact that slides takes one step size for both
    rows (the smaller of the two) and is scaled back onto the disc -- friction antiparallel to the sliding velocity; in between, the point of the
    segment between the two steps that lies on the circle (continuity)"""
    r = np.array(lt, float)
    if np.hypot(r[0], r[1]) <= lim:
        return [r[0], r[1]]
    ac = min(ainv)
    s = np.array([lam_t[0] - vtan[0] * ac, lam_t[1] - vtan[1] * ac])
    nrm = np.hypot(s[0], s[1])
    if nrm >= lim:
        sc = lim / max(nrm, 1e-30)
        return [s[0] * sc, s[1] * sc]
    d = r - s
    a, b, c = max(d @ d, 1e-30), s @ d, s @ s - lim * lim
    t = (np.sqrt(max(b * b - a * c, 0.0)) - b) / a
    x = s + t * d
    return [x[0], x[1]]


def quat2mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def contact_frame(n):
    a = np.array([1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]])
    na = np.linalg.norm(a)
    t1 = a / na if na > 1e-6 else np.array([0.0, 1.0, 0.0])       # n = +-x: fall back to y
    return t1, np.cross(n, t1)


def sphere_box(c_local, r, half):
    """signed distance of a sphere (centre in box frame) to a cube of half size `half`, outward normal (box frame)."""
    qc = np.clip(c_local, -half, half)
    d = c_local - qc
    nd = np.linalg.norm(d)
    if nd > 1e-12:
        return nd - r, d / nd
    pen = half - np.abs(c_local)
    i = int(np.argmin(pen))
    n = np.zeros(3)
    n[i] = 1.0 if c_local[i] >= 0 else -1.0
    return -pen[i] - r, n


def sphere_ellipsoid(c_local, r, axes):
    """first-order signed distance of a sphere (centre in the object frame) to an ellipsoid with semi-axes `axes`:
    f / |grad f| with f = |c / a| - 1, scaled to length; outward normal = normalised gradient (csrc/core/hand_engine.hpp)."""
    a = np.asarray(axes, float)
    u = c_local / a
    g = u / a
    k0 = np.linalg.norm(u); k1 = np.linalg.norm(g)
    if k1 * k1 <= 1e-20:
        return -a.min() - r, np.array([0.0, 0.0, 1.0])
    return k0 * (k0 - 1.0) / k1 - r, g / k1


def sphere_capsule(c_local, r, rc, hl):
    """exact signed distance of a sphere to a capsule along the object's z axis (radius rc, half length hl), outward normal."""
    pz = min(max(c_local[2], -hl), hl)
    d = np.array([c_local[0], c_local[1], c_local[2] - pz])
    n = np.linalg.norm(d)
    if n * n <= 1e-24:
        return -rc - r, np.array([1.0, 0.0, 0.0])
    return n - rc - r, d / n


def _hand_struct():
    r = C.c_double

    class OrHand(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("nos", "ntend", "kmax", "body_cap", "shape", "solver", "nlimb", "pad")] + \
                   [(n, C.c_void_p) for n in ("os_body", "os_pos", "os_rad", "tend_d0", "tend_d1", "tend_c0", "tend_c1", "tend_lo", "tend_hi")] + \
                   [("tend_stiffness", r), ("tend_damping", r), ("kp", C.c_void_p), ("obj_mass", r), ("obj_inertia", r * 3),
                    ("obj_dims", r * 3), ("mu", r), ("limb_of_body", C.c_void_p), ("limb_cap", C.c_void_p), ("fmax", C.c_void_p)] + \
                   [("npair", C.c_int32), ("pad2", C.c_int32)] + \
                   [(n, C.c_void_p) for n in ("pair_ba", "pair_bb", "pair_box", "pair_a0", "pair_a1", "pair_ra", "pair_b0", "pair_b1", "pair_rb")] + \
                   [("pair_k", r), ("pair_sides", C.c_void_p)]
    return OrHand


PAIR_K = 2.0e4       # N/m: the engine's default stiffness of the compliant hand-to-hand pairs (csrc/mi_engine.hip: hv.pair_k for the ShadowHand)


def segment_closest(a0, a1, b0, b1):
    """closest points of two segments (clamped construction, the order of oracle/physics.c seg_seg_closest)"""
    d1, d2, rr = a1 - a0, b1 - b0, a0 - b0
    A, E, F, Cc, B = d1 @ d1, d2 @ d2, d2 @ rr, d1 @ rr, d1 @ d2
    eps, den = 1e-12, A * E - B * B
    s = min(max((B * F - Cc * E) / den, 0.0), 1.0) if (den > eps and A > eps) else 0.0
    t = (B * s + F) / E if E > eps else 0.0
    tc = min(max(t, 0.0), 1.0)
    if (t != tc or not E > eps) and A > eps:
        s = min(max((B * tc - Cc) / A, 0.0), 1.0)
    return a0 + d1 * s, b0 + d2 * tc


def sphere_box3(c_local, r, half):
    """sphere_box for a box with three half sizes"""
    half = np.asarray(half, float)
    d = c_local - np.clip(c_local, -half, half)
    nd = np.linalg.norm(d)
    if nd > 1e-12:
        return nd - r, d / nd
    pen = half - np.abs(c_local)
    i = int(np.argmin(pen))
    n = np.zeros(3)
    n[i] = 1.0 if c_local[i] >= 0 else -1.0
    return -pen[i] - r, n


class OracleHandEngine:
    def __init__(self, spec, extras, num_envs, sim: dict, sensor_bodies, obj=None, backend="c", solver="gs", blocks=None):
        """obj: None = the 5 cm cube; dict(shape="egg" | "pen", dims=semi-axes | (radius, half length), mass=, inertia=principal inertias).
        backend: "c" -- oracle/hand.c (OpenMP over the envs); "numpy" -- the restatement below (solver "gs" only; the cross-check of
        hand.c).  solver: "gs" -- one Gauss-Seidel sequence, KMAX contacts per env (the single-wave kernel's order); "blocks" -- the
        finger-per-wave kernel's order with `blocks` = isaacgymenvs_amd.assets.model.hand_solver_blocks(spec) (hand.c header)."""
        # a block of another size / mass than the ShadowHand's 5 cm cube (AllegroHand: 6.5 cm, density 400): dict(shape="block", half=, mass=)
        self.cube_half, self.cube_mass = CUBE_HALF, CUBE_MASS
        if obj is not None and obj.get("shape") == "block":
            self.cube_half, self.cube_mass = float(obj["half"]), float(obj["mass"])
            obj = None
        self.cube_inertia = self.cube_mass * (2.0 * self.cube_half) ** 2 / 6.0          # isotropic
        self.objp = obj
        self.spec, self.ex, self.N = spec, extras, num_envs
        assert backend in ("c", "numpy") and solver in ("gs", "blocks") and (solver == "gs" or backend == "c")
        self.backend, self.solver = backend, solver
        self.eng = OracleEngine(spec, num_envs, params=dict(sim, gravity=(0.0, 0.0, 0.0)), sensor_bodies=sensor_bodies, precision="f64",
                                solver=solver, blocks=blocks)
        self.blocks = blocks
        self.env_mu = None                                # [N] per-env hand-object friction (negative: the default 1.0)
        self.kmax = KMAX                                  # contacts per env of solver "gs" (set before the first step; backend "c")
        self.sim = sim
        self.nd = spec.nd
        self.kp = np.array(extras["dof_kp"], float)
        # drive force limits (MJCF forcerange, shared.xml:250-269; allegro_hand.py:264 effort 0.5): clamp of the implicit PD force, solved with
        # the rows (physics.c OrDriveClamp).  None: unclamped drives.  Set before the first step.
        self.force_limit = np.array(extras["dof_force_limit"], float) if "dof_force_limit" in extras else None
        self.os_body = np.array(extras["os_body"]); self.os_pos = np.array(extras["os_pos"], float); self.os_rad = np.array(extras["os_rad"], float)
        self.sens = list(sensor_bodies)
        N, nd = num_envs, self.nd
        self.eng.root[:, :3] = [0.0, 0.0, 0.5]
        self.eng.root[:, 3:7] = extras["mount_quat"]
        self.targets = np.zeros((N, nd))
        self.obj = np.zeros((N, 13)); self.obj[:, 6] = 1.0
        self.sensor = np.zeros((N, 6 * len(self.sens)))
        self.dof_force = np.zeros((N, nd))
        self.ncontacts = np.zeros(N, int)
        self.obj_force = np.zeros((N, 3))                 # world-frame external force on the cube for the current step
        self.lo = np.minimum(spec.dof_lower, spec.dof_upper); self.up = np.maximum(spec.dof_lower, spec.dof_upper)
        # per-env `actor_params` factors (reference ShadowHand.yaml:104-159; columns as csrc/core/hand_engine.hpp HS_*): hand link masses,
        # joint damping, drive stiffness, tendon limit stiffness, tendon damping, object mass (+ inertia), object size
        self.scale = np.ones((num_envs, 8))
        self.limit_shift = np.zeros((num_envs, 2 * spec.nd))       # dof_properties.lower / upper: shifts of the lower, then the upper limits
        # the asset's explicit hand-to-hand contact pairs (shared.xml:31-51; extras["pairs"]) as compliant contacts of stiffness pair_k
        # (hand.c h_pairs; 0 = off).  Set before the first step.
        self.pairs = list(extras.get("pairs", []))
        self.pair_k = PAIR_K if self.pairs else 0.0
        self.pair_sides = np.zeros(num_envs, np.int32)             # pair sides pushed in the last sub-step

    # views on the wrapped engine's state
    @property
    def q(self): return self.eng.q
    @property
    def qd(self): return self.eng.qd
    @property
    def laml(self): return self.eng.lam[:, :]     # nsph = 0 => lam is just the limit impulses

    def _poses(self, e):
        _, _, bp = self.eng.energy(e, poses=True)
        return bp

    def fingertip_states(self):
        """[N, 5, 13] world pos, quat xyzw, linvel, angvel of the sensor (fingertip) bodies."""
        from isaacgymenvs_amd.assets.model import mat_to_quat
        out = np.zeros((self.N, len(self.sens), 13))
        if self.backend == "c":                      # one OpenMP pass over the envs (oracle/hand.c or_hand_fingertips)
            if not hasattr(self, "_hlib"):
                self._c_setup()
            st = self.eng.state
            assert st.flags.c_contiguous
            self._hlib.or_hand_fingertips(C.byref(self.eng.model), self.N, _ptr(st), _ptr(out))
            return out
        v6 = np.zeros(6)
        for e in range(self.N):
            bp = self._poses(e)
            s = np.ascontiguousarray(self.eng.state[e])
            for k, b in enumerate(self.sens):
                self.eng.lib.or_body_vel(C.byref(self.eng.model), _ptr(s), b, _ptr(v6))
                p = bp[b, 0:3]; R = bp[b, 3:12].reshape(3, 3)
                r = p - self.eng.root[e, :3]
                out[e, k, 0:3] = p; out[e, k, 3:7] = mat_to_quat(R)
                out[e, k, 7:10] = v6[3:6] + np.cross(v6[0:3], r); out[e, k, 10:13] = v6[0:3]
        return out

    def _c_setup(self):
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        out = os.path.join(here, "_build", "liboracle_hand_f64.so")
        srcs = [os.path.join(here, "hand.c"), os.path.join(here, "physics.c")]
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-s", "-C", here, "_build/liboracle_hand_f64.so"])
        self._hlib = C.CDLL(out)
        OrHand = _hand_struct()
        ex, spec = self.ex, self.spec
        k = self._hkeep = {}
        k["os_body"] = np.ascontiguousarray(self.os_body, np.int32)
        k["os_pos"] = np.ascontiguousarray(self.os_pos, np.float64); k["os_rad"] = np.ascontiguousarray(self.os_rad, np.float64)
        tends = ex["tendons"]
        k["tend_d0"] = np.ascontiguousarray([t["dof"][0] for t in tends], np.int32); k["tend_d1"] = np.ascontiguousarray([t["dof"][1] for t in tends], np.int32)
        k["tend_c0"] = np.ascontiguousarray([t["coef"][0] for t in tends], np.float64); k["tend_c1"] = np.ascontiguousarray([t["coef"][1] for t in tends], np.float64)
        k["tend_lo"] = np.ascontiguousarray([t["range"][0] for t in tends], np.float64); k["tend_hi"] = np.ascontiguousarray([t["range"][1] for t in tends], np.float64)
        k["kp"] = np.ascontiguousarray(self.kp, np.float64)
        k["fmax"] = None if self.force_limit is None else np.ascontiguousarray(self.force_limit, np.float64)
        hd = OrHand(nos=len(self.os_body), ntend=len(tends), kmax=int(self.kmax), body_cap=BODY_CAP, solver=1 if self.solver == "blocks" else 0)
        for n in ("os_body", "os_pos", "os_rad", "tend_d0", "tend_d1", "tend_c0", "tend_c1", "tend_lo", "tend_hi", "kp"):
            setattr(hd, n, _ptr(k[n]))
        hd.tend_stiffness, hd.tend_damping, hd.mu = float(ex["tendon_limit_stiffness"]), float(ex["tendon_damping"]), 1.0
        hd.fmax = _ptr(k["fmax"]) if k["fmax"] is not None else None
        if self.objp is None:
            hd.shape, hd.obj_mass = 0, self.cube_mass
            hd.obj_inertia[:] = [self.cube_inertia] * 3; hd.obj_dims[:] = [self.cube_half, 0.0, 0.0]
        else:
            hd.shape, hd.obj_mass = {"pen": 1, "egg": 2}[self.objp["shape"]], float(self.objp["mass"])
            hd.obj_inertia[:] = [float(x) for x in self.objp["inertia"]]
            dims = [float(x) for x in self.objp["dims"]]
            hd.obj_dims[:] = (dims + [0.0, 0.0, 0.0])[:3]
        if self.solver == "blocks":
            k["limb_of_body"] = np.ascontiguousarray(self.blocks["limb_of_body"], np.int32)
            k["limb_cap"] = np.ascontiguousarray(self.blocks["limb_cap"], np.int32)
            hd.nlimb = len(k["limb_cap"])
            hd.limb_of_body, hd.limb_cap = _ptr(k["limb_of_body"]), _ptr(k["limb_cap"])
        pr = self.pairs
        k["pair_ba"] = np.ascontiguousarray([q["a"]["body"] for q in pr], np.int32); k["pair_bb"] = np.ascontiguousarray([q["b"]["body"] for q in pr], np.int32)
        k["pair_box"] = np.ascontiguousarray([int(q["a"]["kind"] == "box") for q in pr], np.int32)
        for side in "ab":
            k["pair_%s0" % side] = np.ascontiguousarray([q[side]["p0"] for q in pr], np.float64).reshape(-1)
            k["pair_%s1" % side] = np.ascontiguousarray([q[side]["p1"] for q in pr], np.float64).reshape(-1)
            k["pair_r%s" % side] = np.ascontiguousarray([q[side]["r"] for q in pr], np.float64)
        hd.npair, hd.pair_k = len(pr), float(self.pair_k)
        for n in ("pair_ba", "pair_bb", "pair_box", "pair_a0", "pair_a1", "pair_ra", "pair_b0", "pair_b1", "pair_rb"):
            setattr(hd, n, _ptr(k[n]) if len(pr) else None)
        hd.pair_sides = _ptr(self.pair_sides)
        self._hd = hd
        self._nc32 = np.zeros(self.N, np.int32)
        self.limb_counts = np.zeros((self.N, max(int(hd.nlimb), 1)), np.int32)     # solver "blocks": contacts kept per limb, last sub-step

    def step(self):
        P = self.sim
        if self.backend == "c":
            if not hasattr(self, "_hlib"):
                self._c_setup()
            e = self.eng
            e.set_params(**dict(P, gravity=tuple(P["gravity"])))     # (the object's gravity; the hand's is switched off inside hand.c)
            st = e.state
            assert st.shape[1] == 13 + 3 * self.nd and st.flags.c_contiguous
            self._hd.pair_k = float(self.pair_k)
            mu = None if self.env_mu is None else np.ascontiguousarray(self.env_mu, np.float64)
            tg = np.ascontiguousarray(self.targets, np.float64); fo = np.ascontiguousarray(self.obj_force, np.float64)
            sc = np.ascontiguousarray(self.scale, np.float64); ls = np.ascontiguousarray(self.limit_shift, np.float64)
            self._hlib.or_hand_step(C.byref(e.model), C.byref(e.params), C.byref(self._hd), self.N, _ptr(st), _ptr(self.obj), _ptr(tg), _ptr(fo),
                                    _ptr(sc), _ptr(ls), _ptr(mu) if mu is not None else None, _ptr(self.sensor), _ptr(self.dof_force),
                                    _ptr(self._nc32), _ptr(self.limb_counts) if self.solver == "blocks" else None)
            self.ncontacts[:] = self._nc32
            return
        h = P["dt"] / P["substeps"]
        for _ in range(P["substeps"]):
            for e in range(self.N):
                self._substep_env(e, h)

    def _substep_env(self, e, h):
        P, nd, spec, ex = self.sim, self.nd, self.spec, self.ex
        q, qd, tgt = self.q[e].copy(), self.qd[e].copy(), self.targets[e]
        M, bias = self.eng.dynamics(e)
        s_mass, s_damp, s_kp, s_tk, s_td, s_om, s_os = self.scale[e, :7]
        M, bias = M * s_mass, bias * s_mass
        D = np.array(spec.dof_damping, float) * s_damp
        kp = self.kp * s_kp
        Mh = M + np.diag(np.array(spec.dof_armature, float) + h * D + h * h * kp)
        rhs = -bias - kp * (q - tgt) - (D + h * kp) * qd
        for t in ex["tendons"]:                                   # soft two-sided limit on the tendon length
            d0, d1 = t["dof"]; c0, c1 = t["coef"]; lo, hi = t["range"]
            Lt = c0 * q[d0] + c1 * q[d1]
            Ld = c0 * qd[d0] + c1 * qd[d1]
            viol = Lt - min(max(Lt, lo), hi)
            k = ex["tendon_limit_stiffness"] * s_tk if viol != 0.0 else 0.0
            dmp = ex["tendon_damping"] * s_td
            cvec = np.zeros(nd); cvec[d0] = c0; cvec[d1] = c1
            Mh += (h * dmp + h * h * k) * np.outer(cvec, cvec)
            rhs -= cvec * (k * viol + (dmp + h * k) * Ld)
        bp = self._poses(e)
        O = self.eng.root[e, :3]
        s_state = np.ascontiguousarray(self.eng.state[e])
        J3 = np.zeros((3, nd))
        # the asset's hand-to-hand contact pairs as compliant contacts (hand.c h_pairs): each side of an overlapping pair is pushed along its
        # outward direction by the implicit spring k (pen - h J qd+)
        sides = 0
        pair_sens = {}       # per force-sensor body with pushed pair sides: A = sum k pen [pc x u; u] about O, P = sum pen, n = sides (hand.c PairSens)
        for pr in (self.pairs if self.pair_k > 0 else []):
            ba, bb = pr["a"]["body"], pr["b"]["body"]
            Ra, ra_, Rb_, rb_ = bp[ba, 3:12].reshape(3, 3), bp[ba, 0:3], bp[bb, 3:12].reshape(3, 3), bp[bb, 0:3]
            b0, b1 = rb_ + Rb_ @ np.array(pr["b"]["p0"]), rb_ + Rb_ @ np.array(pr["b"]["p1"])
            if pr["a"]["kind"] == "box":
                best = None
                for t_ in (0.0, 0.5, 1.0):
                    cw = b0 + t_ * (b1 - b0)
                    ds, nl = sphere_box3(Ra.T @ (cw - ra_) - np.array(pr["a"]["p0"]), pr["b"]["r"], pr["a"]["p1"])
                    if best is None or ds < best[0]:
                        nw = Ra @ nl
                        best = (ds, -nw, cw - nw * (pr["b"]["r"] + 0.5 * ds))
                dist, n, pc = best
            else:
                ca, cb = segment_closest(ra_ + Ra @ np.array(pr["a"]["p0"]), ra_ + Ra @ np.array(pr["a"]["p1"]), b0, b1)
                dv = ca - cb
                d = np.linalg.norm(dv)
                n = dv / d if d > 1e-9 else np.array([0.0, 0.0, 1.0])
                dist = d - pr["a"]["r"] - pr["b"]["r"]
                pc = cb + n * (pr["b"]["r"] + 0.5 * dist)
            pen = -dist
            if not pen > 0:
                continue
            self.pair_pen_max = max(getattr(self, "pair_pen_max", 0.0), pen)      # (numpy backend: the deepest overlap met so far, a diagnostic)
            for body, u in ((ba, n), (bb, -n)):
                self.eng.lib.or_point_jac(C.byref(self.eng.model), _ptr(s_state), body, _ptr(np.ascontiguousarray(pc - O)), _ptr(J3))
                Js = u @ J3
                Mh += h * h * self.pair_k * np.outer(Js, Js)
                rhs += Js * self.pair_k * (pen - h * (Js @ qd))
                if body in self.sens:
                    ps = pair_sens.setdefault(self.sens.index(body), dict(A=np.zeros(6), P=0.0, n=0))
                    ps["A"] += self.pair_k * pen * np.concatenate([np.cross(pc - O, u), u])
                    ps["P"] += pen; ps["n"] += 1
                sides += 1
        self.pair_sides[e] = sides
        Minv = np.linalg.inv(Mh)
        v = qd + h * (Minv @ rhs)
        g = np.array(P["gravity"], float)
        xo, qo = self.obj[e, 0:3].copy(), self.obj[e, 3:7].copy()
        egg = self.objp is not None
        omass = (float(self.objp["mass"]) if egg else self.cube_mass) * s_om
        vo = self.obj[e, 7:10] + h * (g + self.obj_force[e] / omass)   # + apply_rigid_body_force_tensors on the object
        wo = self.obj[e, 10:13].copy()
        Ro = quat2mat(qo)
        # ---- rows
        rows = []   # (Jh[nd], Jo[6], vt, kind, idx)
        lim_sign = {}
        for d in range(nd):
            if not spec.dof_limited[d]:
                self.laml[e, d] = 0.0
                continue
            dl, du = q[d] - (self.lo[d] + self.limit_shift[e, d]), (self.up[d] + self.limit_shift[e, nd + d]) - q[d]
            Cc, s = (dl, 1.0) if dl < du else (du, -1.0)
            lw = self.laml[e, d]
            l0 = (0.0 if lw * s < 0 else abs(lw)) * P["warm"]
            Jh = np.zeros(nd); Jh[d] = s
            vt = -Cc / h if Cc >= 0 else min(-Cc * P["erp"] / h, P["max_depen_vel"])
            fm = float(self.force_limit[d]) if (self.force_limit is not None and kp[d] > 0) else 0.0
            rows.append(dict(Jh=Jh, Jo=np.zeros(6), vt=vt, lam=l0, kind="lim", d=d, s=s, fmax=fm, fa=-kp[d] * (q[d] - tgt[d]), c=D[d] + h * kp[d], rho=0.0))
        ncon = 0
        contacts = []
        per_body = {}
        for si in range(len(self.os_body)):
            b = int(self.os_body[si])
            c = bp[b, 0:3] + bp[b, 3:12].reshape(3, 3) @ self.os_pos[si]      # world
            if egg and self.objp["shape"] == "pen":
                dist, nl = sphere_capsule(Ro.T @ (c - xo), self.os_rad[si], self.objp["dims"][0] * s_os, self.objp["dims"][1] * s_os)
            elif egg:
                dist, nl = sphere_ellipsoid(Ro.T @ (c - xo), self.os_rad[si], np.asarray(self.objp["dims"], float) * s_os)
            else:
                dist, nl = sphere_box(Ro.T @ (c - xo), self.os_rad[si], self.cube_half * s_os)
            if dist >= P["contact_offset"] or ncon >= KMAX or per_body.get(b, 0) >= BODY_CAP:
                continue
            per_body[b] = per_body.get(b, 0) + 1
            n = Ro @ nl                                                   # from the cube towards the sphere
            t1, t2 = contact_frame(n)
            pc = c - self.os_rad[si] * n                                  # contact point, world
            self.eng.lib.or_point_jac(C.byref(self.eng.model), _ptr(s_state), b, _ptr(np.ascontiguousarray(pc - O)), _ptr(J3))
            rc = pc - xo
            gap = dist - P["rest_offset"]
            vtn = -gap / h if gap >= 0 else min(-gap * P["erp"] / h, P["max_depen_vel"])
            for k, u in enumerate((n, t1, t2)):
                Jo = -np.concatenate([u, np.cross(rc, u)])
                rows.append(dict(Jh=u @ J3, Jo=Jo, vt=vtn if k == 0 else 0.0, lam=0.0, kind="con", k=k, c=ncon))
            contacts.append(dict(b=b, pc=pc, n=n, t1=t1, t2=t2, row0=len(rows) - 3))
            ncon += 1
        self.ncontacts[e] = ncon
        Moinv = np.zeros((6, 6))
        Moinv[:3, :3] = np.eye(3) / omass
        # world-frame inverse inertia: Ro diag(1 / I) Ro^T for the ellipsoid's principal inertias, a multiple of identity for the cube
        Moinv[3:, 3:] = (Ro @ np.diag(1.0 / np.asarray(self.objp["inertia"], float)) @ Ro.T if egg else np.eye(3) / self.cube_inertia) / s_om
        for r in rows:
            r["Bh"] = Minv @ r["Jh"]; r["Bo"] = Moinv @ r["Jo"]
            r["Ainv"] = 1.0 / (P["cfm"] + r["Jh"] @ r["Bh"] + r["Jo"] @ r["Bo"])
            if r["lam"] != 0.0:
                v += r["Bh"] * r["lam"]
        vobj = np.concatenate([vo, wo])
        mu = 0.5 * (1.0 + 1.0)                                            # hand geom friction 1 (shared.xml:12), cube default 1
        for _ in range(P["iters"]):
            i = 0
            while i < len(rows):
                r = rows[i]
                if r["kind"] == "lim":
                    if r["fmax"] > 0:          # the dof's drive clamp, ahead of its limit row (physics.c OrDriveClamp / drive_clamp_update)
                        a = 0.0 if r["lam"] > 0 else 1.0 / r["Ainv"] - P["cfm"]      # held by its limit: the velocity does not answer to rho
                        kk = max(1.0 / h - r["c"] * a, 0.1 / h)
                        Ff = r["fa"] - r["c"] * v[r["d"]] + r["rho"] * r["c"] * a
                        rn_ = (r["fmax"] - Ff) / kk if Ff > r["fmax"] else ((-r["fmax"] - Ff) / kk if Ff < -r["fmax"] else 0.0)
                        dr = rn_ - r["rho"]; r["rho"] = rn_
                        v += r["Bh"] * (r["s"] * dr)
                    vn = r["Jh"] @ v
                    nl_ = max(r["lam"] - (vn - r["vt"]) * r["Ainv"], 0.0)
                    dl = nl_ - r["lam"]; r["lam"] = nl_
                    v += r["Bh"] * dl
                    i += 1
                else:
                    rn, ra, rb = rows[i], rows[i + 1], rows[i + 2]
                    vn = rn["Jh"] @ v + rn["Jo"] @ vobj
                    ln = max(rn["lam"] - (vn - rn["vt"]) * rn["Ainv"], 0.0)
                    dl = ln - rn["lam"]; rn["lam"] = ln
                    v += rn["Bh"] * dl; vobj += rn["Bo"] * dl
                    # both tangent rows from the SAME velocity, then the disc projection, one application (round 6: solving and applying t1
                    # before t2 is looked at let a fast-sliding contact's friction point off the sliding direction)
                    vtan = [rt["Jh"] @ v + rt["Jo"] @ vobj for rt in (ra, rb)]
                    lt = friction_step([rt["lam"] - vt_ * rt["Ainv"] for rt, vt_ in zip((ra, rb), vtan)], (ra["lam"], rb["lam"]), vtan,
                                       (ra["Ainv"], rb["Ainv"]), mu * ln)
                    for rt, nl_ in zip((ra, rb), lt):
                        dl = nl_ - rt["lam"]; rt["lam"] = nl_
                        v += rt["Bh"] * dl; vobj += rt["Bo"] * dl
                    i += 3
        # ---- outputs
        ll, rho, lim = np.zeros(nd), np.zeros(nd), np.zeros(nd, bool)
        for r in rows:
            if r["kind"] == "lim":
                ll[r["d"]] = r["lam"] * r["s"]
                rho[r["d"]] = r["rho"]; lim[r["d"]] = r["fmax"] > 0
        self.laml[e] = ll
        # (a force-limited drive reports the end-of-step force the clamp acts on, fa - c v + rho / h)
        self.dof_force[e] = -kp * (q - tgt) - D * v + ll / h + np.where(lim, rho / h - h * kp * v, 0.0)
        sens = np.zeros(6 * len(self.sens))
        for cdat in contacts:
            if cdat["b"] in self.sens:
                k = self.sens.index(cdat["b"])
                r0 = cdat["row0"]
                f = (cdat["n"] * rows[r0]["lam"] + cdat["t1"] * rows[r0 + 1]["lam"] + cdat["t2"] * rows[r0 + 2]["lam"]) / h
                Rb = bp[cdat["b"], 3:12].reshape(3, 3); pb = bp[cdat["b"], 0:3]
                sens[6 * k:6 * k + 3] += Rb.T @ f
                sens[6 * k + 3:6 * k + 6] += Rb.T @ np.cross(cdat["pc"] - pb, f)
        for k, ps in pair_sens.items():      # the hand's own contacts on the fingertips: wrench = A (1 - n h (A . V) / (k P^2)), V the tip's twist about O
            b = self.sens[k]
            Jp = np.zeros((3, nd))
            def vel(p):
                self.eng.lib.or_point_jac(C.byref(self.eng.model), _ptr(s_state), b, _ptr(np.ascontiguousarray(p, dtype=np.float64)), _ptr(Jp))
                return Jp @ v
            vO, vx, vy = vel([0.0, 0.0, 0.0]), vel([1.0, 0.0, 0.0]), vel([0.0, 1.0, 0.0])
            om = np.array([vy[2] - vO[2], vO[2] - vx[2], vx[1] - vO[1]])
            fac = 1.0 - ps["n"] * h * (ps["A"][:3] @ om + ps["A"][3:] @ vO) / (self.pair_k * ps["P"] ** 2)
            Rb = bp[b, 3:12].reshape(3, 3); pb = bp[b, 0:3]
            f = ps["A"][3:] * fac
            tq = ps["A"][:3] * fac - np.cross(pb - O, f)
            sens[6 * k:6 * k + 3] += Rb.T @ f
            sens[6 * k + 3:6 * k + 6] += Rb.T @ tq
        self.sensor[e] = sens
        # ---- integrate
        self.qd[e] = v; self.q[e] = q + h * v
        self.obj[e, 7:10] = vobj[:3]; self.obj[e, 10:13] = vobj[3:]
        self.obj[e, 0:3] = xo + h * vobj[:3]
        om = vobj[3:]; an = np.linalg.norm(om); th = an * h
        if th > 1e-12:
            dq = np.concatenate([om * np.sin(th / 2) / an, [np.cos(th / 2)]])
        else:
            dq = np.concatenate([om * h / 2, [1.0]])
        Q = qo
        x = dq[3] * Q[0] + dq[0] * Q[3] + dq[1] * Q[2] - dq[2] * Q[1]
        y = dq[3] * Q[1] - dq[0] * Q[2] + dq[1] * Q[3] + dq[2] * Q[0]
        z = dq[3] * Q[2] + dq[0] * Q[1] - dq[1] * Q[0] + dq[2] * Q[3]
        w = dq[3] * Q[3] - dq[0] * Q[0] - dq[1] * Q[1] - dq[2] * Q[2]
        qn = np.array([x, y, z, w]); self.obj[e, 3:7] = qn / np.linalg.norm(qn)

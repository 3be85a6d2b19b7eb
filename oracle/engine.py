"""ctypes front-end of oracle/physics.c (see that file's header for scope and pinning)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    out = os.path.join(_HERE, "_build", "liboracle_f64.so")
    src = os.path.join(_HERE, "physics.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return out


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _struct(real):
    class OrModel(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("nb", "nd", "fixed_base", "nsph", "nsens", "pad0")] + \
                   [(n, C.c_void_p) for n in (
                       "parent", "bpos", "bquat", "mass", "com", "inertia", "dof_body", "dof_type", "dof_axis",
                       "dof_anchor", "dof_lower", "dof_upper", "dof_limited", "dof_armature", "dof_damping",
                       "dof_stiffness", "dof_springref", "sph_body", "sph_pos", "sph_rad", "sph_mu", "sens_body")] + \
                   [(n, C.c_int32) for n in ("ncap", "npg", "ngp", "kmax", "kpair", "warm_slots")] + \
                   [(n, C.c_void_p) for n in ("cap_body", "cap_p0", "cap_p1", "cap_rad", "cap_mu", "gp_a", "gp_b", "pg_first", "pg_count")] + \
                   [(n, C.c_int32) for n in ("solver", "nblk", "pad1", "pad2")] + \
                   [("gi_group", C.c_void_p), ("body_block", C.c_void_p), ("kmax_blk", C.c_void_p)]

    class OrParams(C.Structure):
        _fields_ = [("dt", real), ("substeps", C.c_int32), ("iters", C.c_int32), ("gravity", real * 3),
                    ("contact_offset", real), ("rest_offset", real), ("max_depen_vel", real), ("erp", real),
                    ("plane_mu", real), ("ground_z", real), ("cfm", real), ("warm", real)]
    return OrModel, OrParams


DEFAULT_PARAMS = dict(dt=1.0 / 60.0, substeps=2, iters=4, gravity=(0.0, 0.0, -9.81), contact_offset=0.02,
                      rest_offset=0.0, max_depen_vel=10.0, erp=0.5, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)


class OracleEngine:
    """Batched CPU physics for one ModelSpec.  State is AoS per env:
    root[13] (pos3, quat xyzw4, linvel3, angvel3) | q[nd] | qd[nd] | lam_c[3*nsph] | lam_l[nd]."""

    def __init__(self, spec, num_envs, params=None, sensor_bodies=(), precision="f64", selfcol=None, kmax=0, kpair=0, warm_slots=0,
                 solver="gs", blocks=None):
        """selfcol: self-collision tables (isaacgymenvs_amd.assets.model.self_collision_tables) or None; kmax / kpair: caps of
        the ground / self contacts per env (0 = unlimited; the engine's LDS contact store holds 12 + 3).
        solver: "gs" -- one Gauss-Seidel sequence over all rows (the single-wave kernels' order); "blocks" -- the limb-per-wave
        kernels' order (physics.c OrModel.solver = 1), with `blocks` = isaacgymenvs_amd.assets.model.solver_blocks(spec, ...)."""
        build()
        self.spec = spec
        self.np_real = np.float64 if precision == "f64" else np.float32
        creal = C.c_double if precision == "f64" else C.c_float
        self.lib = C.CDLL(os.path.join(_HERE, "_build", f"liboracle_{precision}.so"))
        assert self.lib.or_sizeof_real() == np.dtype(self.np_real).itemsize
        OrModel, OrParams = _struct(creal)
        r = self.np_real
        self._keep = k = {}
        k["parent"] = np.ascontiguousarray(spec.parent, np.int32)
        for n in ("bpos", "bquat", "mass", "com", "inertia", "dof_axis", "dof_anchor", "dof_lower", "dof_upper",
                  "dof_armature", "dof_damping", "dof_stiffness", "dof_springref", "sph_pos", "sph_rad"):
            k[n] = np.ascontiguousarray(getattr(spec, n), r)
        k["sph_mu"] = np.ascontiguousarray(spec.sph_friction, r)
        for n in ("dof_body", "dof_type", "dof_limited", "sph_body"):
            k[n] = np.ascontiguousarray(getattr(spec, n), np.int32)
        k["sens_body"] = np.ascontiguousarray(np.array(sensor_bodies, np.int32))
        m = OrModel(nb=spec.nb, nd=spec.nd, fixed_base=int(spec.fixed_base), nsph=len(spec.sph_body),
                    nsens=len(sensor_bodies), pad0=0)
        for n, _ in OrModel._fields_[6:28]:
            setattr(m, n, _ptr(k[n]))
        self.npg = 0
        if selfcol and selfcol.get("groups"):
            first, count, ga, gb = [], [], [], []
            for g in selfcol["groups"]:
                first.append(len(ga)); count.append(len(g["pairs"]))
                for i, j in g["pairs"]:
                    ga.append(i); gb.append(j)
            k["cap_body"] = np.ascontiguousarray(selfcol["cap_body"], np.int32)
            for n in ("cap_p0", "cap_p1", "cap_rad", "cap_mu"):
                k[n] = np.ascontiguousarray(selfcol[n], r)
            k["gp_a"], k["gp_b"] = np.ascontiguousarray(ga, np.int32), np.ascontiguousarray(gb, np.int32)
            k["pg_first"], k["pg_count"] = np.ascontiguousarray(first, np.int32), np.ascontiguousarray(count, np.int32)
            m.ncap, m.npg, m.ngp = len(k["cap_body"]), len(first), len(ga)
            for n in ("cap_body", "cap_p0", "cap_p1", "cap_rad", "cap_mu", "gp_a", "gp_b", "pg_first", "pg_count"):
                setattr(m, n, _ptr(k[n]))
            self.npg = len(first)
            self.pair_list = list(zip(ga, gb))
        m.kmax, m.kpair, m.warm_slots = int(kmax), int(kpair), int(warm_slots)
        self.solver = solver
        if solver == "blocks":
            assert blocks is not None, "solver='blocks' needs the model's block tables"
            k["gi_group"] = np.ascontiguousarray(blocks["gi_group"], np.int32)
            k["body_block"] = np.ascontiguousarray(blocks["body_block"], np.int32)
            m.solver, m.nblk = 1, int(blocks["nblk"])
            if blocks.get("kmax_blk") is not None:        # per-block caps of the ground contacts (compact-store limb waves)
                k["kmax_blk"] = np.ascontiguousarray(list(blocks["kmax_blk"]) + [0] * (m.nblk - len(blocks["kmax_blk"])), np.int32)
                m.kmax_blk = _ptr(k["kmax_blk"])
            m.gi_group, m.body_block = _ptr(k["gi_group"]), _ptr(k["body_block"])
        else:
            assert solver == "gs"
        self.model = m
        self.set_params(**(params or {}))
        self.N = num_envs
        self.nd, self.nsph, self.nsens = spec.nd, len(spec.sph_body), len(sensor_bodies)
        self.ss = self.lib.or_state_size(C.byref(m))
        self.os = self.lib.or_out_size(C.byref(m))
        self.state = np.zeros((num_envs, self.ss), r)
        self.state[:, 6] = 1.0
        self.out = np.zeros((num_envs, self.os), r)

    def set_params(self, **kw):
        _, OrParams = _struct(C.c_double if self.np_real == np.float64 else C.c_float)
        d = dict(DEFAULT_PARAMS)
        d.update(kw)
        self.params_dict = d
        p = OrParams()
        for key, val in d.items():
            if key == "gravity":
                for i in range(3):
                    p.gravity[i] = val[i]
            else:
                setattr(p, key, val)
        self.params = p

    # ---- views
    @property
    def root(self):
        return self.state[:, :13]

    @property
    def q(self):
        return self.state[:, 13:13 + self.nd]

    @property
    def qd(self):
        return self.state[:, 13 + self.nd:13 + 2 * self.nd]

    @property
    def lam(self):
        """warm-start impulses of the ground contacts [3 nsph] and the joint limits [nd]"""
        return self.state[:, 13 + 2 * self.nd:13 + 3 * self.nd + 3 * self.nsph]

    @property
    def lam_pair(self):
        """warm-start impulses (normal, two tangents) of the self-collision groups [npg, 3]"""
        return self.state[:, 13 + 3 * self.nd + 3 * self.nsph:].reshape(self.N, self.npg, 3)

    @property
    def pair_info(self):
        """per self-collision group: world force on side a (3), selected capsule-pair index or -1, its signed distance,
        number of contacts dropped by the kmax / kpair caps in this env, contact point relative to the root origin (3)"""
        return self.out[:, 6 * self.nsens + self.nd + 3 * self.nsph:].reshape(self.N, self.npg, 9)

    @property
    def sensor(self):
        return self.out[:, :6 * self.nsens]

    @property
    def dof_force(self):
        return self.out[:, 6 * self.nsens:6 * self.nsens + self.nd]

    @property
    def sph_force(self):
        return self.out[:, 6 * self.nsens + self.nd:6 * self.nsens + self.nd + 3 * self.nsph].reshape(self.N, self.nsph, 3)

    def set_ground(self, height_samples, hscale, vscale, border, slope_threshold=0.0, walls=True):
        """Height field (int16 [rows, cols], reference Terrain.height_field_raw) instead of the z = ground_z plane.
        walls: with the slope correction on, the risers it creates collide from the side (physics.c ground_contact)."""
        creal = C.c_double if self.np_real == np.float64 else C.c_float

        class OrGround(C.Structure):
            _fields_ = [("hs", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("hscale", creal), ("vscale", creal),
                        ("border", creal), ("thr", creal), ("walls", C.c_int32)]
        self._hs = np.ascontiguousarray(height_samples, np.int16)
        # slope_threshold: terrain.slopeTreshold of the mesh generator (anymal_terrain.py:576); 0 = the uncorrected mesh
        thr = slope_threshold * hscale / vscale if slope_threshold and slope_threshold > 0 else 0.0
        self.ground = OrGround(_ptr(self._hs), self._hs.shape[0], self._hs.shape[1], hscale, vscale, border, thr, 1 if walls else 0)

    def step(self, tau, env_mu=None):
        tau = np.ascontiguousarray(tau, self.np_real).reshape(self.N, self.nd)
        gnd = getattr(self, "ground", None)
        if gnd is None and env_mu is None and not getattr(self, "want_netf", False):
            self.lib.or_step(C.byref(self.model), C.byref(self.params), self.N, _ptr(self.state), _ptr(tau), _ptr(self.out))
            return
        if not hasattr(self, "netf"):
            self.netf = np.zeros((self.N, self.spec.nb, 3), self.np_real)
        mu = None if env_mu is None else np.ascontiguousarray(env_mu, self.np_real)
        self.lib.or_step_ex(C.byref(self.model), C.byref(self.params), C.byref(gnd) if gnd is not None else None,
                            _ptr(mu) if mu is not None else None, self.N, _ptr(self.state), _ptr(tau), _ptr(self.out),
                            _ptr(self.netf))

    def step_drive(self, tau, kp, kd, target=None, fext_local=None):
        """One step with implicit PD position drives (DOF_MODE_POS: stiffness kp, damping kd, targets [N, nd]) and external
        forces at the bodies' centres of mass in their local frames ([N, nb, 3]); plane ground, model friction."""
        r = self.np_real
        creal = C.c_double if r == np.float64 else C.c_float
        tau = np.ascontiguousarray(tau, r).reshape(self.N, self.nd)
        tg = None if target is None else np.ascontiguousarray(target, r).reshape(self.N, self.nd)
        fx = None if fext_local is None else np.ascontiguousarray(fext_local, r).reshape(self.N, self.spec.nb * 3)
        self.lib.or_step_drive(C.byref(self.model), C.byref(self.params), self.N, _ptr(self.state), _ptr(tau), _ptr(self.out),
                               creal(kp), creal(kd), _ptr(tg) if tg is not None else None, _ptr(fx) if fx is not None else None)

    def step_drive_v(self, tau, kp, kd, target):
        """as step_drive with per-dof gains kp[nd], kd[nd] (gym dof properties stiffness / damping of DOF_MODE_POS dofs); fills self.netf"""
        r = self.np_real
        tau = np.ascontiguousarray(tau, r).reshape(self.N, self.nd)
        tg = np.ascontiguousarray(target, r).reshape(self.N, self.nd)
        kp, kd = np.ascontiguousarray(kp, r).reshape(self.nd), np.ascontiguousarray(kd, r).reshape(self.nd)
        if not hasattr(self, "netf"):
            self.netf = np.zeros((self.N, self.spec.nb, 3), self.np_real)
        self.lib.or_step_drive_v(C.byref(self.model), C.byref(self.params), self.N, _ptr(self.state), _ptr(tau), _ptr(self.out),
                                 _ptr(kp), _ptr(kd), _ptr(tg), _ptr(self.netf))

    def dynamics(self, env=0):
        nv = self.spec.nv
        M = np.zeros((nv, nv), self.np_real)
        b = np.zeros(nv, self.np_real)
        s = np.ascontiguousarray(self.state[env])
        self.lib.or_dynamics(C.byref(self.model), C.byref(self.params), _ptr(s), _ptr(M), _ptr(b))
        return M, b

    def energy(self, env=0, poses=False):
        creal = C.c_double if self.np_real == np.float64 else C.c_float
        ke, pe = creal(), creal()
        s = np.ascontiguousarray(self.state[env])
        bp = np.zeros((self.spec.nb, 12), self.np_real)
        self.lib.or_energy(C.byref(self.model), C.byref(self.params), _ptr(s), C.byref(ke), C.byref(pe), _ptr(bp))
        return (ke.value, pe.value, bp) if poses else (ke.value, pe.value)

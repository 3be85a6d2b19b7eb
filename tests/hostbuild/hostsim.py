"""ctypes wrapper of the g++ host build of the specialised engine core (test-only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_OUT = os.path.join(_HERE, "..", "_hostbuild")


class SimParamsC(C.Structure):
    _fields_ = [("dt", C.c_float), ("substeps", C.c_int32), ("iters", C.c_int32), ("g", C.c_float * 3),
                ("contact_offset", C.c_float), ("rest_offset", C.c_float), ("max_depen_vel", C.c_float),
                ("erp", C.c_float), ("plane_mu", C.c_float), ("ground_z", C.c_float), ("cfm", C.c_float),
                ("warm", C.c_float)]


def make_params(d):
    p = SimParamsC()
    for k, v in d.items():
        if k == "gravity":
            for i in range(3):
                p.g[i] = v[i]
        else:
            setattr(p, k, v)
    return p


# model -> (compile-time switch, optimisation level): one library per model, compiled in parallel
_MODELS = {"cartpole": ("HOSTSIM_CARTPOLE", "-O2"), "ant": ("HOSTSIM_ANT", "-O2"), "anymal": ("HOSTSIM_ANYMAL", "-O2"),
           "quadcopter": ("HOSTSIM_QUADCOPTER", "-O2"), "humanoid": ("HOSTSIM_HUMANOID", "-O2"), "hand": ("HOSTSIM_HAND", "-O1"),
           "bbot": ("HOSTSIM_BBOT", "-O2")}
_ENTRY_MODEL = {"hs_step_mw_ant": "ant", "hs_step_mw_terrain": "anymal", "hs_step_selfcol": "humanoid", "hs_step_selfcol2": "humanoid", "hs_step_selfcol3": "humanoid", "hs_step_mwc": "humanoid", "hs_step_mwc_fused": "humanoid", "hs_step_terrain": "anymal", "hs_set_slope_threshold": "anymal", "hs_set_walls": "anymal", "hs_ground_contact": "anymal", "hs_step_drive": "quadcopter", "hs_step_hand": "hand", "hs_step_hand_mw": "hand", "hs_hand_fingertips": "hand", "hs_set_hand_pair_stiffness": "hand", "hs_hand_pair_sides": "hand", "hs_step_bbot": "bbot"}
_libs = {}


def _deps():
    from isaacgymenvs_amd.registry import generate_headers
    core = os.path.join(_HERE, "..", "..", "isaacgymenvs_amd", "csrc", "core")
    return generate_headers() + [os.path.join(_HERE, "hostsim.cpp"), os.path.join(core, "engine.hpp"), os.path.join(core, "engine_mw.hpp"), os.path.join(core, "engine_mwc.hpp"),
                                 os.path.join(core, "hand_engine.hpp"), os.path.join(core, "hand_engine_mw.hpp")]


def _build_models(names):
    """Compile the stale per-model libraries concurrently (g++ releases the GIL: plain threads suffice)."""
    from concurrent.futures import ThreadPoolExecutor
    deps = _deps()
    os.makedirs(_OUT, exist_ok=True)
    newest = max(os.path.getmtime(d) for d in deps)
    core = os.path.join(_HERE, "..", "..", "isaacgymenvs_amd", "csrc", "core")
    own = {"bbot": [os.path.join(core, "bbot_engine.hpp")]}      # headers only one model's library includes
    jobs = []
    for n in names:
        out = os.path.join(_OUT, f"libhostsim_{n}.so")
        if not os.path.exists(out) or os.path.getmtime(out) < max([newest] + [os.path.getmtime(d) for d in own.get(n, [])]):
            macro, opt = _MODELS[n]
            jobs.append(["g++", opt, "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-pthread", "-ffp-contract=off", "-D" + macro,
                         os.path.join(_HERE, "hostsim.cpp"), "-o", out])
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(subprocess.check_call, jobs))
    for n in names:
        if n not in _libs:
            _libs[n] = C.CDLL(os.path.join(_OUT, f"libhostsim_{n}.so"))


class _Router:
    """Looks like one library: hs_step(model, ...) goes to the library of `model`, the model-specific entry points to theirs."""

    def __getattr__(self, name):
        if name in _ENTRY_MODEL:
            return getattr(_libs[_ENTRY_MODEL[name]], name)
        if name == "hs_step":
            def call(model, *args):
                return _libs[model.decode()].hs_step(model, *args)
            return call
        raise AttributeError(name)


def build(humanoid=False):
    """All models a CPU test may ask for are (re)built in one parallel batch the first time any of them is needed."""
    _build_models(list(_MODELS))
    return _Router()


def build_hand():
    return build()


def step(lib, model, params, state, tau, out):
    rc = lib.hs_step(model.encode(), C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p),
                     tau.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0


def step_mw_ant(lib, params, state, tau, out):
    """Ant through the multi-wave sub-step (four role threads per env); same layouts as step()."""
    rc = lib.hs_step_mw_ant(C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p), tau.ctypes.data_as(C.c_void_p),
                            out.ctypes.data_as(C.c_void_p))
    assert rc == 0


def step_mw_terrain(lib, params, state, tau, out, hs, hscale, vscale, border, mu, netf):
    hs = np.ascontiguousarray(hs, np.int16)
    lib.hs_step_mw_terrain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    rc = lib.hs_step_mw_terrain(C.cast(C.byref(params), C.c_void_p), state.shape[0], state.ctypes.data_as(C.c_void_p),
                                tau.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), hs.ctypes.data_as(C.c_void_p),
                                hs.shape[0], hs.shape[1], hscale, vscale, border, mu.ctypes.data_as(C.c_void_p),
                                netf.ctypes.data_as(C.c_void_p))
    assert rc == 0


def step_selfcol(lib, params, state, tau, out):
    """Humanoid with self-collision: state rows carry lamp[3 NPG] after laml, out rows 9 floats per group (world force first)."""
    rc = lib.hs_step_selfcol(C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p),
                             tau.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0


def step_mwc(lib, params, state, tau, out, selfcol=True, dropped=None):
    """Humanoid through the limb-per-wave sub-step of the compact store (core/engine_mwc.hpp, four role threads per env); layouts of
    step_selfcol.  dropped: optional int32 [n, 2] -- ground / self contacts refused because the slots were taken."""
    rc = lib.hs_step_mwc(C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p), tau.ctypes.data_as(C.c_void_p),
                         out.ctypes.data_as(C.c_void_p), int(selfcol), dropped.ctypes.data_as(C.c_void_p) if dropped is not None else None)
    assert rc == 0


def step_mwc_fused(lib, params, state, tau, out, selfcol=True, dropped=None):
    """the same with all sub-steps of the step inside one call per role thread (SimMWC::substeps_fused, what substep_mwc_fused_kernel runs)"""
    rc = lib.hs_step_mwc_fused(C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p), tau.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), int(selfcol), dropped.ctypes.data_as(C.c_void_p) if dropped is not None else None)
    assert rc == 0


def step_selfcol2(lib, params, state, tau, out, waves=2):
    """The same sub-step on two / three threads per env (main wave + self-collision helper [+ limit-row helper], Sim::substep roles)."""
    fn = lib.hs_step_selfcol2 if waves == 2 else lib.hs_step_selfcol3
    rc = fn(C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p),
            tau.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0


def step_terrain(lib, model, params, state, tau, out, hs, hscale, vscale, border, mu, netf):
    hs = np.ascontiguousarray(hs, np.int16)
    lib.hs_step_terrain.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    rc = lib.hs_step_terrain(model.encode(), C.cast(C.byref(params), C.c_void_p), state.shape[0], state.ctypes.data_as(C.c_void_p),
                             tau.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), hs.ctypes.data_as(C.c_void_p),
                             hs.shape[0], hs.shape[1], hscale, vscale, border, mu.ctypes.data_as(C.c_void_p),
                             netf.ctypes.data_as(C.c_void_p))
    assert rc == 0

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


def build_hand():
    from isaacgymenvs_amd.registry import generate_headers
    hdrs = generate_headers()
    os.makedirs(_OUT, exist_ok=True)
    out = os.path.join(_OUT, "libhostsim_hand.so")
    core = os.path.join(_HERE, "..", "..", "isaacgymenvs_amd", "csrc", "core")
    deps = hdrs + [os.path.join(_HERE, "hostsim.cpp"), os.path.join(core, "engine.hpp"), os.path.join(core, "hand_engine.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-DHOSTSIM_NO_HUMANOID",
                               "-DHOSTSIM_HAND", os.path.join(_HERE, "hostsim.cpp"), "-o", out])
    return C.CDLL(out)


def build(humanoid=False):
    from isaacgymenvs_amd.registry import generate_headers
    hdrs = generate_headers()
    os.makedirs(_OUT, exist_ok=True)
    name = "libhostsim_full.so" if humanoid else "libhostsim_small.so"
    out = os.path.join(_OUT, name)
    deps = hdrs + [os.path.join(_HERE, "hostsim.cpp"), os.path.join(_HERE, "..", "..", "isaacgymenvs_amd", "csrc", "core", "engine.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off"]
        if not humanoid:
            cmd.append("-DHOSTSIM_NO_HUMANOID")
        subprocess.check_call(cmd + [os.path.join(_HERE, "hostsim.cpp"), "-o", out])
    return C.CDLL(out)


def step(lib, model, params, state, tau, out):
    rc = lib.hs_step(model.encode(), C.byref(params), state.shape[0], state.ctypes.data_as(C.c_void_p),
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

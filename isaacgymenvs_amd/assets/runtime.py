"""Run-time assets: `gym.load_asset(path)` for a robot description file the engine was NOT built with (reference ant.py:149-190 loads
whatever file the task config names).

The engine is specialised per robot at build time (codegen.py: one constexpr header per model, the kernels are templates over it).  A
file whose parsed model differs from the compiled one is therefore COMPILED, once, into a library of its own and cached by the hash of
its generated header:

    parse (assets/model.py, the asset options of the compiled model it stands in for)  ->  ModelSpec
    same topology as a compiled model (bodies, joints, sensors)?  that model's task kernels apply -- otherwise NotImplementedError
    emit the header (codegen.py); equal to the compiled model's header -> the stock library
    otherwise: copy of csrc/ with that one header replaced -> hipcc (gfx950) of the translation units that include it, linked with
    the stock objects of the others; g++ of the CPU backend for sim_device="cpu"  ->  isaacgymenvs_amd/_variants/<hash>/
    native.Engine(task, ..., lib_path=that library)

What a variant may change: every number of the model (link lengths, masses, inertias, joint axes / limits / gains, contact spheres and
their count up to the compiled kernels' row-store limits).  What it may not: the kinematic tree's shape and the dof order, which the task
kernels' observation layout depends on.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

from ..registry import MODELS, load_extras, load_model, load_selfcol, sensor_bodies

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
VARIANT_DIR = os.path.join(_PKG, "_variants")

# the asset options each compiled model was loaded with (tools/compile_models.py ENTRIES cites the reference lines) and the native task
ASSET_OPTIONS = {
    "cartpole": dict(fix_base_link=True, collide_body_filter=lambda n: False),
    "ant": dict(),
    "humanoid": dict(),
    "anymal": dict(density=0.001, replace_cylinder_with_capsule=True),
    # the files Quadcopter / BallBalance write before loading them (quadcopter.py:119-198, ball_balance.py:136-218; restated in procedural.py)
    "quadcopter": dict(collide_body_filter=lambda n: False),
    "balance_bot": dict(collide_body_filter=lambda n: False),
}
TASK_OF_MODEL = {"cartpole": "Cartpole", "ant": "Ant", "humanoid": "Humanoid", "anymal": "AnymalTerrain", "quadcopter": "Quadcopter",
                 "balance_bot": "BallBalance"}


def parse(path, model_name):
    from .model import load_asset
    return load_asset(path, name=model_name, **ASSET_OPTIONS[model_name])


def same_tree(a, b):
    """the same kinematic tree: bodies, joints, their names and order (what the task kernels' observation layout depends on)"""
    import numpy as np
    return (a.nb == b.nb and a.nd == b.nd and bool(a.fixed_base) == bool(b.fixed_base) and np.array_equal(a.parent, b.parent)
            and np.array_equal(a.dof_body, b.dof_body) and np.array_equal(a.dof_type, b.dof_type) and list(a.body_names) == list(b.body_names)
            and list(a.dof_names) == list(b.dof_names))


def same_topology(a, b):
    """same_tree + the same number of contact spheres (the row store of the compiled kernels is laid out per sphere)"""
    return same_tree(a, b) and len(a.sph_body) == len(b.sph_body)


def match_model(path):
    """-> (model name, parsed spec) of the compiled model whose topology the file has, trying the options of every candidate"""
    errs = []
    for name in ASSET_OPTIONS:
        try:
            spec = parse(path, name)
        except Exception as e:  # noqa: BLE001 -- a URDF is no MJCF and vice versa
            errs.append(f"{name}: {e}")
            continue
        if same_topology(spec, load_model(name)):
            return name, spec
    raise NotImplementedError(f"{path}: no compiled model has this kinematic tree (tried {list(ASSET_OPTIONS)}); the task kernels are "
                              f"specialised per robot (isaacgymenvs_amd/codegen.py)")


def header_text(model_name, spec, sensors=None):
    """`sensors`: engine body indices of the force sensors (the Articulation task: wherever the caller put them); None = the model's own"""
    from ..codegen import emit_model_header
    e = MODELS[model_name]
    extras = load_extras(model_name) if e.get("extras") else None
    if load_selfcol(model_name) is not None:
        raise NotImplementedError("run-time variants of self-colliding models need their capsule-pair tables rebuilt (assets/model.py self_collision_tables)")
    sens = [list(spec.body_names).index(n) for n in e["sensors"]] if sensors is None else [int(b) for b in sensors]
    return emit_model_header(spec, e["struct"], sens, extras, None)


# ------------------------------------------------------------------------------------------------ a NEW kinematic tree: the Articulation task
GENERIC_MODEL = "articulation"
MAX_DOF = 32          # include/mi_engine.h MI_MAX_DOF
MAX_SPHERES = 48      # contact spheres a run-time robot keeps (parse_generic coarsens mesh sampling beyond that)


def parse_generic(path, options=None):
    """gym.load_asset of a file whose tree no compiled model has: parsed with the caller's AssetOptions into a ModelSpec of its own."""
    from .model import load_asset
    o = options
    kw = dict(fix_base_link=bool(getattr(o, "fix_base_link", False)), replace_cylinder_with_capsule=bool(getattr(o, "replace_cylinder_with_capsule", False)))
    dens = getattr(o, "density", None)
    if dens is not None and float(dens) != 1000.0:
        kw["density"] = float(dens)
    if path.endswith(".urdf"):           # mesh collision shapes -> spheres inscribed in the meshes' hulls (assets/mesh.py), relative to the URDF's package root
        # (coarse: at most 4 spheres of up to 5 cm per mesh -- the contact set of a whole robot, not of a manipulated object; the Articulation
        #  kernels unroll one contact block per sphere)
        kw.update(mesh_spheres=True, mesh_root=os.path.dirname(os.path.dirname(os.path.abspath(path))),
                  mesh_options=dict(r_cap=0.05, max_count=4))
    spec = load_asset(path, name=GENERIC_MODEL, **kw)
    # The Articulation kernels unroll one contact block per sphere: past ~48 of them the one-wave sub-step leaves the register regime the build
    # gate accepts (kuka_allegro_touch_sensor.urdf with 4 spheres per mesh: 82 spheres, 258 spilled SGPRs on gfx950 -- refused).  A robot with many
    # mesh shapes gets fewer spheres per mesh instead (2, then 1): the same robot, a coarser contact set, on both backends alike.
    for fewer in (2, 1):
        if len(spec.sph_body) <= MAX_SPHERES or "mesh_options" not in kw:
            break
        kw["mesh_options"] = dict(kw["mesh_options"], max_count=fewer)
        spec = load_asset(path, name=GENERIC_MODEL, **kw)
    if spec.nd > MAX_DOF:
        raise NotImplementedError(f"{path}: {spec.nd} dofs, the engine's parameter blocks hold {MAX_DOF} (MI_MAX_DOF)")
    return spec


def drive_split(spec, pos_dofs):
    """gym semantics of DOF_MODE_POS: the dof's `stiffness` / `damping` properties ARE the position drive's gains (amp/humanoid_amp_base.py:219-222
    switches every dof of the AMP humanoid to it, keeping the MJCF's joint stiffness / damping as gains); in DOF_MODE_NONE / EFFORT they are a passive
    spring / damper about the joint's reference.  -> (copy of the spec with the driven dofs' passive terms removed, kp [nd], kd [nd])"""
    import copy
    import numpy as np
    sp = copy.deepcopy(spec)
    kp, kd = np.zeros(spec.nd), np.zeros(spec.nd)
    k, dmp = np.array(sp.dof_stiffness, float), np.array(sp.dof_damping, float)
    for d in pos_dofs:
        kp[d], kd[d] = k[d], dmp[d]
        k[d], dmp[d] = 0.0, 0.0
    sp.dof_stiffness, sp.dof_damping = k, dmp
    return sp, kp, kd


def _includes(path, seen):
    """files reachable through #include "..." from `path` (relative to the including file's directory or csrc/)"""
    import re
    if path in seen or not os.path.exists(path):
        return
    seen.add(path)
    csrc = os.path.join(_PKG, "csrc")
    for m in re.finditer(r'#include\s+"([^"]+)"', open(path).read()):
        for base in (os.path.dirname(path), csrc):
            q = os.path.normpath(os.path.join(base, m.group(1)))
            if os.path.exists(q):
                _includes(q, seen)
                break


def _model_sizes(header):
    """(NB, ND, NSENS, NSPH) of a generated model header"""
    import re
    return tuple(int(re.search(r"\b%s = (\d+)" % k, header).group(1)) for k in ("NB", "ND", "NSENS", "NSPH"))


def dependent_sources(model_name):
    from .. import native
    csrc = os.path.join(_PKG, "csrc")
    hdr = os.path.join(csrc, "gen", f"model_{model_name}.h")
    out = []
    for s in native.SOURCES:
        seen = set()
        _includes(os.path.join(csrc, s), seen)
        if hdr in seen:
            out.append(s)
    return out


def variant_library(model_name, spec, device="cuda", verbose=False, sensors=None):
    """-> path of the library to create the engine from (the stock library when the header is the compiled model's own)"""
    from .. import native
    txt = header_text(model_name, spec, sensors)
    csrc = os.path.join(_PKG, "csrc")
    stock = open(os.path.join(csrc, "gen", f"model_{model_name}.h")).read()
    cpu = str(device).startswith("cpu")
    if txt == stock:
        return None
    h = hashlib.sha1((txt + model_name).encode()).hexdigest()[:16]
    vdir = os.path.join(VARIANT_DIR, h)
    out = os.path.join(vdir, "libmi_engine_cpu.so" if cpu else "libmi_engine.so")
    stock_lib = native.CPU_LIB_PATH if cpu else native.LIB_PATH
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(stock_lib):
        return out
    os.makedirs(vdir, exist_ok=True)
    # the ranks of a multi-GPU job ask for the same variant at the same time: one of them builds, the others wait for it and find the library
    import fcntl
    with open(os.path.join(vdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(stock_lib):
                return out
            _build_variant(model_name, txt, vdir, out, cpu, verbose)
            spec.save(os.path.join(vdir, "model.json"))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def _build_variant(model_name, txt, vdir, out, cpu, verbose):
    from .. import native
    csrc = os.path.join(_PKG, "csrc")
    src = os.path.join(vdir, "csrc")
    if os.path.isdir(src):
        shutil.rmtree(src)
    shutil.copytree(csrc, src, ignore=shutil.ignore_patterns("build", "*.o", "*.log"))
    shutil.copytree(os.path.join(_PKG, "..", "include"), os.path.join(vdir, "include"), dirs_exist_ok=True)      # csrc includes ../../include/mi_engine.h
    # (csrc/ sits two levels below the repo root: keep that shape so that "../../include" resolves)
    deep = os.path.join(vdir, "pkg", "csrc")
    if os.path.isdir(os.path.join(vdir, "pkg")):
        shutil.rmtree(os.path.join(vdir, "pkg"))
    os.makedirs(os.path.join(vdir, "pkg"))
    os.replace(src, deep)
    with open(os.path.join(deep, "gen", f"model_{model_name}.h"), "w") as f:
        f.write(txt)
    tmp = out + ".tmp"
    os.makedirs(os.path.join(deep, "build"), exist_ok=True)
    objs, procs = [], []
    if cpu:
        # the translation units of the CPU backend that instantiate this model (native.CPU_UNITS), linked with the stock objects of the others;
        # a variant that changes one of the model's SIZES (force sensors on a hand whose compiled model has none) also needs the unit that lays
        # out the arena from the task table
        resized = _model_sizes(txt) != _model_sizes(open(os.path.join(csrc, "gen", f"model_{model_name}.h")).read())
        for s, suffix, opt, defs, models in native.CPU_UNITS:
            if models is None or model_name in models or (resized and s == "mi_engine_cpu.cpp"):
                o = os.path.join(deep, "build", os.path.basename(native.cpu_object(s, suffix)))
                cmd = ["g++", opt] + native.CPU_FLAGS + defs + ["-c", os.path.join(deep, "cpu", s), "-o", o]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((s, subprocess.Popen(cmd, cwd=deep, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            else:
                o = native.cpu_object(s, suffix)
                if not os.path.exists(o):
                    raise RuntimeError(f"{o} is missing: build the stock CPU library first (native.build_cpu())")
            objs.append(o)
        for s, p in procs:
            log = p.communicate()[0]
            if p.returncode != 0:
                raise RuntimeError(f"g++ failed for the variant of {s}:\n{log.decode()[-3000:]}")
        subprocess.check_call(["g++", "-shared", "-fPIC", "-fopenmp"] + objs + ["-o", tmp], cwd=deep)
    else:
        deps = dependent_sources(model_name)
        logs = {}
        for s in native.SOURCES:
            if s in deps:
                o = os.path.join(deep, "build", s.replace(".hip", ".o"))
                cmd = [native.hipcc_path()] + native.HIPCC_FLAGS + ["-c", os.path.join(deep, s), "-o", o]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((s, subprocess.Popen(cmd, cwd=deep, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            else:
                o = os.path.join(native.BUILD_DIR, s.replace(".hip", ".o"))
                if not os.path.exists(o):
                    raise RuntimeError(f"{o} is missing: build the stock library first (native.build())")
            objs.append(o)
        for s, p in procs:
            logs[s] = p.communicate()[0].decode()
            if p.returncode != 0:
                raise RuntimeError(f"hipcc failed for the variant of {s}:\n{logs[s][-3000:]}")
        # the same gate as native.build(): a variant whose constants push a physics kernel into heavy SGPR spilling is refused, not linked
        # (that regime returned run-to-run different results on gfx950)
        for s, text in logs.items():
            with open(os.path.join(deep, "build", s.replace(".hip", ".log")), "w") as f:
                f.write(text)
        usage = native.resource_usage(os.path.join(deep, "build"), deps)
        bad = native.over_sgpr_budget(usage)
        if bad:
            raise RuntimeError(f"variant of {model_name}: step kernels spill SGPRs (known-bad regime on gfx950): {bad}")
        subprocess.check_call([native.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp], cwd=deep)
    os.replace(tmp, out)

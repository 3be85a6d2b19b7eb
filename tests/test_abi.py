"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol include/mi_engine.h declares, and the
host-only entry points (no kernel launch) behave.  No compute calls here -- those are the `-m gpu` tests."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from isaacgymenvs_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(native.LIB_PATH):
        native.build()
    return native.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mi_engine.h")).read()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.mi_abi_version() == native.MI_ABI_VERSION == 4


def test_task_info_and_unknown_task(lib):
    for task, nobs, nact in (("Cartpole", 4, 1), ("Ant", 60, 8), ("Humanoid", 108, 21)):
        info = native.task_info(task)
        assert (info.num_obs, info.num_actions) == (nobs, nact)
    info = native.MiTaskInfo()
    assert lib.mi_task_info(b"Nope", C.byref(info)) != 0
    assert b"unknown task" in lib.mi_last_error()


def test_arena_layout_host_only(lib):
    """mi_engine_create only lays out the arena (no device access): check views with a host buffer."""
    n = 100
    nbytes = lib.mi_engine_arena_bytes(b"Ant", n)
    assert nbytes > 0 and nbytes % 256 == 0
    buf = np.zeros(nbytes, np.uint8)
    sim = native.MiSimParams(dt=0.0166, substeps=2, iters=4)
    tp = native.MiLocoParams()
    h = C.c_void_p()
    rc = lib.mi_engine_create(b"Ant", C.byref(sim), C.cast(C.byref(tp), C.c_void_p), C.sizeof(tp), n, 0, 1,
                              buf.ctypes.data, nbytes, C.byref(h))
    assert rc == 0, lib.mi_last_error()
    descs = {}
    for i in range(lib.mi_engine_num_tensors(h)):
        d = native.MiTensorDesc()
        assert lib.mi_engine_tensor_desc(h, i, C.byref(d)) == 0
        descs[d.name.decode()] = d
    esz = {0: 4, 1: 8, 2: 1, 3: 4}
    spans = []
    for name, d in descs.items():
        ext = 1 + sum((d.shape[k] - 1) * d.stride[k] for k in range(d.ndim))
        assert d.byte_offset % 256 == 0
        assert d.byte_offset + ext * esz[d.dtype] <= nbytes, name
        spans.append((d.byte_offset, d.byte_offset + ext * esz[d.dtype], name))
    spans.sort()
    for (a0, a1, an), (b0, b1, bn) in zip(spans, spans[1:]):
        assert a1 <= b0, (an, bn)  # no overlap
    # reference dtypes / shapes (vec_task.py:310-323, ant.py:77-95)
    assert tuple(descs["obs_buf"].shape[:2]) == (n, 60) and tuple(descs["obs_buf"].stride[:2]) == (60, 1)
    assert descs["reset_buf"].dtype == 1
    assert descs["progress_buf"].dtype == 1 and descs["rew_buf"].dtype == 0
    assert tuple(descs["root_states"].shape[:2]) == (n, 13) and tuple(descs["root_states"].stride[:2]) == (1, n)
    assert tuple(descs["dof_state"].shape[:3]) == (n, 8, 2) and tuple(descs["dof_state"].stride[:3]) == (1, n, 8 * n)
    # bad arguments are rejected with a message
    assert lib.mi_engine_create(b"Ant", C.byref(sim), C.cast(C.byref(tp), C.c_void_p), 12, n, 0, 1, buf.ctypes.data,
                                nbytes, C.byref(h)) != 0
    assert b"size mismatch" in lib.mi_last_error()
    assert lib.mi_engine_create(b"Ant", C.byref(sim), C.cast(C.byref(tp), C.c_void_p), C.sizeof(tp), n, 0, 1,
                                buf.ctypes.data, 16, C.byref(h)) != 0
    lib.mi_engine_destroy(h)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import isaacgymenvs_amd
    with pytest.raises(RuntimeError):
        isaacgymenvs_amd.make(seed=0, task="Ant", num_envs=64, sim_device="cuda:0", rl_device="cuda:0", headless=True)
    # the reference's CPU pipeline (sim_device="cpu") is the CPU product backend, for every task since round 4 (tests/test_cpu_backend.py):
    # a library of its own that is asked for by name -- a missing HIP library never falls back onto it
    from isaacgymenvs_amd import native
    saved = native.LIB_PATH
    try:
        native.LIB_PATH, native._lib = "/nonexistent/libmi_engine.so", None
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            native.lib()
    finally:
        native.LIB_PATH = saved


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "isaacgymenvs_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("oracle/ for tests", "").replace("oracle/physics.c", "").replace("oracle/tasks.py", "").replace("oracle/hand.py", "").replace("oracle/hand.c", "").replace("oracle/bbot.py", "").replace("oracle/scene.py", "").replace("oracle/terrain_mesh.py", "").replace("NOT use oracle/ (test", ""), os.path.join(d, f)


def test_all_tasks_lay_out_and_reject_device_calls_on_a_host_arena(lib):
    """Every task of the table lays out its arena on a host buffer (views inside the arena, non-overlapping, reference shapes of
    the API tensors); the launching entry points refuse an arena that is not device memory instead of faulting."""
    cases = {"Cartpole": (native.MiCartpoleParams, 4, 1, 2), "Ant": (native.MiLocoParams, 60, 8, 8), "Humanoid": (native.MiLocoParams, 108, 21, 21),
             "AnymalTerrain": (native.MiAnymalParams, 188, 12, 12), "Anymal": (native.MiAnymalFlatParams, 48, 12, 12),
             "Quadcopter": (native.MiQuadcopterParams, 21, 12, 8), "Ingenuity": (native.MiIngenuityParams, 13, 6, 4), "BallBalance": (native.MiBallBalanceParams, 24, 3, 6), "ShadowHand": (native.MiHandParams, 211, 20, 24)}
    n = 70
    for task, (ptype, nobs, nact, nd) in cases.items():
        info = native.task_info(task)
        assert (info.num_obs, info.num_actions, info.num_dofs) == (nobs, nact, nd), task
        assert info.task_params_bytes == C.sizeof(ptype), task
        nbytes = lib.mi_engine_arena_bytes(task.encode(), n)
        buf = np.zeros(nbytes, np.uint8)
        sim = native.MiSimParams(dt=0.0166, substeps=2, iters=4)
        tp = ptype()
        if task == "ShadowHand":
            tp.obs_type, tp.num_obs, tp.cube_mass = 0, 211, 0.07
        if task == "Ingenuity":
            tp.target_period = 500
        if task == "BallBalance":
            tp.ball_mass, tp.ball_inertia, tp.ball_radius, tp.pin_stiffness = 0.84, 3.4e-3, 0.1, 5e7
        h = C.c_void_p()
        rc = lib.mi_engine_create(task.encode(), C.byref(sim), C.cast(C.byref(tp), C.c_void_p), C.sizeof(tp), n, 0, 1,
                                  buf.ctypes.data, nbytes, C.byref(h))
        assert rc == 0, (task, lib.mi_last_error())
        names = {}
        for i in range(lib.mi_engine_num_tensors(h)):
            d = native.MiTensorDesc()
            assert lib.mi_engine_tensor_desc(h, i, C.byref(d)) == 0
            names[d.name.decode()] = d
            ext = 1 + sum((d.shape[k] - 1) * d.stride[k] for k in range(d.ndim))
            assert d.byte_offset + ext * {0: 4, 1: 8, 2: 1, 3: 4}[d.dtype] <= nbytes, (task, d.name)
        assert tuple(names["obs_buf"].shape[:2]) == (n, nobs) and tuple(names["actions"].shape[:2]) == (n, nact), task
        assert tuple(names["dof_state"].shape[:3]) == (n, nd, 2), task
        for must in ("root_states", "rew_buf", "reset_buf", "progress_buf", "timeout_buf", "randomize_buf"):
            assert must in names, (task, must)
        d = native.MiTensorDesc()
        assert lib.mi_engine_tensor_desc(h, 10_000, C.byref(d)) != 0                     # bad index
        acts = np.zeros((n, nact), np.float32)
        assert lib.mi_engine_step(h, acts.ctypes.data, None) != 0 and b"not device memory" in lib.mi_last_error(), task
        assert lib.mi_engine_simulate(h, None) != 0
        ids = np.zeros(1, np.int64)
        assert lib.mi_engine_reset_idx(h, ids.ctypes.data, 1, None) != 0
        assert lib.mi_engine_reset_idx(h, ids.ctypes.data, 0, None) == 0                  # empty id list: no-op
        assert lib.mi_engine_set_option(h, b"no_such_option", 1.0) != 0 and b"unknown option" in lib.mi_last_error()
        assert lib.mi_engine_set_option(h, b"control_freq_inv", 0.0) != 0
        lib.mi_engine_destroy(h)
    # ShadowHand observation layouts are validated at creation
    tp = native.MiHandParams()
    tp.obs_type, tp.num_obs = 1, 500
    nbytes = lib.mi_engine_arena_bytes(b"ShadowHand", n)
    buf = np.zeros(nbytes, np.uint8)
    h = C.c_void_p()
    assert lib.mi_engine_create(b"ShadowHand", C.byref(native.MiSimParams(dt=0.01, substeps=2, iters=4)), C.cast(C.byref(tp), C.c_void_p),
                                C.sizeof(tp), n, 0, 1, buf.ctypes.data, nbytes, C.byref(h)) != 0
    assert b"obs_type" in lib.mi_last_error()
    # ... and so is the object: shape 0 (block), 1 (pen: radius, half length) or 2 (egg: semi-axes), the latter two with positive inertias
    def create_hand(**kw):
        t = native.MiHandParams()
        t.obs_type, t.num_obs, t.cube_mass = 0, 211, 0.07
        for k, v in kw.items():
            if isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    getattr(t, k)[i] = x
            else:
                setattr(t, k, v)
        hh = C.c_void_p()
        rc = lib.mi_engine_create(b"ShadowHand", C.byref(native.MiSimParams(dt=0.01, substeps=2, iters=4)), C.cast(C.byref(t), C.c_void_p),
                                  C.sizeof(t), n, 0, 1, buf.ctypes.data, nbytes, C.byref(hh))
        if rc == 0:
            lib.mi_engine_destroy(hh)
        return rc
    assert create_hand(object_shape=3) != 0 and b"object_shape" in lib.mi_last_error()
    assert create_hand(object_shape=2) != 0 and b"egg" in lib.mi_last_error()
    assert create_hand(object_shape=1, object_dims=[0.008, 0.1, 0.0]) != 0 and b"object_inertia" in lib.mi_last_error()
    assert create_hand(object_shape=1, object_dims=[0.008, 0.1, 0.0], object_inertia=[1.5e-4, 1.5e-4, 1.4e-6]) == 0
    assert create_hand(object_shape=2, object_dims=[0.03, 0.03, 0.04], object_inertia=[7.5e-5, 7.5e-5, 5.4e-5]) == 0
    assert create_hand(cube_mass=0.0) != 0 and b"mass" in lib.mi_last_error()
    # invalid sim parameters / env counts
    tp2 = native.MiLocoParams()
    assert lib.mi_engine_create(b"Ant", C.byref(native.MiSimParams(dt=0.0, substeps=2, iters=4)), C.cast(C.byref(tp2), C.c_void_p), C.sizeof(tp2),
                                n, 0, 1, buf.ctypes.data, nbytes, C.byref(h)) != 0
    assert lib.mi_engine_arena_bytes(b"Ant", 0) == 0 and lib.mi_engine_arena_bytes(b"Nope", 8) == 0


def test_header_is_plain_c_and_a_c_client_can_drive_the_library(lib, tmp_path):
    """include/mi_engine.h compiles as strict C99 and examples/c_abi_probe.c -- dlopen, task table, arena layout, error path -- runs
    against the built library: the boundary is a C ABI with plain pointers and sizes, nothing torch- or C++-typed."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "c_abi_probe")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_probe.c"), "-ldl", "-o", exe])
    out = subprocess.run([exe, native.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Ant            obs  60  actions  8  dofs  8" in out.stdout and "ShadowHand     obs 211  actions 20  dofs 24" in out.stdout
    assert "root_states  dtype 0  shape [64, 13]  stride [1, 64]" in out.stdout
    assert "not device memory" in out.stdout


def test_hot_kernels_stay_inside_their_register_budgets():
    """The sub-step kernels are one huge unrolled basic block each; how the register allocator copes with them depends on details that
    look innocent in the source (the `actor_params` block between the tree pass and the right-hand side is worth 44 % on the
    Humanoid step and 9 % on the ShadowHand step, the never-taken block after the kinematics pass of hand_post_kernel removes all of its
    spills).  The spilled-VGPR counts hipcc reports for the last build must stay near what was measured on the MI355X, so that an edit
    which tips a kernel into another regime fails here, on the CPU, before anybody benchmarks it."""
    from isaacgymenvs_amd import native
    native.build()
    ru = native.resource_usage()
    budgets = {                       # kernel name fragment -> most spilled VGPRs allowed (measured values in comments)
        # (the one-wave hand form: round 4's drive clamps -- one more closed-form update per limit row -- took it from 146 / 156 / 150 to these;
        #  the shapes make() launches at the BASELINE sizes are the finger-per-wave ones below, which stayed at 0)
        # (round 6: the pairs' forces on the fingertip sensors -- P, n per fingertip through the solve, A parked in the `sensor` tensor -- took it from
        #  232 / 245 / 249 to these; with A in registers as well: 426)
        "hand_substep_kernelINS_14ShadowHandTaskELi0E": 430,                 # 379 (399 before the block split)
        "hand_substep_kernelINS_14ShadowHandTaskELi1E": 440,                 # 388
        "hand_substep_kernelINS_14ShadowHandTaskELi2E": 440,                 # 393
        "hand_post_kernel": 20,                          # 0   (139 without the phi barrier)
        "hand_substep_kernelINS_15AllegroHandTaskELi0E": 0,   # 0 (16 dofs, chains of 4: the one-wave form fits its registers; 832 B scratch = the body poses handed to the narrow phase)
        "substep_sc2_kernelI13ModelHumanoid": 280,       # 235
        "substep_mwc_kernelI13ModelHumanoid": 90,        # 60, scratch 152 B / lane (round 3: one limb per wave; 154 / 312 B while the `actor_params` code was still in this kernel, 220 / 488 B without the allocation fence)
        "substep_mwc_post_kernelI13ModelHumanoid": 100,  # 72, scratch 288 B / lane (round 4: the step's last sub-step launch with post_physics_step on its role waves)
        "substep_mw_fused_post_kernelI8ModelAnt": 0,     # 0   (round 4: the Ant's whole control step in one launch)
        "substep_kernelI13ModelHumanoid": 370,           # 312
        "substep_mw_kernelI8ModelAnt": 0,                # 0
        "substep_mw_kernelI11ModelAnymal": 0,            # 0
        "substep_kernelI8ModelAnt": 0,                   # 0
        # the instantiations that read the per-body / per-dof `actor_params` factors (Sim<Scaled<M>>, kernels_scaled_*.hip)
        "substep_mwc_kernelINS_6ScaledI13ModelHumanoid": 260,   # 200 (467 without the fence)
        "substep_mw_kernelINS_6ScaledI8ModelAnt": 0,            # 0
        "substep_kernelINS_6ScaledI8ModelAnt": 48,              # 32
        "loco_post_kernelI8ModelAnt": 0,                 # 0
        "substep_mw_post_kernelI8ModelAnt": 0,           # 0   (round 3: post_physics_step on one wave of the last sub-step launch)
        # round 3, structural rather than a compiler accident: a role wave of the finger-per-wave hand sub-step holds one finger's state; with
        # 64-env workgroups (one wave per SIMD, 512 registers) nothing is spilled and no scratch is used
        "hand_substep_mw64_kernelINS_14ShadowHandTaskELi0E": 0, "hand_substep_mw64_kernelINS_14ShadowHandTaskELi1E": 0,
        "hand_substep_mw64_kernelINS_14ShadowHandTaskELi2E": 0,
        "hand_substep_mw_kernelINS_14ShadowHandTaskELi0E": 300,              # 248 (32-env workgroups: two waves per SIMD, 256 registers each)
        # the Allegro hand's finger waves (four-joint fingers, no wrist, no tendons): nothing spilled in either workgroup shape
        "hand_substep_mw64_kernelINS_15AllegroHandTaskELi0E": 0, "hand_substep_mw_kernelINS_15AllegroHandTaskELi0E": 0,
        # round 5: the ShadowHand's finger waves on Sim<Scaled<M>> (per-body link-mass factors, option hand_body_mass): 2 / 0 / 1 spilled, no scratch
        "hand_substep_mw64_kernelINS_20ScaledShadowHandTaskELi0E": 4, "hand_substep_mw64_kernelINS_20ScaledShadowHandTaskELi1E": 4,
        "hand_substep_mw64_kernelINS_20ScaledShadowHandTaskELi2E": 4,
    }
    seen = set()
    for name, use in ru.items():
        for frag, cap in budgets.items():
            if frag in name:
                seen.add(frag)
                assert use.get("VGPRs Spill", 0) <= cap, (name, use.get("VGPRs Spill"), cap)
                if "mw64" in frag:
                    assert use.get("ScratchSize", 0) == 0, (name, use.get("ScratchSize"))
    assert seen == set(budgets), set(budgets) - seen
    # spilled SGPRs: every physics kernel inside its family's budget (native.SGPR_SPILL_BUDGETS: what the build itself enforces), the kernels
    # the BASELINE sizes launch at the measured counts -- 0 for Ant / ANYmal / the hands' finger waves, 84 / 79 for the Humanoid's limb waves
    assert native.over_sgpr_budget(ru) == {}
    hot = {"substep_mw_fused_post_kernelI8ModelAntNS_11PlaneGroundELi16E": 2, "hand_substep_mw64_kernelINS_14ShadowHandTaskELi0E": 0,
           "substep_mw_fused_kernelI11ModelAnymalNS_17HeightfieldGroundELi16E": 8, "substep_mwc_kernelI13ModelHumanoid": 90,
           "substep_mwc_post_kernelI13ModelHumanoid": 90}
    for frag, cap in hot.items():
        got = [u.get("SGPRs Spill", 0) for k, u in ru.items() if frag in k]
        assert got and max(got) <= cap, (frag, got, cap)

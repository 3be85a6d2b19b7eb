"""The reference's OWN task files, unmodified, on this engine through the `isaacgym` stand-in (isaacgymenvs_amd/shims, SURVEY 8b
"B-inner"): /root/reference/isaacgymenvs/tasks/{cartpole,ant,humanoid}.py are imported as they are, construct their sim through
`gymapi`, and step.  Their jitted observation / reward functions then run on the engine's state; on the same state and actions the
fused kernels of the native task classes must give the same observations and rewards.

Runs where the reference tree is reachable (the development container; the CPU backend makes that possible without a GPU)."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "isaacgymenvs", "tasks")), reason="reference tree not reachable")

DEV = "cuda:0" if torch.cuda.is_available() else "cpu"


@pytest.fixture()
def reference_tasks():
    """Import the reference's task modules with the stand-ins registered as `isaacgym` / `gym`; its package __init__ files (hydra)
    are skipped by pre-registering the packages as namespaces, the way tools/gen_golden.py does."""
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    if DEV == "cpu":
        native.build_cpu()
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")}
    for k in saved:
        del sys.modules[k]
    shims.install(force=True)
    for name, rel in (("isaacgymenvs", "isaacgymenvs"), ("isaacgymenvs.tasks", "isaacgymenvs/tasks"),
                      ("isaacgymenvs.utils", "isaacgymenvs/utils"), ("isaacgymenvs.tasks.base", "isaacgymenvs/tasks/base")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = mod
    mods = {n: importlib.import_module("isaacgymenvs.tasks." + n) for n in ("cartpole", "ant", "humanoid")}
    vt = importlib.import_module("isaacgymenvs.tasks.base.vec_task")
    yield mods, vt
    for k in [k for k in sys.modules if k.split(".")[0] in ("isaacgymenvs", "isaacgym", "gym")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _ref_cfg(task, n):
    """the REFERENCE's own task YAML, composed by this repo's Hydra-subset composer"""
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    cfg = omegaconf_to_dict(compose("config", overrides=[f"task={task}"], cfg_dir=os.path.join(REF, "isaacgymenvs", "cfg"))["task"])
    cfg["env"]["numEnvs"] = n
    cfg["sim"]["use_gpu_pipeline"] = DEV != "cpu"
    return cfg


def test_the_file_really_is_the_reference_one(reference_tasks):
    mods, vt = reference_tasks
    assert mods["ant"].__file__ == os.path.join(REF, "isaacgymenvs", "tasks", "ant.py")
    assert vt.__file__ == os.path.join(REF, "isaacgymenvs", "tasks", "base", "vec_task.py")
    import isaacgym
    assert isaacgym._mi_shim and "shims" in isaacgym.gymapi.__file__


def test_reference_cartpole_steps_on_the_engine(reference_tasks):
    mods, vt = reference_tasks
    vt.EXISTING_SIM = None
    n = 64
    env = mods["cartpole"].Cartpole(_ref_cfg("Cartpole", n), rl_device=DEV, sim_device=DEV, graphics_device_id=-1, headless=True,
                                    virtual_screen_capture=False, force_render=False)
    g = torch.Generator().manual_seed(0)
    resets = 0
    for step in range(300):
        obs, rew, reset, info = env.step((torch.rand((n, 1), generator=g) * 2 - 1).to(DEV))
        resets += int(reset.sum())
        assert torch.isfinite(obs["obs"]).all() and obs["obs"].shape == (n, 4)
    # an unbalanced pole falls: episodes end by the angle limit well before the 500-step horizon, and restart near upright
    assert resets > n // 2
    assert float(obs["obs"][:, 2].abs().max()) < 2.0 and float(rew.max()) <= 1.0
    # the cart follows the applied effort: push right for a while from rest -> carts move right
    env.reset_idx(torch.arange(n, device=DEV))
    for _ in range(10):
        env.step(torch.ones((n, 1), device=DEV))
    assert float(env.dof_pos[:, 0].mean()) > 0.05


@pytest.mark.parametrize("task,mod,nact,z0", [("Ant", "ant", 8, 0.44), ("Humanoid", "humanoid", 21, 1.34)])
def test_reference_locomotion_task_matches_the_fused_kernels_on_the_same_state(reference_tasks, task, mod, nact, z0):
    import isaacgymenvs_amd
    mods, vt = reference_tasks
    vt.EXISTING_SIM = None
    n = 96
    ref = getattr(mods[mod], task)(_ref_cfg(task, n), rl_device=DEV, sim_device=DEV, graphics_device_id=-1, headless=True,
                                   virtual_screen_capture=False, force_render=False)
    assert ref.num_dof == nact and ref.obs_buf.shape[1] == (60 if task == "Ant" else 108)
    if task == "Humanoid":
        assert int(ref.gym.get_sim_params(ref.sim).substeps) == 2 and ref.sim.engine.get_option("self_collision") == 1.0   # filter 0 (humanoid.py:194)
    g = torch.Generator().manual_seed(1)
    for step in range(12):                                    # the reference's own step(): its reset_idx, its jitted obs / reward
        obs, rew, reset, _ = ref.step((torch.rand((n, nact), generator=g) * 2 - 1).to(DEV))
        assert torch.isfinite(obs["obs"]).all()
    assert float(ref.root_states[:, 2].min()) > 0.05 and float(ref.root_states[:, 2].mean()) < z0 + 0.5    # on the ground, not through it
    # ---- same state, same actions, one more step on both: native task class (fused kernels) vs reference task (jitted fns on the shim)
    nat = isaacgymenvs_amd.make(seed=0, task=task, num_envs=n, sim_device=DEV, rl_device=DEV, headless=True)
    nat.step(torch.zeros((n, nact), device=DEV))              # consumes the initial all-env reset
    et, nt = ref.sim.engine.tensors, nat.engine.tensors
    for k in ("root_states", "dof_state", "contact_impulse", "limit_impulse", "self_contact_impulse", "force_sensor", "dof_force"):
        if k in nt:
            nt[k].copy_(et[k])
    nat.potentials.copy_(ref.potentials); nat.prev_potentials.copy_(ref.prev_potentials)
    nat.progress_buf.copy_(ref.progress_buf); nat.reset_buf.copy_(ref.reset_buf)
    a = (torch.rand((n, nact), generator=g) * 2 - 1).to(DEV)
    keep = (ref.reset_buf == 0) & (ref.progress_buf < 900)    # envs the reference resets in this step draw from torch's RNG: not comparable
    r_obs, r_rew, r_reset, _ = ref.step(a.clone())
    n_obs, n_rew, n_reset, _ = nat.step(a.clone())
    assert int(keep.sum()) > n // 2
    d = (r_obs["obs"] - n_obs["obs"]).abs()[keep]
    d[:, [7, 8, 9]] = torch.minimum(d[:, [7, 8, 9]], (d[:, [7, 8, 9]] - 2 * np.pi).abs())
    assert float(d.max()) < 2e-4, float(d.max())              # same engine state -> jitted observations == fused kernel's
    assert float((r_rew - n_rew).abs()[keep].max()) < 2e-3 * max(1.0, float(r_rew.abs().max()))
    assert torch.equal(r_reset[keep], n_reset[keep])

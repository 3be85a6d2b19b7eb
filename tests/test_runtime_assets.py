"""Run-time assets (isaacgymenvs_amd/assets/runtime.py): `gym.load_asset` of a robot file the engine was not built with.  A copy of the
reference's nv_ant.xml with a longer front-left ankle and a heavier torso is parsed, recognised as the Ant's kinematic tree, compiled once
into a library of its own (cached by the hash of its generated header) and stepped through the `isaacgym` stand-in; the oracle, given the
same parsed spec, follows it.  Reference call site: isaacgymenvs/tasks/ant.py:149-190 (`load_asset` of whatever file the config names).

Needs the original XML: the development container's /root/reference, or the copy a GPU session stages (tools/debug/stage_reference.sh)."""
import os
import tempfile

import numpy as np
import pytest
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
XML = next((p for p in ("/root/reference/assets/mjcf/nv_ant.xml", os.path.join(_HERE, "..", "ab", "ref_stage", "assets", "mjcf", "nv_ant.xml"))
            if os.path.isfile(p)), None)
pytestmark = pytest.mark.skipif(XML is None, reason="the reference's nv_ant.xml is not reachable")


def _perturbed_ant(dirname):
    src = open(XML).read()
    mod = src.replace('fromto="0.0 0.0 0.0 0.4 0.4 0.0" name="left_ankle_geom"', 'fromto="0.0 0.0 0.0 0.55 0.55 0.0" name="left_ankle_geom"')
    mod = mod.replace('name="torso_geom" pos="0 0 0" size="0.25"', 'name="torso_geom" pos="0 0 0" size="0.28"')
    assert mod != src
    path = os.path.join(dirname, "my_ant.xml")
    with open(path, "w") as f:
        f.write(mod)
    return path


def _run(device):
    import isaacgymenvs_amd.shims as shims
    from isaacgymenvs_amd import native
    from isaacgymenvs_amd.assets import runtime
    from isaacgymenvs_amd.registry import load_model
    from oracle.engine import OracleEngine
    shims.install(force=True)
    from isaacgym import gymapi
    tmp = tempfile.mkdtemp()
    path = _perturbed_ant(tmp)
    gym = gymapi.acquire_gym()
    sp = gymapi.SimParams()
    sp.up_axis, sp.gravity, sp.dt, sp.substeps, sp.use_gpu_pipeline = gymapi.UP_AXIS_Z, gymapi.Vec3(0, 0, -9.81), 1 / 60.0, 2, device != "cpu"
    sp.physx.max_depenetration_velocity = 10.0
    sim = gym.create_sim(0, -1, gymapi.SIM_PHYSX, sp)
    gym.add_ground(sim, gymapi.PlaneParams())
    asset = gym.load_asset(sim, tmp, "my_ant.xml", gymapi.AssetOptions())
    stock = load_model("ant")
    assert asset.variant and asset.model_name == "ant" and asset.task == "Ant"
    assert runtime.same_topology(asset.spec, stock)
    assert abs(asset.spec.total_mass() - stock.total_mass()) > 0.05                     # the file's numbers, not the compiled model's
    n = 16
    for i in range(n):
        env = gym.create_env(sim, gymapi.Vec3(), gymapi.Vec3(), 4)
        gym.create_actor(env, asset, gymapi.Transform(gymapi.Vec3(0, 0, 1.0)), "ant", i, 1, 0)
    gym.prepare_sim(sim)                                                                  # builds (or finds) the variant library
    lib_path = runtime.variant_library("ant", asset.spec, sim.device)
    assert lib_path is not None and os.path.exists(lib_path) and "_variants" in lib_path
    assert sim.engine.L is native.variant_lib(lib_path)
    t_build = os.path.getmtime(lib_path)
    assert runtime.variant_library("ant", asset.spec, sim.device) == lib_path and os.path.getmtime(lib_path) == t_build   # cached: built once
    root = gym.acquire_actor_root_state_tensor(sim)
    dof = gym.acquire_dof_state_tensor(sim)
    lo, up = np.minimum(asset.spec.dof_lower, asset.spec.dof_upper), np.maximum(asset.spec.dof_lower, asset.spec.dof_upper)
    q0 = torch.tensor(0.5 * (lo + up), dtype=torch.float32, device=sim.device).repeat(n, 1)
    ds = dof.view(n, 8, 2).clone()
    ds[..., 0], ds[..., 1] = q0, 0.0
    gym.set_dof_state_tensor(sim, ds)
    sens = [list(asset.spec.body_names).index(b) for b in ("front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot")]
    prm = dict(dt=1 / 60.0, substeps=2, iters=4, gravity=(0, 0, -9.81), contact_offset=0.02, rest_offset=0.0, max_depen_vel=10.0, erp=0.5, plane_mu=1.0,
               ground_z=0.0, cfm=1e-6, warm=1.0)
    mw = int(sim.engine.get_option("multi_wave")) if device != "cpu" else 0
    kw = {}
    if mw:
        from isaacgymenvs_amd.assets.model import solver_blocks
        kw = dict(solver="blocks", blocks=solver_blocks(asset.spec))
    orc = OracleEngine(asset.spec, n, params=prm, sensor_bodies=sens, precision="f64", **kw)
    other = OracleEngine(stock, n, params=prm, sensor_bodies=sens, precision="f64", **({"solver": "blocks", "blocks": solver_blocks(stock)} if mw else {}))
    for o in (orc, other):
        o.root[:] = sim.engine.tensors["root_states"].cpu().numpy()
        o.q[:] = q0.cpu().numpy()
    g = torch.Generator().manual_seed(0)
    touched = 0
    for step in range(45):
        tau = (torch.rand((n, 8), generator=g) * 2 - 1) * 15
        gym.set_dof_actuation_force_tensor(sim, tau.to(sim.device))
        gym.simulate(sim)
        orc.step(tau.numpy()); other.step(tau.numpy())
        touched += int((np.abs(orc.sph_force).sum(-1) > 0).sum())
        gym.refresh_actor_root_state_tensor(sim); gym.refresh_dof_state_tensor(sim)
        # fp32 engine vs fp64 oracle: tight until the landing, then per env (an impact amplifies rounding differences in single envs)
        tol = 2e-3 * (1 + step / 5)
        err = np.maximum(np.abs(root.cpu().numpy()[:, :7] - orc.root[:, :7]).max(1), np.abs(dof.view(n, 8, 2)[..., 0].cpu().numpy() - orc.q).max(1))
        assert (err < tol).mean() >= 0.85 and np.median(err) < 0.2 * tol, (step, err)
    assert touched > 50                                                                   # they land and stand on the (longer) legs
    assert float(root[:, 2].min()) > 0.15
    # and it is NOT the compiled Ant: the stock model, given the same efforts, ends up elsewhere
    assert np.abs(other.q - orc.q).max() > 0.05
    gym.destroy_sim(sim)


def test_perturbed_ant_loads_builds_once_and_matches_the_oracle_on_the_cpu_backend():
    from isaacgymenvs_amd import native
    native.build_cpu()
    _run("cpu")


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X")
def test_perturbed_ant_loads_builds_once_and_matches_the_oracle_on_the_gpu():
    """the same on the HIP backend: hipcc (gfx950) of the translation units that include the Ant's header, linked with the stock objects"""
    _run("cuda:0")

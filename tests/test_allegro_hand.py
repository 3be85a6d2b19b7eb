"""AllegroHand (reference isaacgymenvs/tasks/allegro_hand.py) on the CPU: the mesh -> sphere sampler, the compiled model, the oracle's
observation layouts against the reference's own method bodies (tests/golden/allegro_hand.npz, tools/gen_golden_allegro.py), the column maps the
engine uses for the narrower layouts, and a rollout of the oracle task (the HIP path is tests/test_gpu_allegro_hand.py)."""
import os

import numpy as np
import pytest

from isaacgymenvs_amd.registry import load_extras, load_model

REF_ASSETS = "/root/reference/assets"


def _box_points(hx, hy, hz):
    return np.array([[x, y, z] for x in (-hx, hx) for y in (-hy, hy) for z in (-hz, hz)], float)


@pytest.mark.parametrize("half", [(0.05, 0.02, 0.01), (0.036, 0.0125, 0.014), (0.05, 0.058, 0.049)])
def test_hull_spheres_lie_inside_the_hull_touch_it_and_cover_it(half):
    from isaacgymenvs_amd.assets.mesh import hull_spheres
    V = _box_points(*half)
    C, r = hull_spheres(V)
    h = np.array(half)
    assert len(C) >= 1 and r <= 0.95 * h.min() + 1e-12 and r <= 0.015 + 1e-12
    depth = (h[None, :] - np.abs(C)).min(axis=1)                 # distance of a centre to the nearest face of the box
    assert (depth >= r - 1e-6).all()                             # inside
    assert (depth <= r + 1e-6).all()                             # tangent to at least one face
    # coverage: every point of the inner offset surface is within 0.8 r of a centre (so neighbours are <= 1.6 r apart) unless the cap cut it
    if len(C) < 48:
        g = np.linspace(-1, 1, 9)
        inner = h - r
        pts = np.array([[a * inner[0], b * inner[1], c * inner[2]] for a in g for b in g for c in g if max(abs(a), abs(b), abs(c)) == 1.0])
        d = np.linalg.norm(pts[:, None, :] - C[None, :, :], axis=2).min(axis=1)
        assert d.max() <= 0.8 * r + 0.3 * r, d.max() / r            # candidate grid step 0.8 r / ... : a quarter radius of slack
    # farthest-point order: the first two centres are the farthest pair's ends (spread-out manifold)
    if len(C) >= 3:
        assert np.linalg.norm(C[0] - C[1]) >= np.linalg.norm(C[0] - C[2]) - 1e-9


def test_hull_spheres_are_deterministic_and_refuse_slivers():
    from isaacgymenvs_amd.assets.mesh import hull_spheres
    V = _box_points(0.03, 0.02, 0.015)
    a, ra = hull_spheres(V)
    b, rb = hull_spheres(V.copy())
    np.testing.assert_array_equal(a, b)
    assert ra == rb
    C, r = hull_spheres(_box_points(0.03, 0.02, 0.001))          # 2 mm thick: no ball of radius 3 mm fits
    assert len(C) == 0 and r == 0.0


def test_obj_and_stl_loaders(tmp_path):
    from isaacgymenvs_amd.assets.mesh import load_mesh
    p = tmp_path / "t.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\nf -4 -3 -2\n")
    V, F = load_mesh(str(p))
    assert V.shape == (4, 3) and F.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]
    s = tmp_path / "t.stl"
    s.write_text("solid t\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid t\n")
    V, F = load_mesh(str(s))
    assert V.shape == (3, 3) and F.tolist() == [[0, 1, 2]]


def test_compiled_model_is_the_allegro_hand_of_the_task():
    """allegro_hand.py:233-264: 16 dofs, all driven, stiffness 3 / damping 0.1 / armature 0.001 written into every actor; fix_base_link +
    collapse_fixed_joints leave the mount + palm as the root body and four 4-link fingers."""
    spec, ex = load_model("allegro_hand"), load_extras("allegro_hand")
    assert (spec.nb, spec.nd, spec.nv) == (17, 16, 16)
    assert list(spec.dof_names) == [f"{f}_joint_{k}" for f in ("index", "middle", "ring", "thumb") for k in range(4)]
    assert list(spec.parent) == [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15]
    np.testing.assert_allclose(spec.dof_damping, 0.1)
    np.testing.assert_allclose(spec.dof_armature, 0.001)
    assert ex["dof_kp"] == [3.0] * 16 and ex["actuated_dofs"] == list(range(16)) and ex["tendons"] == [] and ex["fingertips"] == []
    assert abs(spec.total_mass() - 1.0217) < 1e-3                 # the URDF's link masses
    body, rad = np.array(ex["os_body"]), np.array(ex["os_rad"])
    assert len(body) == 119 and set(body.tolist()) == set(range(17))          # every body carries contact spheres
    assert (np.diff(body) >= 0).all()                                          # grouped by body (the engine walks a body's spheres in a loop)
    assert 0.005 < rad.min() and rad.max() <= 0.015 + 1e-12
    assert (body == 0).sum() == 48                                             # the palm: capped


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference asset tree not present")
def test_committed_extras_regenerate_from_the_reference_meshes():
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    sp = importlib.util.spec_from_file_location("compile_models", os.path.join(here, "..", "tools", "compile_models.py"))
    cm = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(cm)
    from isaacgymenvs_amd.assets.model import load_asset
    e = cm.ENTRIES["allegro_hand"]
    full = load_asset(os.path.join(REF_ASSETS, e["file"]), name="allegro_hand", fix_base_link=True, mesh_spheres=True,
                      mesh_root=os.path.join(REF_ASSETS, "urdf"), mesh_link_filter=lambda link: link != "allegro_mount")
    ex = cm.allegro_extras(REF_ASSETS, full)
    ours = load_extras("allegro_hand")
    assert ex["os_body"] == ours["os_body"]
    np.testing.assert_allclose(ex["os_pos"], ours["os_pos"], atol=1e-9)
    np.testing.assert_allclose(ex["os_rad"], ours["os_rad"], atol=1e-12)


def test_oracle_observation_layouts_equal_the_reference_method_bodies(golden_dir):
    from oracle.tasks import compute_allegro_observations
    g = np.load(os.path.join(golden_dir, "allegro_hand.npz"))
    args = (g["dof_pos"], g["dof_vel"], g["dof_force"], g["dof_lower"], g["dof_upper"], g["object_state"], g["goal_pose"], g["actions"], 0.2, 10.0)
    for name, key in (("full_state", "full_state"), ("full_state", "states"), ("full", "full"), ("full_no_vel", "full_no_vel")):
        np.testing.assert_allclose(compute_allegro_observations(name, *args), g[key], rtol=0, atol=1e-6)


def test_engine_column_maps_select_the_reference_layouts(golden_dir):
    """The engine computes the 88-wide full state and copies columns of it for the narrower layouts (MiHandParams.obs_map)."""
    from isaacgymenvs_amd.tasks.allegro_hand import NUM_OBS, obs_columns
    g = np.load(os.path.join(golden_dir, "allegro_hand.npz"))
    for name in ("full_no_vel", "full", "full_state"):
        cols = obs_columns(name)
        assert len(cols) == NUM_OBS[name]
        np.testing.assert_array_equal(g["full_state"][:, cols], g[name])
    with pytest.raises(Exception, match="Unknown type of observations"):
        obs_columns("openai")


def test_task_parameters_follow_the_reference_constructor():
    from isaacgymenvs_amd.tasks.allegro_hand import allegro_params_from_cfg, hand_start_quat
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    cfg = omegaconf_to_dict(compose("config", overrides=["task=AllegroHand"])["task"])
    p = allegro_params_from_cfg(cfg)
    q = hand_start_quat()
    assert abs(np.linalg.norm(q) - 1) < 1e-12
    np.testing.assert_allclose(list(p.hand_quat), q, atol=1e-7)
    np.testing.assert_allclose(list(p.hand_pos), [0, 0, 0.5])
    np.testing.assert_allclose(list(p.object_init_pos), [0, -0.2, 0.56], atol=1e-7)          # allegro_hand.py:285-291
    np.testing.assert_allclose(list(p.goal_init_pos), [0, -0.2, 0.52], atol=1e-7)            # :376-377
    assert abs(p.cube_half - 0.0325) < 1e-7 and abs(p.cube_mass - 400.0 * 0.065 ** 3) < 1e-7   # cube_multicolor_allegro.urdf
    assert (p.num_obs, p.obs_type, list(p.actuated[:16])) == (88, 0, list(range(16)))
    assert p.rew.max_episode_length == 600.0 and p.rew.ignore_z_rot == 0
    cfg["env"]["objectType"], cfg["env"]["observationType"] = "pen", "full_no_vel"
    p = allegro_params_from_cfg(cfg)
    assert abs(p.object_init_pos[2] - 0.52) < 1e-7 and p.rew.ignore_z_rot == 1 and p.num_obs == 50 and p.object_shape == 1
    assert list(p.obs_map[:50]) == list(range(16)) + list(range(48, 55)) + list(range(61, 72)) + list(range(72, 88))


def test_oracle_task_rollout_keeps_the_cube_in_hand_and_counts_resets():
    """The CPU restatement of the task (oracle/tasks.py OracleAllegroHandEnv on oracle/hand.c) from the reference's reset state: the cube lands
    on the fingers (contacts), stays finite, resets follow compute_hand_reward's fall test."""
    from isaacgymenvs_amd.tasks.allegro_hand import allegro_params_from_cfg
    from isaacgymenvs_amd.utils.config import compose, omegaconf_to_dict
    from oracle.tasks import OracleAllegroHandEnv
    cfg = omegaconf_to_dict(compose("config", overrides=["task=AllegroHand"])["task"])
    p = allegro_params_from_cfg(cfg)
    sim = dict(dt=cfg["sim"]["dt"], substeps=2, iters=8, gravity=(0, 0, -9.81), contact_offset=0.002, rest_offset=0.0, max_depen_vel=1000.0,
               erp=0.2, plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=0.0)
    n = 24
    env = OracleAllegroHandEnv(load_model("allegro_hand"), load_extras("allegro_hand"), [], sim, p, n, seed=3,
                               control_freq_inv=cfg["env"]["controlFrequencyInv"])
    rng = np.random.default_rng(0)
    touched = np.zeros(n, bool)
    for step in range(30):
        obs, rew, reset = env.step((rng.random((n, 16)) * 2 - 1).astype(np.float32) * 0.3)
        assert obs.shape == (n, 88) and np.isfinite(obs).all() and np.isfinite(rew).all()
        touched |= env.eng.ncontacts > 0
        fell = np.linalg.norm(env.eng.obj[:, 0:3] - env.goal_states[:, 0:3], axis=1) >= p.rew.fall_dist
        assert (reset[fell] == 1).all()
    assert touched.mean() > 0.9
    assert (np.abs(env.eng.q) <= np.maximum(np.abs(env.lo), np.abs(env.up)) + 0.05).all()      # joint limits hold

"""Property tests of the host terrain generator (isaacgymenvs_amd/tasks/terrain.py), the restatement of the reference's
`Terrain` (isaacgymenvs/tasks/anymal_terrain.py:543-673) and of the `isaacgym.terrain_utils` primitives it calls (that
module is part of the closed Isaac Gym package and absent from the reference tree, so these are geometry known-answer
checks, not golden comparisons)."""
import numpy as np
import pytest

from isaacgymenvs_amd.tasks import terrain as T

HS, VS = 0.1, 0.005


def _sub(n=80):
    return T.SubTerrain("terrain", width=n, length=n, vertical_scale=VS, horizontal_scale=HS)


@pytest.mark.parametrize("slope", [0.2, -0.3, 0.0])
def test_pyramid_slope_geometry(slope):
    t = T.pyramid_sloped_terrain(_sub(), slope=slope, platform_size=3.0)
    h = t.height_field_raw.astype(float) * VS
    assert t.height_field_raw.dtype == np.int16
    # borders are at height 0, the centre platform is flat at the clipped apex height, the sign follows the slope
    assert h[0, :].max() == 0 and h[:, 0].max() == 0 and h[0, :].min() == 0
    plat = h[40 - 14:40 + 14, 40 - 14:40 + 14]
    assert np.ptp(plat) == 0.0
    assert np.sign(plat[0, 0]) == np.sign(slope)
    # along the mid line the height profile is piecewise linear with the requested slope (int16 truncation)
    mid = h[:12, 40]
    if slope != 0:
        np.testing.assert_allclose(np.diff(mid), slope * HS, atol=VS + 1e-9)
    # 4-fold symmetry of the pyramid
    np.testing.assert_array_equal(t.height_field_raw[1:, 1:], t.height_field_raw[1:, 1:][::-1, ::-1])


@pytest.mark.parametrize("step_height", [0.15, -0.1])
def test_pyramid_stairs_geometry(step_height):
    t = T.pyramid_stairs_terrain(_sub(), step_width=0.31, step_height=step_height, platform_size=3.0)
    raw = t.height_field_raw
    sw, sh = int(0.31 / HS), int(step_height / VS)
    levels = np.unique(raw)
    # heights are integer multiples of the step, rings are `step_width` pixels wide, the outermost ring is level 0
    assert set(levels.tolist()) <= {k * sh for k in range(0, 40)}
    assert (raw[:sw, :] == 0).all() and (raw[sw:2 * sw, sw:-sw] == sh).all()
    # monotone towards the centre
    line = raw[40, :41].astype(int)
    assert (np.diff(line) * np.sign(sh) >= 0).all()
    assert abs(raw[40, 40]) == abs(levels).max()


def test_discrete_obstacles_properties():
    rng = np.random.RandomState(3)
    t = T.discrete_obstacles_terrain(_sub(), 0.15, 1.0, 2.0, 40, platform_size=3.0, rng=rng)
    raw = t.height_field_raw
    mh = int(0.15 / VS)
    assert set(np.unique(raw).tolist()) <= {-mh, -mh // 2, 0, mh // 2, mh}
    assert (raw[25:55, 25:55] == 0).all()          # 3 m platform
    assert (raw != 0).sum() > 200                  # obstacles exist
    # deterministic for a given stream
    t2 = T.discrete_obstacles_terrain(_sub(), 0.15, 1.0, 2.0, 40, platform_size=3.0, rng=np.random.RandomState(3))
    np.testing.assert_array_equal(raw, t2.height_field_raw)


def test_stepping_stones_properties():
    rng = np.random.RandomState(4)
    t = T.stepping_stones_terrain(_sub(), stone_size=1.0, stone_distance=0.1, max_height=0.0, platform_size=3.0, rng=rng)
    raw = t.height_field_raw
    depth = int(-10 / VS)
    vals = set(np.unique(raw).tolist())
    assert depth in vals and vals <= {depth, -1, 0}     # pit, stones at height in [-1, 0) units, platform 0
    assert (raw[25:55, 25:55] == 0).all()
    # gaps are one pixel wide (0.1 m) columns/rows of pit between 10-pixel stones
    col_is_gap = (raw[:, :20] == depth).all(axis=0)
    assert col_is_gap.sum() >= 1


def test_random_uniform_is_additive_bounded_and_quantised():
    rng = np.random.RandomState(5)
    t = _sub()
    t.height_field_raw[:] = 7
    T.random_uniform_terrain(t, min_height=-0.1, max_height=0.1, step=0.025, downsampled_scale=0.2, rng=rng)
    d = t.height_field_raw.astype(int) - 7
    assert d.min() >= int(-0.1 / VS) - 1 and d.max() <= int(0.1 / VS) + 1
    assert d.std() > 2          # noise present
    # the coarse and fine grids (both linspace over the tile) share only the corner nodes: those are exact multiples of the
    # height step (0.025 m = 5 units); everything else is a linear blend of such values
    assert all(int(v) % 5 == 0 for v in (d[0, 0], d[0, -1], d[-1, 0], d[-1, -1]))
    assert np.abs(np.diff(d, axis=0)).max() <= 40 // 2 + 1     # no jump larger than one coarse cell allows


def _cfg(curriculum):
    return dict(terrainType="trimesh", mapLength=8.0, mapWidth=8.0, numLevels=4, numTerrains=5, curriculum=curriculum,
                terrainProportions=[0.1, 0.1, 0.35, 0.25, 0.2])


@pytest.mark.parametrize("curriculum", [True, False])
def test_terrain_layout_and_origins(curriculum):
    ter = T.Terrain(_cfg(curriculum), num_robots=64, seed=1)
    assert ter.border == 200 and ter.tot_rows == 4 * 80 + 400 and ter.tot_cols == 5 * 80 + 400
    assert ter.heightsamples.dtype == np.int16 and ter.heightsamples.shape == (ter.tot_rows, ter.tot_cols)
    # the 20 m border stays flat
    hs = ter.heightsamples
    assert (hs[:200] == 0).all() and (hs[-200:] == 0).all() and (hs[:, :200] == 0).all() and (hs[:, -200:] == 0).all()
    # env origins: tile centres, z = max height of the central 2 m x 2 m patch (anymal_terrain.py:664-672)
    for i in range(4):
        for j in range(5):
            ox, oy, oz = ter.env_origins[i, j]
            assert ox == (i + 0.5) * 8.0 and oy == (j + 0.5) * 8.0
            tile = hs[200 + 80 * i:200 + 80 * (i + 1), 200 + 80 * j:200 + 80 * (j + 1)]
            assert oz == pytest.approx(tile[30:50, 30:50].max() * VS)
    # same seed -> same terrain on every rank; another seed differs (when random parts exist)
    ter2 = T.Terrain(_cfg(curriculum), num_robots=64, seed=1)
    np.testing.assert_array_equal(hs, ter2.heightsamples)


def test_curriculum_difficulty_grows_with_level():
    cfg = dict(_cfg(True), numLevels=10, numTerrains=20)
    ter = T.Terrain(cfg, num_robots=200, seed=0)
    hs = ter.heightsamples.astype(float) * VS
    # cumulative proportions [.1,.2,.55,.8,1]: column 12 of 20 (choice 0.6) is stairs-up, column 8 (0.4) stairs-down
    # (anymal_terrain.py:650-653): the apex height grows / falls monotonically with the level
    apex = [hs[200 + 80 * i + 40, 200 + 80 * 12 + 40] for i in range(10)]
    assert all(b >= a for a, b in zip(apex, apex[1:])) and apex[-1] > apex[0] > 0
    pit = [hs[200 + 80 * i + 40, 200 + 80 * 8 + 40] for i in range(10)]
    assert all(b <= a for a, b in zip(pit, pit[1:])) and pit[-1] < pit[0] < 0
    # column 0 (choice 0 < 0.05) is a downward slope
    assert hs[200 + 80 * 9 + 40, 200 + 40] < 0


def test_plane_terrain_builds_nothing():
    ter = T.Terrain(dict(terrainType="plane"), num_robots=4)
    assert not hasattr(ter, "heightsamples")


def test_ground_query_slope_correction_tracks_the_corrected_trimesh():
    """terrain.slopeTreshold (anymal_terrain.py:576): the reference's mesh generator slides the lower vertex of a steep edge under the
    upper one.  The engine's ground query levels steep cell edges to their lower end instead (csrc/core/engine.hpp, oracle/physics.c);
    oracle/terrain_mesh.py holds the corrected mesh itself (restated from the Isaac Gym package's terrain_utils, not in /root/reference)
    and this test measures how close the two surfaces are on the task's own terrain."""
    from isaacgymenvs_amd.tasks.terrain import Terrain
    from isaacgymenvs_amd.utils.config import compose
    from oracle.terrain_mesh import snapped_height, trimesh_height, trimesh_vertices
    cfg = compose(overrides=["task=AnymalTerrain"])["task"]["env"]["terrain"]
    t = Terrain(cfg, num_robots=512, seed=3)
    assert t.slope_threshold == 0.5
    hf, hs, vs = t.height_field_raw, t.horizontal_scale, t.vertical_scale
    rng = np.random.default_rng(0)
    n = 60000
    px = rng.uniform(2 * hs, (hf.shape[0] - 3) * hs, n); py = rng.uniform(2 * hs, (hf.shape[1] - 3) * hs, n)
    exact = trimesh_height(hf, hs, vs, t.slope_threshold, px, py)
    snapped = np.abs(snapped_height(hf, hs, vs, t.slope_threshold, px, py) - exact)
    plain = np.abs(snapped_height(hf, hs, vs, None, px, py) - exact)
    assert (snapped < 1e-3).mean() > 0.96 and snapped.mean() < 1e-3           # measured 97.0 %, mean 0.6 mm
    assert (plain < 1e-3).mean() < 0.90 and plain.mean() > 5 * snapped.mean()  # the uncorrected grid: 88 %, mean 6.9 mm
    # without a threshold the two restatements are the same surface
    exact0 = trimesh_height(hf, hs, vs, None, px[:5000], py[:5000], reach=0)
    np.testing.assert_allclose(snapped_height(hf, hs, vs, None, px[:5000], py[:5000]), exact0, atol=1e-9)
    # the corrected mesh really has vertical risers: a stair tile has vertices that share their xy position with a neighbour
    xx, yy, _ = trimesh_vertices(hf, hs, vs, t.slope_threshold)
    moved = (np.abs(xx - np.arange(hf.shape[0])[:, None] * hs) > 1e-9) | (np.abs(yy - np.arange(hf.shape[1])[None, :] * hs) > 1e-9)
    assert 0.02 < moved.mean() < 0.5


def test_oracle_ground_query_applies_the_slope_threshold():
    """oracle/physics.c ground_query against its numpy twin (oracle/terrain_mesh.py::snapped_height):
    a noisy stair field queried point by point through the C library's or_ground_query."""
    import ctypes as C
    from oracle.engine import OracleEngine, _ptr
    from isaacgymenvs_amd.registry import load_model
    from oracle.terrain_mesh import snapped_height
    rng = np.random.default_rng(1)
    rows, cols, hs, vs, border = 60, 50, 0.1, 0.005, 1.0
    hf = (np.arange(rows)[:, None] // 3 * 26 + rng.integers(-2, 3, (rows, cols))).astype(np.int16)     # stairs of 13 cm, noisy treads
    eng = OracleEngine(load_model("anymal"), 1, precision="f64")
    eng.set_ground(hf, hs, vs, border, slope_threshold=0.5)
    px = rng.uniform(0.3, (rows - 3) * hs, 400); py = rng.uniform(0.3, (cols - 3) * hs, 400)
    z = np.zeros(1); nrm = np.zeros(3)
    got = []
    for x, y in zip(px, py):
        eng.lib.or_ground_query(C.byref(eng.ground), C.c_double(0.0), C.c_double(x - border), C.c_double(y - border), _ptr(z), _ptr(nrm))
        got.append(z[0])
        assert abs(np.linalg.norm(nrm) - 1) < 1e-12 and nrm[2] > 0
    np.testing.assert_allclose(got, snapped_height(hf, hs, vs, 0.5, px, py), atol=1e-12)
    assert np.abs(np.array(got) - snapped_height(hf, hs, vs, None, px, py)).max() > 0.05


def _contact(eng_ground, lib, creal, x, y, z, r):
    import ctypes as C
    d = creal(0.0); n = (creal * 3)()
    lib.or_ground_contact(C.byref(eng_ground), creal(0.0), creal(x), creal(y), creal(z), creal(r), C.byref(d), n)
    return d.value, np.array(list(n))


def test_riser_walls_of_the_corrected_mesh_collide_from_the_side():
    """The slope-corrected triangle mesh the reference hands to PhysX (anymal_terrain.py:198-211, :576) has vertical faces where a stair rises;
    feet meet them from the side.  The engine's ground is the height field itself: HeightfieldGround::contact / oracle ground_contact add the
    wall of a riser in the sphere's own cell as a contact candidate.  Checked here against the restated mesh (oracle/terrain_mesh.py
    trimesh_closest, exact point-to-triangle distances): beside a riser, clear of tread and top edge, distance and normal are the mesh's;
    above the top, far from the wall, or with the walls switched off the contact is the surface below; host build == oracle."""
    import ctypes as C
    from oracle.engine import OracleEngine
    from isaacgymenvs_amd.registry import load_model
    from oracle.terrain_mesh import trimesh_closest
    from tests.hostbuild import hostsim
    rows, cols, hs, vs, border, r = 40, 36, 0.1, 0.005, 0.0, 0.03
    ii, jj = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    hf = (np.maximum(np.abs(ii - 20) // 4, np.abs(jj - 18) // 4) * 30).astype(np.int16)      # a pit of square 15 cm stairs: walls in +-x and +-y
    eng = OracleEngine(load_model("anymal"), 1, precision="f64")
    eng.set_ground(hf, hs, vs, border, slope_threshold=0.5)
    lib = eng.lib
    lib.or_ground_contact.restype = None
    rng = np.random.default_rng(5)
    n = 4000
    px = rng.uniform(3 * hs, (rows - 4) * hs, n); py = rng.uniform(3 * hs, (cols - 4) * hs, n)
    # heights between the tread below (+ r + 1 cm: the tread is not the nearer surface) and the riser's top (- 1 cm)
    ci, cj = np.floor(px / hs).astype(int), np.floor(py / hs).astype(int)
    low = np.minimum(np.minimum(hf[ci, cj], hf[ci + 1, cj]), np.minimum(hf[ci, cj + 1], hf[ci + 1, cj + 1])) * vs
    pz = low + r + 0.01 + rng.uniform(0, 0.15 - r - 0.02, n)
    dist, closest = trimesh_closest(hf, hs, vs, 0.5, np.stack([px, py, pz], 1))
    side = 0
    for k in range(n):
        d, nrm = _contact(eng.ground, lib, C.c_double, px[k], py[k], pz[k], r)
        to_mesh = np.array([px[k], py[k], pz[k]]) - closest[k]
        horizontal = abs(to_mesh[2]) < 1e-9 and dist[k] > 1e-6
        c = hf[ci[k]:ci[k] + 2, cj[k]:cj[k] + 2].astype(int)
        clean = (c[0, 0] == c[0, 1] and c[1, 0] == c[1, 1]) or (c[0, 0] == c[1, 0] and c[0, 1] == c[1, 1])   # a straight riser crosses the cell
        if horizontal and dist[k] - r < 0.02 and clean:      # the mesh's nearest feature is a wall face within the contact range
            # (corner cells, where the corrected mesh folds two walls into one cell, are modelled as one full wall: not compared)
            side += 1
            assert abs(d - (dist[k] - r)) < 1e-9, (k, d, dist[k] - r)
            np.testing.assert_allclose(nrm, to_mesh / dist[k], atol=1e-9)
        elif abs(nrm[2]) < 0.5 and clean and d < 0.02:                    # a wall contact of ours in a straight-riser cell: the mesh has that wall, at that distance
            assert abs(d - (dist[k] - r)) < 1e-9
    assert side > 150
    # above the risers' tops, or with the walls off, the contact is the surface below (normal z > 0)
    for k in range(200):
        d, nrm = _contact(eng.ground, lib, C.c_double, px[k], py[k], 0.15 * 6 + 0.2, r)
        assert nrm[2] > 0.5
    eng.set_ground(hf, hs, vs, border, slope_threshold=0.5, walls=False)
    for k in range(400):
        d, nrm = _contact(eng.ground, lib, C.c_double, px[k], py[k], pz[k], r)
        assert nrm[2] > 0.5
    eng.set_ground(hf, hs, vs, border, slope_threshold=0.5)
    # the fp32 host build of the engine's header agrees with the oracle
    hl = hostsim.build()
    hl.hs_set_slope_threshold.argtypes = [C.c_float]; hl.hs_set_slope_threshold.restype = None
    hl.hs_set_slope_threshold(0.5 * hs / vs)
    hl.hs_ground_contact.restype = None
    hfc = np.ascontiguousarray(hf)
    for k in range(600):
        d, nrm = _contact(eng.ground, lib, C.c_double, px[k], py[k], pz[k], r)
        df = C.c_float(0); nf = (C.c_float * 3)()
        hl.hs_ground_contact(hfc.ctypes.data_as(C.c_void_p), rows, cols, C.c_float(hs), C.c_float(vs), C.c_float(border), C.c_float(px[k]),
                             C.c_float(py[k]), C.c_float(pz[k]), C.c_float(r), C.byref(df), nf)
        if abs(d - df.value) > 1e-4:      # (a point within fp32 rounding of a cell boundary may sit in the other cell)
            gx, gy = px[k] / hs, py[k] / hs
            assert min(abs(gx - round(gx)), abs(gy - round(gy))) < 1e-4
            continue
        np.testing.assert_allclose(np.array(list(nf)), nrm, atol=1e-5)
    hl.hs_set_slope_threshold(0.0)


def test_a_riser_stops_a_walking_foot_from_the_side():
    """Dynamics of the wall contact in the oracle's physics (the GPU kernels follow it, tests/test_gpu_*): an ANYmal holding its default pose
    runs at 1 m/s into a 25 cm riser.  With the walls (default) its front feet stop at the riser's face -- never deeper than a few
    millimetres --, the net contact force on the front shanks points back and the robot stays on its feet; with `terrain_walls` off the feet
    pass through the face and are thrown out from below."""
    from oracle.engine import OracleEngine
    from isaacgymenvs_amd.registry import load_model
    spec = load_model("anymal")
    rows, cols, hs, vs, border = 80, 40, 0.1, 0.005, 0.0
    hf = np.zeros((rows, cols), np.int16)
    hf[40:, :] = 50                                         # tread at 0.25 m from x = 4.0 on: the riser's wall stands at x = 4.0
    q0 = np.array([0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8])      # AnymalTerrain.yaml defaultJointAngles, dof order
    sim = dict(dt=0.005, substeps=1, iters=5, gravity=(0.0, 0.0, -9.81), contact_offset=0.02, rest_offset=0.0, max_depen_vel=100.0, erp=0.5,
               plane_mu=1.0, ground_z=0.0, cfm=1e-6, warm=1.0)
    feet = [k for k in range(len(spec.sph_body)) if spec.body_names[spec.sph_body[k]].endswith("SHANK")]
    sph_pos = np.array(spec.sph_pos).reshape(-1, 3)
    deepest, back_force, lowest = {}, {}, {}
    for walls in (True, False):
        eng = OracleEngine(spec, 1, params=sim, precision="f64")
        eng.set_ground(hf, hs, vs, border, slope_threshold=0.5, walls=walls)
        eng.root[:] = 0; eng.root[0, :3] = (3.5, 2.05, 0.57); eng.root[0, 6] = 1.0; eng.root[0, 7] = 1.0
        eng.q[:] = q0; eng.qd[:] = 0
        deep, fx, zmin = -1.0, 0.0, 1.0
        for it in range(240):
            tau = np.clip(80.0 * (q0 - eng.q) - 2.0 * eng.qd, -80, 80)
            eng.step(tau, env_mu=np.ones(1))
            bp = eng.energy(0, poses=True)[2]
            for k in feet:
                b = spec.sph_body[k]
                c = bp[b, :3] + bp[b, 3:12].reshape(3, 3) @ sph_pos[k]
                if c[2] < 0.25 - 0.01:                       # below the top of the riser
                    deep = max(deep, c[0] + spec.sph_rad[k] - 4.0)
            fx = min(fx, min(eng.netf[0, spec.body_names.index(n), 0] for n in ("LF_SHANK", "RF_SHANK")))
            zmin = min(zmin, eng.root[0, 2])
        deepest[walls], back_force[walls], lowest[walls] = deep, fx, zmin
        assert np.isfinite(eng.root).all()
    assert -0.02 < deepest[True] < 0.006, deepest            # the feet reach the face and stay out of it (measured 1.3 mm)
    assert deepest[False] > 0.03, deepest                    # no walls: through the face (7.7 cm)
    assert back_force[True] < -50.0, back_force              # (-231 N)
    assert lowest[True] > 0.4 and lowest[False] < 0.35, lowest   # stays on its feet / is thrown over

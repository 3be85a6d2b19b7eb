"""Procedural rough terrain for AnymalTerrain (host side, NumPy; runs once at env creation).

Restates `Terrain` of the reference (isaacgymenvs/tasks/anymal_terrain.py:543-673).  The sub-terrain primitives it calls
live in `isaacgym.terrain_utils` (`from isaacgym.terrain_utils import *`, anymal_terrain.py:542), a pure-Python module of
the closed Isaac Gym Preview-4 package that is NOT in the reference tree; they are re-implemented here from their
documented behaviour (same parameters, same discretisation to int16 height units, same use of NumPy's legacy global
RNG order: one `choice` per call site), and covered by property tests (tests/test_terrain.py).

The engine reads the int16 height grid directly (csrc/core/engine.hpp HeightfieldGround, the piecewise-linear surface of
the mesh `convert_heightfield_to_trimesh` would build); no triangle mesh is materialised.
"""
from __future__ import annotations

import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _bilinear_resample(z, x_src, y_src, x_dst, y_dst):
    """Linear interpolation of z[len(x_src), len(y_src)] onto the (x_dst, y_dst) grid (= scipy interp2d kind='linear')."""
    ix = np.clip(np.searchsorted(x_src, x_dst, side="right") - 1, 0, len(x_src) - 2)
    iy = np.clip(np.searchsorted(y_src, y_dst, side="right") - 1, 0, len(y_src) - 2)
    tx = ((x_dst - x_src[ix]) / (x_src[ix + 1] - x_src[ix]))[:, None]
    ty = ((y_dst - y_src[iy]) / (y_src[iy + 1] - y_src[iy]))[None, :]
    z00 = z[np.ix_(ix, iy)]; z10 = z[np.ix_(ix + 1, iy)]; z01 = z[np.ix_(ix, iy + 1)]; z11 = z[np.ix_(ix + 1, iy + 1)]
    return z00 * (1 - tx) * (1 - ty) + z10 * tx * (1 - ty) + z01 * (1 - tx) * ty + z11 * tx * ty


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None, rng=np.random):
    """Uniform noise sampled on a coarse grid, linearly up-sampled and ADDED to the height field."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    min_height = int(min_height / terrain.vertical_scale)
    max_height = int(max_height / terrain.vertical_scale)
    step = int(step / terrain.vertical_scale)
    heights_range = np.arange(min_height, max_height + step, step)
    nx = int(terrain.width * terrain.horizontal_scale / downsampled_scale)
    ny = int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = rng.choice(heights_range, (nx, ny)).astype(np.float64)
    x = np.linspace(0, terrain.width * terrain.horizontal_scale, nx)
    y = np.linspace(0, terrain.length * terrain.horizontal_scale, ny)
    xu = np.linspace(0, terrain.width * terrain.horizontal_scale, terrain.width)
    yu = np.linspace(0, terrain.length * terrain.horizontal_scale, terrain.length)
    terrain.height_field_raw += np.rint(_bilinear_resample(coarse, x, y, xu, yu)).astype(np.int16)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """Pyramid with the given slope (negative = pit) and a flat platform at the centre."""
    x = np.arange(0, terrain.width)
    y = np.arange(0, terrain.length)
    center_x = int(terrain.width / 2)
    center_y = int(terrain.length / 2)
    xx = ((center_x - np.abs(center_x - x)) / center_x).reshape(terrain.width, 1)
    yy = ((center_y - np.abs(center_y - y)) / center_y).reshape(1, terrain.length)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (max_height * xx * yy).astype(terrain.height_field_raw.dtype)
    platform_size = int(platform_size / terrain.horizontal_scale / 2)
    x1 = terrain.width // 2 - platform_size
    y1 = terrain.length // 2 - platform_size
    min_h = min(terrain.height_field_raw[x1, y1], 0)
    max_h = max(terrain.height_field_raw[x1, y1], 0)
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min_h, max_h)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps towards the centre (negative step_height = stairs down)."""
    step_width = int(step_width / terrain.horizontal_scale)
    step_height = int(step_height / terrain.vertical_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    height = 0
    start_x, stop_x, start_y, stop_y = 0, terrain.width, 0, terrain.length
    while (stop_x - start_x) > platform_size and (stop_y - start_y) > platform_size:
        start_x += step_width
        stop_x -= step_width
        start_y += step_width
        stop_y -= step_width
        height += step_height
        terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = height
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0, rng=np.random):
    """`num_rects` random rectangles of height in {-h, -h/2, h/2, h}; flat platform at the centre."""
    max_height = int(max_height / terrain.vertical_scale)
    min_size = int(min_size / terrain.horizontal_scale)
    max_size = int(max_size / terrain.horizontal_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    (i, j) = terrain.height_field_raw.shape
    height_range = [-max_height, -max_height // 2, max_height // 2, max_height]
    width_range = range(min_size, max_size, 4)
    length_range = range(min_size, max_size, 4)
    for _ in range(num_rects):
        width = rng.choice(width_range)
        length = rng.choice(length_range)
        start_i = rng.choice(range(0, i - width, 4))
        start_j = rng.choice(range(0, j - length, 4))
        terrain.height_field_raw[start_i:start_i + width, start_j:start_j + length] = rng.choice(height_range)
    x1 = (terrain.width - platform_size) // 2
    x2 = (terrain.width + platform_size) // 2
    y1 = (terrain.length - platform_size) // 2
    y2 = (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10, rng=np.random):
    """Square stones of random height over a pit of the given depth; flat platform at the centre."""
    stone_size = int(stone_size / terrain.horizontal_scale)
    stone_distance = int(stone_distance / terrain.horizontal_scale)
    max_height = int(max_height / terrain.vertical_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    height_range = np.arange(-max_height - 1, max_height, step=1)
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    start_x, start_y = 0, 0
    if terrain.length >= terrain.width:
        while start_y < terrain.length:
            stop_y = min(terrain.length, start_y + stone_size)
            start_x = rng.randint(0, stone_size)
            stop_x = max(0, start_x - stone_distance)
            terrain.height_field_raw[0:stop_x, start_y:stop_y] = rng.choice(height_range)
            while start_x < terrain.width:
                stop_x = min(terrain.width, start_x + stone_size)
                terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = rng.choice(height_range)
                start_x += stone_size + stone_distance
            start_y += stone_size + stone_distance
    else:
        while start_x < terrain.width:
            stop_x = min(terrain.width, start_x + stone_size)
            start_y = rng.randint(0, stone_size)
            stop_y = max(0, start_y - stone_distance)
            terrain.height_field_raw[start_x:stop_x, 0:stop_y] = rng.choice(height_range)
            while start_y < terrain.length:
                stop_y = min(terrain.length, start_y + stone_size)
                terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = rng.choice(height_range)
                start_y += stone_size + stone_distance
            start_x += stone_size + stone_distance
    x1 = (terrain.width - platform_size) // 2
    x2 = (terrain.width + platform_size) // 2
    y1 = (terrain.length - platform_size) // 2
    y2 = (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


class Terrain:
    """reference anymal_terrain.py:543-673 (same attribute names).  `seed` fixes the NumPy stream so that every rank
    of a multi-GPU job builds the same terrain (the reference relies on the global `np.random` seeded by set_seed)."""

    def __init__(self, cfg, num_robots, seed=0) -> None:
        self.type = cfg["terrainType"]
        if self.type in ["none", "plane"]:
            return
        self.rng = np.random.RandomState(seed)
        self.horizontal_scale = 0.1
        self.vertical_scale = 0.005
        self.border_size = 20
        self.env_length = cfg["mapLength"]
        self.env_width = cfg["mapWidth"]
        self.proportions = [np.sum(cfg["terrainProportions"][:i + 1]) for i in range(len(cfg["terrainProportions"]))]
        self.env_rows = cfg["numLevels"]
        self.env_cols = cfg["numTerrains"]
        self.num_maps = self.env_rows * self.env_cols
        self.num_per_env = int(num_robots / self.num_maps)
        self.env_origins = np.zeros((self.env_rows, self.env_cols, 3))
        self.width_per_env_pixels = int(self.env_width / self.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / self.horizontal_scale)
        self.border = int(self.border_size / self.horizontal_scale)
        self.tot_cols = int(self.env_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(self.env_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg["curriculum"]:
            self.curiculum(num_robots, num_terrains=self.env_cols, num_levels=self.env_rows)
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw
        # anymal_terrain.py:576 hands this to the height-field -> triangle-mesh conversion; the engine's ground query applies it
        self.slope_threshold = float(cfg.get("slopeTreshold", 0.0) or 0.0)

    def _new_sub(self):
        return SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                          vertical_scale=self.vertical_scale, horizontal_scale=self.horizontal_scale)

    def _place(self, terrain, i, j):
        start_x = self.border + i * self.length_per_env_pixels
        end_x = self.border + (i + 1) * self.length_per_env_pixels
        start_y = self.border + j * self.width_per_env_pixels
        end_y = self.border + (j + 1) * self.width_per_env_pixels
        self.height_field_raw[start_x:end_x, start_y:end_y] = terrain.height_field_raw
        env_origin_x = (i + 0.5) * self.env_length
        env_origin_y = (j + 0.5) * self.env_width
        x1 = int((self.env_length / 2. - 1) / self.horizontal_scale)
        x2 = int((self.env_length / 2. + 1) / self.horizontal_scale)
        y1 = int((self.env_width / 2. - 1) / self.horizontal_scale)
        y2 = int((self.env_width / 2. + 1) / self.horizontal_scale)
        env_origin_z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * self.vertical_scale
        self.env_origins[i, j] = [env_origin_x, env_origin_y, env_origin_z]

    # ---- which sub-terrain goes on tile (i, j).  Both generators walk the tiles in the reference's order and draw from the NumPy stream in the
    # reference's order (anymal_terrain.py:577-673): the terrain of a seed is the same on every rank and the same as the reference's mesh.
    # A generator is a table of (upper edge of the selector, recipe); the first row whose edge exceeds the selector builds the tile.
    def _build(self, table, selector, terrain, *args):
        for edge, recipe in table:
            if selector < edge:
                recipe(terrain, *args)
                return

    def randomized_terrain(self):  # :577-619: one uniform draw per tile picks the kind, further draws its parameters
        rng = self.rng
        slopes, noise = [-0.3, -0.2, 0, 0.2, 0.3], dict(min_height=-0.1, max_height=0.1, step=0.05, downsampled_scale=0.2)

        def slope_tile(t):
            rough = rng.choice([0, 1])                                     # (drawn before the slope, as the reference does)
            pyramid_sloped_terrain(t, rng.choice(slopes))
            if rough:
                random_uniform_terrain(t, rng=rng, **noise)
        table = ((0.1, slope_tile),
                 (0.6, lambda t: pyramid_stairs_terrain(t, step_width=0.31, step_height=rng.choice([-0.15, 0.15]), platform_size=3.)),
                 (1.0, lambda t: discrete_obstacles_terrain(t, 0.15, 1., 2., 40, platform_size=3., rng=rng)))
        for k in range(self.num_maps):
            (i, j) = np.unravel_index(k, (self.env_rows, self.env_cols))
            terrain = self._new_sub()
            self._build(table, rng.uniform(0, 1), terrain)
            self._place(terrain, i, j)

    def curiculum(self, num_robots, num_terrains, num_levels):  # :621-673 (the method name is spelled as the reference spells it)
        """column j -> kind through the cumulative `terrainProportions` (selector c = j / num_terrains), row i -> difficulty d = i / num_levels:
        smooth slope up to 0.4 d (sunk for c < 0.05) | rough slope (sunk for c < 0.15) | stairs of 0.05 + 0.175 d (down below proportions[2]) |
        40 discrete obstacles up to 0.025 + 0.15 d | stepping stones of 2 - 1.8 d m"""
        rng, prop = self.rng, self.proportions
        sign = lambda below: -1 if below else 1
        table = (
            (prop[0], lambda t, d, c: pyramid_sloped_terrain(t, slope=d * 0.4 * sign(c < 0.05), platform_size=3.)),
            (prop[1], lambda t, d, c: (pyramid_sloped_terrain(t, slope=d * 0.4 * sign(c < 0.15), platform_size=3.),
                                       random_uniform_terrain(t, min_height=-0.1, max_height=0.1, step=0.025, downsampled_scale=0.2, rng=rng))),
            (prop[3], lambda t, d, c: pyramid_stairs_terrain(t, step_width=0.31, step_height=(0.05 + 0.175 * d) * sign(c < prop[2]), platform_size=3.)),
            (prop[4], lambda t, d, c: discrete_obstacles_terrain(t, 0.025 + d * 0.15, 1., 2., 40, platform_size=3., rng=rng)),
            (np.inf, lambda t, d, c: stepping_stones_terrain(t, stone_size=2 - 1.8 * d, stone_distance=0.1, max_height=0., platform_size=3., rng=rng)),
        )
        for j in range(num_terrains):
            for i in range(num_levels):
                terrain = self._new_sub()
                self._build(table, j / num_terrains, terrain, i / num_levels, j / num_terrains)
                self._place(terrain, i, j)

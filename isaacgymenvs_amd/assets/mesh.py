"""Mesh collision geometry -> contact spheres.

The engine's narrow phase knows spheres on the robot side (against the ground plane / height field, and against the manipulated
object's exact box / capsule / ellipsoid).  Robots whose collision shapes are primitives are sampled analytically (capsules along
their axis, boxes on a grid: tools/compile_models.py).  Robots whose collision shapes are triangle meshes -- the Allegro hand of
reference isaacgymenvs/tasks/allegro_hand.py:216, assets/urdf/kuka_allegro_description/allegro_touch_sensor.urdf -- go through this
module: PhysX collides a mesh shape as its CONVEX HULL (gymapi.AssetOptions: convex decomposition is off unless vhacd_enabled), so
the hull is what gets sampled:

  * spheres of one radius r per shape, r = min(0.95 x the radius of the largest ball the hull contains, r_cap): (nearly) the largest
    sphere that fits the thin direction, so a slender link becomes a string of spheres along its axis, the way a capsule is sampled;
  * every sphere lies INSIDE the hull and is tangent to it (centre = surface point - r * outward normal, then pushed off the
    other faces it would cross), i.e. the spheres' union never exceeds the hull and touches it wherever a sphere sits;
  * centres are picked from a dense candidate set by farthest-point sampling until every candidate lies within `spacing` / 2 * r of
    a picked one (neighbours then sit 0.8 r .. 1.6 r apart: the surface of the union dips <= 0.4 r between them) or `max_count` is
    reached; the order is the
    sampling order -- the engine admits a body's first few touching spheres as its contact manifold, and a spread-out order makes
    them span the contact patch (same rule as the Shadow Hand's boxes).

numpy + scipy.spatial.ConvexHull only; deterministic (no random numbers)."""
from __future__ import annotations

import numpy as np


def load_obj(path):
    """-> vertices [n, 3] (float64), triangles [m, 3] (int; polygons are fanned).  Wavefront OBJ: 'v' and 'f' records only."""
    V, F = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                V.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(V) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    F.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(V, float).reshape(-1, 3), np.asarray(F, int).reshape(-1, 3)


def load_stl(path):
    """-> vertices [n, 3], triangles [m, 3] of a binary or ASCII STL (vertices are not merged: the hull does not care)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:5].lower() == b"solid" and b"facet" in data[:1000]:
        V = [[float(x) for x in ln.split()[1:4]] for ln in data.decode("ascii", "ignore").splitlines() if ln.strip().startswith("vertex")]
        V = np.asarray(V, float).reshape(-1, 3)
    else:
        n = int(np.frombuffer(data[80:84], "<u4")[0])
        rec = np.frombuffer(data[84:84 + 50 * n], np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
        V = rec["v"].reshape(-1, 3).astype(float)
    return V, np.arange(len(V)).reshape(-1, 3)


def load_mesh(path):
    return load_stl(path) if path.lower().endswith(".stl") else load_obj(path)


def _hull(V):
    from scipy.spatial import ConvexHull
    h = ConvexHull(V)
    # h.equations: [normal (outward, unit), offset] with normal . x + offset <= 0 inside
    return h


def _surface_candidates(V, hull, step):
    """points on the hull surface (with their face index), a barycentric grid per triangle fine enough for `step` spacing"""
    pts, face = [], []
    for fi, tri in enumerate(hull.simplices):
        a, b, c = V[tri]
        L = max(np.linalg.norm(b - a), np.linalg.norm(c - a), np.linalg.norm(c - b))
        n = max(1, int(np.ceil(L / step)))
        for i in range(n + 1):
            for j in range(n + 1 - i):
                u, v = i / n, j / n
                pts.append(a + u * (b - a) + v * (c - a))
                face.append(fi)
    return np.asarray(pts), np.asarray(face)


def hull_spheres(V, r_cap=0.015, spacing=1.6, max_count=48, r_min=0.003, fit=0.95):
    """Spheres inscribed in the convex hull of the points V (see the module docstring) -> centres [k, 3], radius (float).
    A hull that contains no ball of radius r_min / fit yields no spheres."""
    V = np.asarray(V, float)
    hull = _hull(V)
    eq = hull.equations                      # [m, 4]
    # the largest ball the hull contains (Chebyshev radius, a linear programme over centre and radius): a sphere of radius r has room
    # for its centre exactly where the depth below every face is >= r, and that set is non-empty for r below this radius
    from scipy.optimize import linprog
    m = len(eq)
    lp = linprog(c=[0, 0, 0, -1.0], A_ub=np.hstack([eq[:, :3], np.ones((m, 1))]), b_ub=-eq[:, 3], bounds=[(None, None)] * 3 + [(0, None)])
    if not lp.success:
        return np.zeros((0, 3)), 0.0
    r = min(fit * float(lp.x[3]), r_cap)
    if r < r_min:
        return np.zeros((0, 3)), 0.0
    P, face = _surface_candidates(V, hull, 0.5 * spacing * r)
    C = P - r * eq[face, :3]
    # a centre r below its own face can still be closer than r to a neighbouring face (near edges and corners): project it onto the
    # violated half-spaces in turn (cyclic projections onto convex sets with a common interior point converge)
    for _ in range(200):
        viol = C @ eq[:, :3].T + eq[:, 3] + r                 # > 0: the sphere crosses that face
        worst = np.argmax(viol, axis=1)
        amt = viol[np.arange(len(C)), worst]
        bad = amt > 1e-7
        if not bad.any():
            break
        C[bad] -= amt[bad, None] * eq[worst[bad], :3]
    dist = C @ eq[:, :3].T + eq[:, 3]
    C = C[(dist + r).max(axis=1) <= 1e-6]
    if len(C) == 0:
        return np.zeros((0, 3)), 0.0
    # merge coincident candidates (opposite faces of a link of thickness 2 r give the same centre line)
    key = np.round(C / (0.05 * r)).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    C = C[np.sort(first)]
    # farthest-point sampling: start with the candidate farthest from the centroid
    order = [int(np.argmax(np.linalg.norm(C - C.mean(0), axis=1)))]
    d = np.linalg.norm(C - C[order[0]], axis=1)
    while len(order) < min(max_count, len(C)):
        k = int(np.argmax(d))
        if d[k] < 0.5 * spacing * r:
            break
        order.append(k)
        d = np.minimum(d, np.linalg.norm(C - C[k], axis=1))
    return C[order], float(r)


def hull_mass_properties(V, density):
    """Mass, centre of mass and inertia tensor about it of the convex hull of the points V at uniform density (what the simulator derives
    for a link that has a collision mesh and no <inertial>): signed tetrahedra from the hull's centroid to its faces."""
    V = np.asarray(V, float)
    hull = _hull(V)
    c0 = V[hull.vertices].mean(axis=0)
    vol, first = 0.0, np.zeros(3)
    second = np.zeros((3, 3))                 # integral of x x^T over the body, about c0
    canon = (np.ones((3, 3)) + np.eye(3)) / 120.0
    for tri in hull.simplices:
        a, b, c = V[tri] - c0
        A = np.stack([a, b, c], axis=1)       # columns: the tetrahedron's edge vectors from c0
        d = abs(float(np.linalg.det(A)))      # 6 x volume; the hull is convex and c0 inside, so every tetrahedron counts positively
        vol += d / 6.0
        first += d / 24.0 * (a + b + c)
        second += d * (A @ canon @ A.T)
    if vol <= 0.0:
        return 0.0, c0, np.zeros((3, 3))
    com_rel = first / vol
    second_c = second - vol * np.outer(com_rel, com_rel)
    I = density * (np.trace(second_c) * np.eye(3) - second_c)
    return density * vol, c0 + com_rel, I


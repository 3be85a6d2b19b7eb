"""TEST INFRASTRUCTURE (never imported by the product): the triangle mesh the reference's AnymalTerrain hands to PhysX, restated, and
an exact height query on it -- the yardstick for the engine's height-field ground (csrc/core/engine.hpp HeightfieldGround).

The mesh comes from `isaacgym.terrain_utils.convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale,
slope_threshold)` (called at reference tasks/anymal_terrain.py:576).  That module ships inside the closed Isaac Gym package (Preview 4),
not in /root/reference: PARITY UNPINNED -- the algorithm below is restated from the published Python source of that package:
  * vertex (i, j) at (i * hscale, j * hscale, h[i, j] * vscale); each cell split along the (i, j)-(i+1, j+1) diagonal into the triangles
    ((i,j), (i+1,j+1), (i+1,j)) and ((i,j), (i,j+1), (i+1,j+1));
  * slope correction: with thr = slope_threshold * hscale / vscale (raw height units), a vertex that lies more than thr below its +x
    (-x) neighbour moves one grid step in +x (-x), likewise in y, and along the diagonal (only in a direction that did not move
    already): the lower vertex of a riser slides under the upper one, the riser becomes a vertical wall at the UPPER vertex's position
    and the lower tread is stretched up to it."""
import numpy as np


def trimesh_vertices(hf, hscale, vscale, slope_threshold):
    """-> xx, yy [rows, cols] vertex positions after the slope correction (grid units * hscale), zz heights (metres)."""
    hf = np.asarray(hf, np.float64)
    rows, cols = hf.shape
    xx, yy = np.meshgrid(np.arange(rows) * hscale, np.arange(cols) * hscale, indexing="ij")
    xx, yy = xx.astype(np.float64), yy.astype(np.float64)
    if slope_threshold is not None:
        thr = slope_threshold * hscale / vscale
        mx, my, mc = np.zeros((rows, cols)), np.zeros((rows, cols)), np.zeros((rows, cols))
        mx[:-1, :] += (hf[1:, :] - hf[:-1, :]) > thr
        mx[1:, :] -= (hf[:-1, :] - hf[1:, :]) > thr
        my[:, :-1] += (hf[:, 1:] - hf[:, :-1]) > thr
        my[:, 1:] -= (hf[:, :-1] - hf[:, 1:]) > thr
        mc[:-1, :-1] += (hf[1:, 1:] - hf[:-1, :-1]) > thr
        mc[1:, 1:] -= (hf[:-1, :-1] - hf[1:, 1:]) > thr
        xx = xx + (mx + mc * (mx == 0)) * hscale
        yy = yy + (my + mc * (my == 0)) * hscale
    return xx, yy, hf * vscale


def trimesh_height(hf, hscale, vscale, slope_threshold, px, py, reach=2):
    """Exact height of the top surface of that mesh under the points (px, py) (grid frame, no border shift): the highest triangle whose
    xy projection contains the point, searched over the cells within `reach` of the point's own cell."""
    xx, yy, zz = trimesh_vertices(hf, hscale, vscale, slope_threshold)
    rows, cols = zz.shape
    px, py = np.asarray(px, np.float64), np.asarray(py, np.float64)
    ci = np.clip(np.floor(px / hscale).astype(int), 0, rows - 2)
    cj = np.clip(np.floor(py / hscale).astype(int), 0, cols - 2)
    best = np.full(px.shape, -np.inf)
    for di in range(-reach, reach + 1):
        for dj in range(-reach, reach + 1):
            i = np.clip(ci + di, 0, rows - 2); j = np.clip(cj + dj, 0, cols - 2)
            for tri in (((0, 0), (1, 1), (1, 0)), ((0, 0), (0, 1), (1, 1))):
                (a0, b0), (a1, b1), (a2, b2) = tri
                x0, y0, z0 = xx[i + a0, j + b0], yy[i + a0, j + b0], zz[i + a0, j + b0]
                x1, y1, z1 = xx[i + a1, j + b1], yy[i + a1, j + b1], zz[i + a1, j + b1]
                x2, y2, z2 = xx[i + a2, j + b2], yy[i + a2, j + b2], zz[i + a2, j + b2]
                den = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2)
                ok = np.abs(den) > 1e-12
                d = np.where(ok, den, 1.0)
                l0 = ((y1 - y2) * (px - x2) + (x2 - x1) * (py - y2)) / d
                l1 = ((y2 - y0) * (px - x2) + (x0 - x2) * (py - y2)) / d
                l2 = 1.0 - l0 - l1
                inside = ok & (l0 >= -1e-9) & (l1 >= -1e-9) & (l2 >= -1e-9)
                z = l0 * z0 + l1 * z1 + l2 * z2
                best = np.where(inside & (z > best), z, best)
    return best


def snapped_height(hf, hscale, vscale, slope_threshold, px, py):
    """The engine's surface (csrc/core/engine.hpp HeightfieldGround::query), restated: the cell's own two triangles after every cell
    edge that rises by more than thr has been levelled to its lower end (x edges first, then y edges)."""
    hf = np.asarray(hf, np.float64)
    rows, cols = hf.shape
    gx, gy = np.asarray(px, np.float64) / hscale, np.asarray(py, np.float64) / hscale
    i = np.clip(np.floor(gx).astype(int), 0, rows - 2); j = np.clip(np.floor(gy).astype(int), 0, cols - 2)
    fx, fy = np.clip(gx - i, 0, 1), np.clip(gy - j, 0, 1)
    h00, h10, h01, h11 = hf[i, j], hf[i + 1, j], hf[i, j + 1], hf[i + 1, j + 1]
    if slope_threshold is not None:
        thr = slope_threshold * hscale / vscale
        s = np.abs(h10 - h00) > thr; m = np.minimum(h00, h10); h00, h10 = np.where(s, m, h00), np.where(s, m, h10)
        s = np.abs(h11 - h01) > thr; m = np.minimum(h01, h11); h01, h11 = np.where(s, m, h01), np.where(s, m, h11)
        s = np.abs(h01 - h00) > thr; m = np.minimum(h00, h01); h00, h01 = np.where(s, m, h00), np.where(s, m, h01)
        s = np.abs(h11 - h10) > thr; m = np.minimum(h10, h11); h10, h11 = np.where(s, m, h10), np.where(s, m, h11)
    lower = fx >= fy
    dzx = np.where(lower, h10 - h00, h11 - h01); dzy = np.where(lower, h11 - h10, h01 - h00)
    return (h00 + dzx * fx + dzy * fy) * vscale


def _closest_on_triangle(p, a, b, c):
    """closest point of the triangles (a, b, c) [n, 3] to the points p [n, 3] (Ericson, Real-Time Collision Detection 5.1.5), vectorised"""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = np.einsum("ij,ij->i", ab, ap), np.einsum("ij,ij->i", ac, ap)
    bp = p - b
    d3, d4 = np.einsum("ij,ij->i", ab, bp), np.einsum("ij,ij->i", ac, bp)
    cp = p - c
    d5, d6 = np.einsum("ij,ij->i", ab, cp), np.einsum("ij,ij->i", ac, cp)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out = np.empty_like(p)
    done = np.zeros(len(p), bool)

    def put(mask, val):
        m = mask & ~done
        out[m] = val[m]
        done[m] = True
    with np.errstate(divide="ignore", invalid="ignore"):
        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), c)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None] * (c - b))
        den = va + vb + vc
        v, w = vb / den, vc / den
        put(np.ones(len(p), bool), a + v[:, None] * ab + w[:, None] * ac)
    return out


def trimesh_closest(hf, hscale, vscale, slope_threshold, pts, reach=2):
    """Closest point of the (corrected) triangle mesh to every point of pts [n, 3] (grid frame), searched over the cells within `reach` of
    the point's own cell -> (distance [n], closest point [n, 3]).  The yardstick for HeightfieldGround::contact's riser walls."""
    xx, yy, zz = trimesh_vertices(hf, hscale, vscale, slope_threshold)
    rows, cols = zz.shape
    pts = np.asarray(pts, np.float64)
    ci = np.clip(np.floor(pts[:, 0] / hscale).astype(int), 0, rows - 2)
    cj = np.clip(np.floor(pts[:, 1] / hscale).astype(int), 0, cols - 2)
    best = np.full(len(pts), np.inf)
    bestp = np.zeros_like(pts)
    for di in range(-reach, reach + 1):
        for dj in range(-reach, reach + 1):
            i = np.clip(ci + di, 0, rows - 2); j = np.clip(cj + dj, 0, cols - 2)
            for tri in (((0, 0), (1, 1), (1, 0)), ((0, 0), (0, 1), (1, 1))):
                v = [np.stack([xx[i + a, j + b], yy[i + a, j + b], zz[i + a, j + b]], axis=1) for a, b in tri]
                q = _closest_on_triangle(pts, v[0], v[1], v[2])
                d = np.linalg.norm(pts - q, axis=1)
                better = d < best
                best = np.where(better, d, best)
                bestp[better] = q[better]
    return best, bestp

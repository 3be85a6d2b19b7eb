"""Software viewer: what `VecTask.render(mode="rgb_array")` returns and `record_frames` writes.

The reference draws through Isaac Gym's OpenGL viewer and grabs the virtual display for `rgb_array` (reference tasks/base/vec_task.py:457-512,
`gym.create_viewer` / `draw_viewer` / `write_viewer_image_to_file`); neither a display nor OpenGL exists beside this engine, so the frame is
rasterised on the host from the rigid-body state tensor: a pinhole camera that follows one env, a checkered ground plane, and every body
drawn by the spheres the engine collides it with (the model's contact samples; a body without any gets a small marker) plus a string of
beads to its parent body, depth-sorted and
shaded by a fixed light.  A diagnostic view of the simulated state, not a renderer of the asset's visual meshes.
"""
import struct
import zlib

import numpy as np


def _look_at(eye, target, up=(0.0, 0.0, 1.0)):
    f = np.asarray(target, np.float64) - np.asarray(eye, np.float64)
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    return np.stack([r, u, f])          # rows: camera x (right), y (up), z (forward)


def quat_to_matrix(q):
    """xyzw quaternions [n, 4] -> rotation matrices [n, 3, 3]"""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


class SoftwareViewer:
    """width x height RGB frames of one env.  cam_offset: camera position relative to the followed point, fov: vertical field of view (deg)."""

    def __init__(self, width=640, height=480, cam_offset=(2.2, -2.6, 1.4), fov=45.0, ground_z=0.0):
        self.width, self.height = int(width), int(height)
        self.cam_offset = np.asarray(cam_offset, np.float64)
        self.focal = 0.5 * self.height / np.tan(np.radians(fov) / 2)
        self.ground_z = ground_z
        jj, ii = np.meshgrid(np.arange(self.width), np.arange(self.height))
        self._px = (jj - 0.5 * self.width + 0.5) / self.focal            # camera-frame ray directions (x right, y up, z forward = 1)
        self._py = -(ii - 0.5 * self.height + 0.5) / self.focal

    def spheres_of(self, spec, body_state):
        """world centres, radii and colours of the spheres that stand for the bodies of one env; body_state [nb, 13]"""
        pos, R = body_state[:, :3], quat_to_matrix(body_state[:, 3:7])
        nb = body_state.shape[0]
        sb = np.asarray(spec.sph_body, int)
        sp = np.asarray(spec.sph_pos, np.float64).reshape(-1, 3)
        sr = np.asarray(spec.sph_rad, np.float64)
        keep = sb < nb
        sb, sp, sr = sb[keep], sp[keep], sr[keep]
        c = pos[sb] + np.einsum("nij,nj->ni", R[sb], sp)
        bare = np.setdiff1d(np.arange(nb), sb)                             # bodies the engine has no collision sample on
        c = np.concatenate([c, pos[bare]])
        r = np.concatenate([sr, np.full(len(bare), 0.02)])
        b = np.concatenate([sb, bare])
        par = np.asarray(getattr(spec, "parent", []), int)                 # "bones": small beads from every body's origin to its parent's
        if len(par) >= nb:
            for k in range(1, nb):
                if par[k] >= 0:
                    seg = pos[par[k]][None, :] + np.linspace(0.0, 1.0, 7)[1:-1, None] * (pos[k] - pos[par[k]])[None, :]
                    c = np.concatenate([c, seg]); r = np.concatenate([r, np.full(len(seg), 0.018)]); b = np.concatenate([b, np.full(len(seg), k)])
        hue = (b * 0.61803398875) % 1.0                                    # one colour per body
        col = np.stack([0.55 + 0.4 * np.cos(2 * np.pi * (hue + k / 3.0)) for k in range(3)], -1)
        return c, r, col

    def draw(self, centres, radii, colours, focus):
        """-> uint8 [height, width, 3]"""
        # the camera keeps its direction and backs off to where the drawn bodies fill about a third of the frame (a hand is 0.2 m, ANYmal 1 m)
        centres = np.asarray(centres, np.float64)
        extent = float(np.max(np.linalg.norm(centres - np.asarray(focus, np.float64), axis=1) + np.asarray(radii))) if len(centres) else 1.0
        dist = min(max(3.2 * extent, 0.5), 8.0)
        eye = np.asarray(focus, np.float64) + self.cam_offset / np.linalg.norm(self.cam_offset) * dist
        Rc = _look_at(eye, focus)
        H, W = self.height, self.width
        # ground: intersect every pixel's ray with z = ground_z, checker it
        dirs = self._px[..., None] * Rc[0] + self._py[..., None] * Rc[1] + Rc[2]
        img = np.empty((H, W, 3))
        sky = np.array([0.62, 0.75, 0.92])
        t = (self.ground_z - eye[2]) / np.where(np.abs(dirs[..., 2]) > 1e-9, dirs[..., 2], 1e-9)
        hit = (t > 0) & (dirs[..., 2] < -1e-9)
        gx, gy = eye[0] + t * dirs[..., 0], eye[1] + t * dirs[..., 1]
        check = ((np.floor(gx) + np.floor(gy)) % 2 == 0)
        fade = np.clip(1.0 - t / 40.0, 0.0, 1.0)[..., None]
        ground = np.where(check[..., None], np.array([0.52, 0.54, 0.50]), np.array([0.40, 0.42, 0.39]))
        img[:] = sky
        img[hit] = (ground * fade + sky * (1 - fade))[hit]
        depth = np.where(hit, t * np.linalg.norm(dirs, axis=-1), np.inf)
        # spheres, far to near; each shaded as a lit ball
        pc = (np.asarray(centres, np.float64) - eye) @ Rc.T
        light = np.array([0.35, 0.5, -0.8]); light /= np.linalg.norm(light)     # camera frame, from the upper left behind the camera
        for k in np.argsort(-pc[:, 2]):
            x, y, z = pc[k]
            if z <= 0.05:
                continue
            rad = float(radii[k])
            u0, v0, rp = 0.5 * W + self.focal * x / z, 0.5 * H - self.focal * y / z, self.focal * rad / z
            j0, j1 = int(max(np.floor(u0 - rp), 0)), int(min(np.ceil(u0 + rp) + 1, W))
            i0, i1 = int(max(np.floor(v0 - rp), 0)), int(min(np.ceil(v0 + rp) + 1, H))
            if j0 >= j1 or i0 >= i1:
                continue
            du = (np.arange(j0, j1) + 0.5 - u0) / max(rp, 1e-9)
            dv = (np.arange(i0, i1) + 0.5 - v0) / max(rp, 1e-9)
            rr = du[None, :] ** 2 + dv[:, None] ** 2
            inside = rr <= 1.0
            nz = -np.sqrt(np.clip(1.0 - rr, 0.0, 1.0))                          # surface normal, camera frame (towards the camera: -z)
            zs = z + nz * rad
            sub_d = depth[i0:i1, j0:j1]
            vis = inside & (zs < sub_d)
            lam = np.clip(-(du[None, :] * light[0] - dv[:, None] * light[1] + nz * light[2]), 0.0, 1.0)
            shade = (0.35 + 0.65 * lam)[..., None] * np.asarray(colours[k])
            sub = img[i0:i1, j0:j1]
            sub[vis] = shade[vis]
            sub_d[vis] = zs[vis]
        return (np.clip(img, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)


def write_png(path, img):
    """uint8 [h, w, 3] -> an 8-bit RGB PNG (zlib + struct: no imaging library in the image)"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[i].tobytes() for i in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) +
                chunk(b"IEND", b""))

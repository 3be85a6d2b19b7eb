"""Model compiler: MJCF / URDF robot description -> flat ``ModelSpec``.

This replaces what the closed ``gym.load_asset`` does for the reference tasks
(call sites: reference ``isaacgymenvs/tasks/ant.py:135-157``, ``humanoid.py:138-157``,
``cartpole.py:72-88``, ``anymal_terrain.py:214-231``).  The output is a flat,
array-only description of one articulated robot that both the CPU oracle
(``oracle/physics.c``, table driven) and the HIP code generator
(``isaacgymenvs_amd/codegen.py``, compile-time specialised) consume.

Conventions (documented design decisions):
  * bodies in depth-first document order (= Isaac Gym rigid-body order),
    DoFs in joint traversal order (= Isaac Gym DoF order);
  * quaternions are xyzw everywhere (reference ``torch_jit_utils.py:48``);
  * body frame pose = parent frame * T(pos, quat) * prod_j [rotate/translate
    about joint j's axis through its anchor], joints given in the child frame
    (MuJoCo semantics; URDF joints have anchor = child origin);
  * link inertia from collision geoms x density when no explicit inertial;
  * welded children (no joint) are merged into their parent for dynamics but
    remain addressable as "api bodies" (rigid-body tensor rows).
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX, GEOM_CYLINDER = 0, 1, 2, 3
JOINT_HINGE, JOINT_SLIDE = 0, 1


# ----------------------------------------------------------------------------
# small rotation helpers (numpy, xyzw)
# ----------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_to_mat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def mat_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def quat_from_z_to(v):
    """Quaternion rotating +z onto unit vector v."""
    v = np.asarray(v, float)
    v = v / np.linalg.norm(v)
    z = np.array([0.0, 0.0, 1.0])
    c = float(np.dot(z, v))
    if c > 1 - 1e-12:
        return np.array([0.0, 0.0, 0.0, 1.0])
    if c < -1 + 1e-12:
        return np.array([1.0, 0.0, 0.0, 0.0])
    ax = np.cross(z, v)
    ax /= np.linalg.norm(ax)
    ang = math.acos(c)
    return np.array([*(ax * math.sin(ang / 2)), math.cos(ang / 2)])


def rpy_to_quat(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
    return np.array([
        sr * cp * cy - cr * sp * sy,
        cr * sp * cy + sr * cp * sy,
        cr * cp * sy - sr * sp * cy,
        cr * cp * cy + sr * sp * sy,
    ])


# ----------------------------------------------------------------------------
# geom mass properties (uniform density), returned about the geom's own frame
# ----------------------------------------------------------------------------
def geom_mass_inertia(gtype, size, density):
    """-> (mass, diag inertia about geom COM in geom axes)."""
    if gtype == GEOM_SPHERE:
        r = size[0]
        m = density * 4.0 / 3.0 * math.pi * r ** 3
        i = 0.4 * m * r * r
        return m, np.array([i, i, i])
    if gtype == GEOM_CAPSULE:
        r, l = size[0], size[1]  # l = half length of the cylinder part, axis z
        mc = density * math.pi * r * r * 2 * l
        mh = density * 2.0 / 3.0 * math.pi * r ** 3
        m = mc + 2 * mh
        izz = 0.5 * mc * r * r + 2 * mh * 0.4 * r * r
        ixx = mc * (r * r / 4 + l * l / 3) + 2 * mh * (0.4 * r * r + l * l + 0.75 * r * l)
        return m, np.array([ixx, ixx, izz])
    if gtype == GEOM_CYLINDER:
        r, l = size[0], size[1]
        m = density * math.pi * r * r * 2 * l
        return m, np.array([m * (r * r / 4 + l * l / 3), m * (r * r / 4 + l * l / 3), 0.5 * m * r * r])
    if gtype == GEOM_BOX:
        hx, hy, hz = size[:3]
        m = density * 8 * hx * hy * hz
        return m, np.array([m / 3 * (hy * hy + hz * hz), m / 3 * (hx * hx + hz * hz), m / 3 * (hx * hx + hy * hy)])
    raise ValueError(gtype)


# ----------------------------------------------------------------------------
# intermediate tree
# ----------------------------------------------------------------------------
@dataclass
class _Joint:
    name: str
    jtype: int
    axis: np.ndarray
    anchor: np.ndarray
    lower: float = 0.0
    upper: float = 0.0
    limited: bool = False
    armature: float = 0.0
    damping: float = 0.0
    stiffness: float = 0.0
    springref: float = 0.0
    frictionloss: float = 0.0
    effort: float = 1e30
    velocity: float = 1e30


@dataclass
class _Geom:
    name: str
    gtype: int
    pos: np.ndarray
    quat: np.ndarray
    size: np.ndarray
    friction: float = 1.0
    density: float = 1000.0
    mass: float | None = None
    contype: int = 1
    conaffinity: int = 1


@dataclass
class _Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    joints: list = field(default_factory=list)
    geoms: list = field(default_factory=list)
    free: bool = False
    inertial: tuple | None = None  # (mass, com, quat, diag) explicit
    gravity: bool = True


@dataclass
class ModelSpec:
    name: str
    fixed_base: bool
    # dynamic bodies (welds merged)
    body_names: list
    parent: np.ndarray      # [nb] int
    bpos: np.ndarray        # [nb,3] in parent frame
    bquat: np.ndarray       # [nb,4] xyzw
    mass: np.ndarray        # [nb]
    com: np.ndarray         # [nb,3] body frame
    inertia: np.ndarray     # [nb,6] xx,yy,zz,xy,xz,yz about COM, body axes
    # dofs
    dof_names: list
    dof_body: np.ndarray    # [nd]
    dof_type: np.ndarray    # [nd]
    dof_axis: np.ndarray    # [nd,3] body frame (unit)
    dof_anchor: np.ndarray  # [nd,3] body frame
    dof_lower: np.ndarray
    dof_upper: np.ndarray
    dof_limited: np.ndarray
    dof_armature: np.ndarray
    dof_damping: np.ndarray
    dof_stiffness: np.ndarray
    dof_springref: np.ndarray
    dof_effort: np.ndarray
    dof_velocity: np.ndarray
    # collision geoms (attached to dynamic bodies)
    geom_names: list
    geom_body: np.ndarray
    geom_type: np.ndarray
    geom_pos: np.ndarray    # [ng,3]
    geom_quat: np.ndarray   # [ng,4]
    geom_size: np.ndarray   # [ng,3]
    geom_friction: np.ndarray
    # derived: contact spheres vs ground (sphere centres, capsule end caps, box corners)
    sph_body: np.ndarray
    sph_pos: np.ndarray
    sph_rad: np.ndarray
    sph_friction: np.ndarray
    sph_geom: np.ndarray
    # actuators (MJCF <motor>) in file order
    act_names: list
    act_dof: np.ndarray
    act_gear: np.ndarray
    # api bodies (before weld merge): name, dynamic body index, local offset pose
    api_body_names: list
    api_body_dyn: np.ndarray
    api_body_pos: np.ndarray
    api_body_quat: np.ndarray

    @property
    def nb(self):
        return len(self.parent)

    @property
    def nd(self):
        return len(self.dof_body)

    @property
    def nv(self):
        return self.nd + (0 if self.fixed_base else 6)

    def total_mass(self):
        return float(self.mass.sum())

    # ---- (de)serialisation: compiled models are committed under isaacgymenvs_amd/models
    def to_json(self):
        d = {}
        for k, v in self.__dict__.items():
            d[k] = v.tolist() if isinstance(v, np.ndarray) else v
        return d

    @staticmethod
    def from_json(d):
        ints = {"parent", "dof_body", "dof_type", "dof_limited", "geom_body", "geom_type", "sph_body", "sph_geom",
                "act_dof", "api_body_dyn"}
        kw = {}
        for k, v in d.items():
            if isinstance(v, list) and not (k.endswith("names")):
                kw[k] = np.array(v, dtype=np.int32 if k in ints else np.float64)
                if kw[k].size == 0:
                    shape = {"geom_pos": (0, 3), "geom_quat": (0, 4), "geom_size": (0, 3), "sph_pos": (0, 3),
                             "dof_axis": (0, 3), "dof_anchor": (0, 3)}.get(k)
                    if shape:
                        kw[k] = kw[k].reshape(shape)
            else:
                kw[k] = v
        return ModelSpec(**kw)

    def save(self, path):
        with open(path, "w") as f:
            json.dump(self.to_json(), f, indent=0)

    @staticmethod
    def load(path):
        with open(path) as f:
            return ModelSpec.from_json(json.load(f))


# ----------------------------------------------------------------------------
# MJCF
# ----------------------------------------------------------------------------
def _floats(s, n=None):
    v = np.array([float(x) for x in s.split()], dtype=float)
    if n is not None and len(v) != n:
        raise ValueError(f"expected {n} floats in '{s}'")
    return v


class _MjcfDefaults:
    """MJCF <default> class tree: attribute dict per (class, element tag)."""

    def __init__(self):
        self.classes = {"main": {}}
        self.parent = {"main": None}

    def load(self, node, cls="main"):
        for ch in node:
            if ch.tag == "default":
                name = ch.get("class")
                if name is None:  # top-level anonymous default = main
                    self.load(ch, cls)
                    continue
                self.classes[name] = {}
                self.parent[name] = cls
                self.load(ch, name)
            else:
                self.classes[cls].setdefault(ch.tag, {}).update(ch.attrib)

    def resolve(self, tag, cls, attrib):
        chain = []
        c = cls or "main"
        while c is not None:
            chain.append(c)
            c = self.parent.get(c)
        out = {}
        for c in reversed(chain):
            out.update(self.classes.get(c, {}).get(tag, {}))
        out.update(attrib)
        return out


def _expand_includes(root, base_dir):
    for parent in list(root.iter()):
        for i, ch in enumerate(list(parent)):
            if ch.tag == "include":
                sub = ET.parse(os.path.join(base_dir, ch.get("file"))).getroot()
                _expand_includes(sub, base_dir)
                idx = list(parent).index(ch)
                parent.remove(ch)
                for k, el in enumerate(list(sub)):
                    parent.insert(idx + k, el)


def _mjcf_orientation(attr, angle_scale):
    """Orientation of an MJCF frame (body, geom, site) as a quaternion xyzw.  MuJoCo accepts exactly one of `quat` (wxyz), `axisangle`
    ("x y z a", the angle in the compiler's unit), `euler` (compiler eulerseq, default "xyz": INTRINSIC rotations about x, then the new y, then
    the new z -- R = Rx Ry Rz, not URDF's fixed-axis rpy), `xyaxes` (the frame's x axis and a vector in its xy plane), `zaxis` (minimal rotation
    taking z there).  Round 5: `axisangle` was silently dropped until then -- the Shadow Hand's thumb base (robot.xml:126, 0.785 rad about y) sat
    unrotated -- and `euler` went through rpy_to_quat, which agrees with MuJoCo only while at most one angle is non-zero."""
    get = attr.get
    if get("quat"):
        w, x, y, z = _floats(get("quat"), 4)
        q = np.array([x, y, z, w])
        return q / np.linalg.norm(q)
    if get("axisangle"):
        x, y, z, a = _floats(get("axisangle"), 4)
        ax = np.array([x, y, z]) / np.linalg.norm([x, y, z])
        a *= angle_scale
        return np.concatenate([ax * math.sin(0.5 * a), [math.cos(0.5 * a)]])
    if get("euler"):
        e = _floats(get("euler"), 3) * angle_scale
        qs = [np.array([math.sin(0.5 * e[0]), 0.0, 0.0, math.cos(0.5 * e[0])]), np.array([0.0, math.sin(0.5 * e[1]), 0.0, math.cos(0.5 * e[1])]),
              np.array([0.0, 0.0, math.sin(0.5 * e[2]), math.cos(0.5 * e[2])])]
        return quat_mul(quat_mul(qs[0], qs[1]), qs[2])
    if get("xyaxes"):
        v = _floats(get("xyaxes"), 6)
        x = v[:3] / np.linalg.norm(v[:3])
        y = v[3:] - x * (x @ v[3:])
        y /= np.linalg.norm(y)
        return mat_to_quat(np.stack([x, y, np.cross(x, y)], axis=1))
    if get("zaxis"):
        return quat_from_z_to(_floats(get("zaxis"), 3))
    return np.array([0.0, 0.0, 0.0, 1.0])


def _mjcf_geom(attr, angle_scale):
    tname = attr.get("type", "sphere")
    gtype = {"sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE, "box": GEOM_BOX, "cylinder": GEOM_CYLINDER}.get(tname)
    if gtype is None:
        return None  # plane / mesh / etc: not a primitive we collide
    size = _floats(attr.get("size", "0"))
    pos = _floats(attr.get("pos", "0 0 0"), 3)
    quat = _mjcf_orientation(attr, angle_scale)
    s = np.zeros(3)
    if "fromto" in attr:
        ft = _floats(attr["fromto"], 6)
        a, b = ft[:3], ft[3:]
        pos = 0.5 * (a + b)
        quat = quat_from_z_to(b - a)
        s[0] = size[0]
        s[1] = 0.5 * np.linalg.norm(b - a)
    else:
        s[:len(size)] = size[:3]
    fr = _floats(attr.get("friction", "1 0.005 0.0001"))[0]
    g = _Geom(attr.get("name", ""), gtype, pos, quat, s, friction=fr, density=float(attr.get("density", 1000.0)))
    if "mass" in attr:
        g.mass = float(attr["mass"])
    g.contype = int(attr.get("contype", 1))
    g.conaffinity = int(attr.get("conaffinity", 1))
    return g


def mjcf_self_collision_filter(path):
    """The collision filters an MJCF asset itself carries, the ones `create_actor(..., collision_filter=-1, ...)` asks for ("use asset
    collision filters set in mjcf loader", reference shadow_hand.py:357-358): two geoms may touch when
    `contype_a & conaffinity_b or contype_b & conaffinity_a`.  -> dict(collision_geoms = number of primitive geoms that collide with
    anything, accepting = the geoms with a non-zero conaffinity [(body, geom, largest half extent)], pairs = the (geom_a, geom_b)
    names on different bodies that may touch).  For the Shadow Hand every collision geom is contype 1 / conaffinity 0 (shared.xml:21,
    class robot0:DC_Hand): no hand shape accepts another hand shape.  What is left are two 1 mm placeholder boxes at the thumb's joint
    origins (robot.xml:129,138, default class) -- the manipulated objects (contype = conaffinity = 1) accept all hand shapes."""
    bodies, _ = parse_mjcf(path)
    geoms = [(b.name, g) for b in bodies for g in b.geoms if (g.contype | g.conaffinity)]
    pairs = []
    for i, (ba, ga) in enumerate(geoms):
        for bb, gb in geoms[i + 1:]:
            if ba != bb and ((ga.contype & gb.conaffinity) or (gb.contype & ga.conaffinity)):
                pairs.append((ga.name or ba, gb.name or bb))
    accepting = [(b, g.name or b, float(np.max(g.size))) for b, g in geoms if g.conaffinity]
    return dict(collision_geoms=len(geoms), accepting=accepting, pairs=pairs)


def parse_mjcf(path):
    """Parse an MJCF file into the intermediate body list + actuators."""
    root = ET.parse(path).getroot()
    _expand_includes(root, os.path.dirname(path))
    comp = root.find("compiler")
    angle = (comp.get("angle") if comp is not None and comp.get("angle") else "degree")
    ascale = math.pi / 180.0 if angle == "degree" else 1.0
    defaults = _MjcfDefaults()
    for d in root.findall("default"):
        defaults.load(d, "main")

    bodies: list[_Body] = []

    def visit(node, parent_idx, childclass):
        cc = node.get("childclass", childclass)
        pos = _floats(node.get("pos", "0 0 0"), 3)
        quat = _mjcf_orientation(node.attrib, ascale)
        b = _Body(node.get("name", f"body{len(bodies)}"), parent_idx, pos, quat)
        idx = len(bodies)
        bodies.append(b)
        for ch in node:
            if ch.tag == "freejoint":
                b.free = True
            elif ch.tag == "joint":
                a = defaults.resolve("joint", ch.get("class", cc), ch.attrib)
                jt = a.get("type", "hinge")
                if jt == "free":
                    b.free = True
                    continue
                jtype = JOINT_HINGE if jt == "hinge" else JOINT_SLIDE
                axis = _floats(a.get("axis", "0 0 1"), 3)
                axis = axis / np.linalg.norm(axis)
                rng = _floats(a.get("range", "0 0"), 2)
                sc = ascale if jtype == JOINT_HINGE else 1.0
                limited = a.get("limited", "false") == "true"
                j = _Joint(a.get("name", f"joint{idx}"), jtype, axis, _floats(a.get("pos", "0 0 0"), 3),
                           lower=rng[0] * sc, upper=rng[1] * sc, limited=limited,
                           armature=float(a.get("armature", 0)), damping=float(a.get("damping", 0)),
                           stiffness=float(a.get("stiffness", 0)), springref=float(a.get("springref", 0)) * sc,
                           frictionloss=float(a.get("frictionloss", 0)))
                b.joints.append(j)
            elif ch.tag == "geom":
                a = defaults.resolve("geom", ch.get("class", cc), ch.attrib)
                g = _mjcf_geom(a, ascale)
                if g is not None:
                    b.geoms.append(g)
            elif ch.tag == "inertial":
                ipos = _floats(ch.get("pos", "0 0 0"), 3)
                iq = np.array([0.0, 0.0, 0.0, 1.0])
                if ch.get("quat"):
                    w, x, y, z = _floats(ch.get("quat"), 4)
                    iq = np.array([x, y, z, w])
                    iq /= np.linalg.norm(iq)
                b.inertial = (float(ch.get("mass")), ipos, iq, _floats(ch.get("diaginertia", "0 0 0"), 3))
            elif ch.tag == "body":
                visit(ch, idx, cc)

    wb = root.find("worldbody")
    tops = [n for n in wb if n.tag == "body"]
    if len(tops) != 1:
        raise ValueError("expected exactly one top-level body")
    visit(tops[0], -1, None)

    acts = []
    an = root.find("actuator")
    if an is not None:
        for m in an:
            a = defaults.resolve(m.tag, m.get("class"), m.attrib)
            acts.append(dict(name=a.get("name", a.get("joint", "")), joint=a.get("joint"), kind=m.tag,
                             gear=_floats(a.get("gear", "1"))[0], kp=float(a.get("kp", 0)),
                             ctrlrange=_floats(a.get("ctrlrange", "0 0"), 2).tolist(),
                             forcerange=_floats(a.get("forcerange", "0 0"), 2).tolist()))
    return bodies, acts


# ----------------------------------------------------------------------------
# URDF
# ----------------------------------------------------------------------------
def _resolve_mesh(filename, urdf_path, mesh_root=None):
    """A URDF mesh file name (relative to the asset root the task passes to gym.load_asset, or to the URDF's own directory; package://
    URIs lose their prefix) -> an existing path, or None."""
    name = filename.split("package://")[-1]
    here = os.path.dirname(os.path.abspath(urdf_path))
    roots = ([mesh_root] if mesh_root else []) + [here]
    for _ in range(5):          # a package:// name starts at the package's PARENT directory, which may sit several levels above the URDF (trifingerpro.urdf: three)
        roots.append(os.path.dirname(roots[-1]))
    for r in roots:
        cand = os.path.join(r, name)
        if os.path.exists(cand):
            return cand
    return None


def parse_urdf(path, default_density=1000.0, replace_cylinder_with_capsule=False, mesh_spheres=False, mesh_root=None, mesh_options=None,
               mesh_link_filter=None):
    """mesh_spheres: collision <mesh> shapes become sphere geoms inscribed in the mesh's convex hull (assets/mesh.py hull_spheres; PhysX
    collides a mesh shape as its hull); without it they are skipped -- fine for robots that never touch anything with those links."""
    root = ET.parse(path).getroot()
    links = {}
    for ln in root.findall("link"):
        name = ln.get("name")
        geoms = []
        mesh_hulls = []                  # (vertices in the link frame, density) of the link's collision meshes
        for col in ln.findall("collision"):
            org = col.find("origin")
            pos = _floats(org.get("xyz", "0 0 0"), 3) if org is not None else np.zeros(3)
            quat = rpy_to_quat(*_floats(org.get("rpy", "0 0 0"), 3)) if org is not None else np.array([0, 0, 0, 1.0])
            geo = col.find("geometry")
            if geo is None or len(geo) == 0:
                continue
            g = geo[0]
            if g.tag == "box":
                geoms.append(_Geom(name, GEOM_BOX, pos, quat, 0.5 * _floats(g.get("size"), 3), density=default_density))
            elif g.tag == "sphere":
                geoms.append(_Geom(name, GEOM_SPHERE, pos, quat, np.array([float(g.get("radius")), 0, 0]), density=default_density))
            elif g.tag == "cylinder":
                r, L = float(g.get("radius")), float(g.get("length"))
                if replace_cylinder_with_capsule:
                    geoms.append(_Geom(name, GEOM_CAPSULE, pos, quat, np.array([r, max(0.5 * L - r, 0.0), 0]), density=default_density))
                else:
                    geoms.append(_Geom(name, GEOM_CYLINDER, pos, quat, np.array([r, 0.5 * L, 0]), density=default_density))
            elif g.tag == "mesh" and mesh_spheres and (mesh_link_filter is None or mesh_link_filter(name)):
                from .mesh import hull_spheres, load_mesh
                mpath = _resolve_mesh(g.get("filename"), path, mesh_root)
                if mpath is None:
                    raise FileNotFoundError(f"{path}: collision mesh {g.get('filename')} of link {name} not found")
                V, _ = load_mesh(mpath)
                V = V * (_floats(g.get("scale"), 3) if g.get("scale") else 1.0)
                C, rad = hull_spheres(V, **(mesh_options or {}))
                R = quat_to_mat(quat)
                mesh_hulls.append((V @ R.T + pos, default_density))
                for c in C:      # sphere geoms in the link frame; mass properties never come from them (density 0)
                    geoms.append(_Geom(name, GEOM_SPHERE, pos + R @ c, np.array([0, 0, 0, 1.0]), np.array([rad, 0, 0]), density=0.0))
        inertial = None
        if mesh_hulls and ln.find("inertial") is None and not any(gm.density > 0 for gm in geoms):
            # a link described by collision meshes alone (franka_panda_gripper.urdf): mass properties of the meshes' convex hulls at the
            # asset's density, as the simulator derives them
            from .mesh import hull_mass_properties
            parts = [hull_mass_properties(Vh, dens) for Vh, dens in mesh_hulls]
            mt = sum(pm for pm, _, _ in parts)
            if mt > 0:
                ct = sum(pm * pc for pm, pc, _ in parts) / mt
                It = sum(pI + pm * (np.dot(pc - ct, pc - ct) * np.eye(3) - np.outer(pc - ct, pc - ct)) for pm, pc, pI in parts)
                inertial = (mt, ct, np.array([0, 0, 0, 1.0]), It)
        iner = ln.find("inertial")
        if iner is not None and iner.find("mass") is not None:
            m = float(iner.find("mass").get("value"))
            org = iner.find("origin")
            ipos = _floats(org.get("xyz", "0 0 0"), 3) if org is not None else np.zeros(3)
            iq = rpy_to_quat(*_floats(org.get("rpy", "0 0 0"), 3)) if org is not None else np.array([0, 0, 0, 1.0])
            it = iner.find("inertia")
            full = None
            if it is not None:
                full = np.array([[float(it.get("ixx", 0)), float(it.get("ixy", 0)), float(it.get("ixz", 0))],
                                 [float(it.get("ixy", 0)), float(it.get("iyy", 0)), float(it.get("iyz", 0))],
                                 [float(it.get("ixz", 0)), float(it.get("iyz", 0)), float(it.get("izz", 0))]])
            inertial = (m, ipos, iq, full)
        links[name] = dict(geoms=geoms, inertial=inertial)
    joints = []
    for jn in root.findall("joint"):
        org = jn.find("origin")
        pos = _floats(org.get("xyz", "0 0 0"), 3) if org is not None else np.zeros(3)
        quat = rpy_to_quat(*_floats(org.get("rpy", "0 0 0"), 3)) if org is not None else np.array([0, 0, 0, 1.0])
        ax = jn.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else np.array([1.0, 0, 0])
        lim = jn.find("limit")
        dyn = jn.find("dynamics")
        joints.append(dict(name=jn.get("name"), type=jn.get("type"), parent=jn.find("parent").get("link"),
                           child=jn.find("child").get("link"), pos=pos, quat=quat, axis=axis / max(np.linalg.norm(axis), 1e-12),
                           lower=float(lim.get("lower", 0)) if lim is not None else 0.0,
                           upper=float(lim.get("upper", 0)) if lim is not None else 0.0,
                           has_limits=lim is not None and lim.get("lower") is not None,
                           effort=float(lim.get("effort", 1e30)) if lim is not None else 1e30,
                           velocity=float(lim.get("velocity", 1e30)) if lim is not None else 1e30,
                           damping=float(dyn.get("damping", 0)) if dyn is not None else 0.0))
    children = {j["child"] for j in joints}
    roots = [n for n in links if n not in children]
    if len(roots) != 1:
        raise ValueError(f"URDF must have a single root link, got {roots}")
    bodies: list[_Body] = []

    def visit(link_name, parent_idx, joint):
        L = links[link_name]
        if joint is None:
            b = _Body(link_name, -1, np.zeros(3), np.array([0, 0, 0, 1.0]))
        else:
            b = _Body(link_name, parent_idx, joint["pos"], joint["quat"])
            t = joint["type"]
            if t in ("revolute", "continuous", "prismatic"):
                limited = (t != "continuous") and joint["has_limits"]
                b.joints.append(_Joint(joint["name"], JOINT_SLIDE if t == "prismatic" else JOINT_HINGE, joint["axis"],
                                       np.zeros(3), lower=joint["lower"], upper=joint["upper"], limited=limited,
                                       damping=joint["damping"], effort=joint["effort"], velocity=joint["velocity"]))
            elif t != "fixed":
                raise ValueError(f"unsupported URDF joint type {t}")
        b.geoms = L["geoms"]
        if L["inertial"] is not None:
            m, ipos, iq, full = L["inertial"]
            b.inertial = (m, ipos, iq, full)
        idx = len(bodies)
        bodies.append(b)
        for j in joints:
            if j["parent"] == link_name:
                visit(j["child"], idx, j)

    visit(roots[0], -1, None)
    return bodies, []


# ----------------------------------------------------------------------------
# intermediate tree -> ModelSpec
# ----------------------------------------------------------------------------
def _body_mass_props(b: _Body, density_override=None):
    """-> mass, com (3), inertia about com 3x3, all in body frame."""
    if b.inertial is not None and b.inertial[3] is not None and np.any(np.asarray(b.inertial[3]) != 0):
        m, ipos, iq, I = b.inertial
        R = quat_to_mat(iq)
        I = np.asarray(I, float)
        Ifull = R @ (np.diag(I) if I.ndim == 1 else I) @ R.T
        return m, np.asarray(ipos, float), Ifull
    if not b.geoms:
        if b.inertial is not None:
            return b.inertial[0], np.asarray(b.inertial[1], float), np.eye(3) * 1e-9
        return 0.0, np.zeros(3), np.zeros((3, 3))
    ms, cs, Is = [], [], []
    for g in b.geoms:
        dens = density_override if density_override is not None else g.density
        m, d = geom_mass_inertia(g.gtype, g.size, dens)
        if g.mass is not None:
            d = d * (g.mass / m)
            m = g.mass
        R = quat_to_mat(g.quat)
        ms.append(m)
        cs.append(g.pos)
        Is.append(R @ np.diag(d) @ R.T)
    M = sum(ms)
    if M <= 0.0:
        # only mass-less geometry (the spheres a collision MESH was sampled by) and an explicit mass without an inertia tensor: a solid
        # ball of that mass reaching the farthest sphere (franka_panda_gripper.urdf's finger links)
        if b.inertial is None:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        m_exp, ipos = b.inertial[0], np.asarray(b.inertial[1], float)
        r = max([float(np.linalg.norm(np.asarray(c, float) - ipos)) + float(g.size[0]) for g, c in zip(b.geoms, cs)] + [0.01])
        return m_exp, ipos, 0.4 * m_exp * r * r * np.eye(3)
    com = sum(m * c for m, c in zip(ms, cs)) / M
    I = np.zeros((3, 3))
    for m, c, Ig in zip(ms, cs, Is):
        r = c - com
        I += Ig + m * (np.dot(r, r) * np.eye(3) - np.outer(r, r))
    if b.inertial is not None:  # explicit mass, geometry-derived shape (URDF without <inertia>)
        m_exp, ipos = b.inertial[0], np.asarray(b.inertial[1], float)
        I = I * (m_exp / M)
        M, com = m_exp, ipos
    return M, com, I


def _contact_spheres(geom_body, geom_type, geom_pos, geom_quat, geom_size, geom_friction):
    sb, sp, sr, sf, sg = [], [], [], [], []
    for gi in range(len(geom_body)):
        t, p, q, s = geom_type[gi], geom_pos[gi], geom_quat[gi], geom_size[gi]
        R = quat_to_mat(q)
        pts = []
        if t == GEOM_SPHERE:
            pts = [(p, s[0])]
        elif t == GEOM_CAPSULE:
            pts = [(p + R @ np.array([0, 0, s[1]]), s[0]), (p - R @ np.array([0, 0, s[1]]), s[0])]
        elif t == GEOM_BOX:
            for sx in (-1, 1):
                for sy in (-1, 1):
                    for sz in (-1, 1):
                        pts.append((p + R @ (np.array([sx, sy, sz]) * s), 0.0))
        elif t == GEOM_CYLINDER:  # rim approximated by 4 points per cap
            for sz in (-1, 1):
                for a in range(4):
                    pts.append((p + R @ np.array([s[0] * math.cos(a * math.pi / 2), s[0] * math.sin(a * math.pi / 2), sz * s[1]]), 0.0))
        for c, r in pts:
            sb.append(geom_body[gi]); sp.append(c); sr.append(r); sf.append(geom_friction[gi]); sg.append(gi)
    return (np.array(sb, np.int32), np.array(sp, float).reshape(-1, 3), np.array(sr, float), np.array(sf, float),
            np.array(sg, np.int32))


# weld of a locked joint (bounds coincide) in the mass matrix, see build_model
LOCKED_ARMATURE, LOCKED_STIFFNESS, LOCKED_DAMPING = 100.0, 1000.0, 600.0


def build_model(name, bodies, acts, fix_base_link=False, density=None, dedupe_spheres=True,
                collide_body_filter=None):
    """Flatten the intermediate tree.  Welded (jointless, non-root) bodies are merged into parents."""
    n = len(bodies)
    root = bodies[0]
    fixed_base = bool(fix_base_link or not (root.free or True) )
    # Isaac Gym: a root body without fix_base_link is a free-floating base regardless of <freejoint>
    fixed_base = bool(fix_base_link)
    props = [_body_mass_props(b, density) for b in bodies]
    # dynamic body assignment
    dyn_of = [-1] * n
    off_R = [np.eye(3)] * n
    off_p = [np.zeros(3)] * n
    dyn_bodies = []
    for i, b in enumerate(bodies):
        if i == 0 or b.joints:
            dyn_of[i] = len(dyn_bodies)
            dyn_bodies.append(i)
        else:  # weld into parent's dynamic body
            p = b.parent
            dyn_of[i] = dyn_of[p]
            Rb = quat_to_mat(b.quat)
            off_R[i] = off_R[p] @ Rb
            off_p[i] = off_p[p] + off_R[p] @ b.pos
    nb = len(dyn_bodies)
    parent = np.full(nb, -1, np.int32)
    bpos = np.zeros((nb, 3)); bquat = np.zeros((nb, 4)); bquat[:, 3] = 1
    mass = np.zeros(nb); com = np.zeros((nb, 3)); inertia = np.zeros((nb, 6))
    # accumulate mass props (weld-aware)
    acc = [[] for _ in range(nb)]
    for i, b in enumerate(bodies):
        m, c, I = props[i]
        if m > 0:
            acc[dyn_of[i]].append((m, off_p[i] + off_R[i] @ c, off_R[i] @ I @ off_R[i].T))
    for k, i in enumerate(dyn_bodies):
        b = bodies[i]
        if b.parent >= 0:
            p = b.parent
            parent[k] = dyn_of[p]
            bpos[k] = off_p[p] + off_R[p] @ b.pos
            bquat[k] = mat_to_quat(off_R[p] @ quat_to_mat(b.quat))
        else:
            bpos[k] = 0.0  # root pose comes from the actor start pose (reference ant.py:164), not the file
        M = sum(a[0] for a in acc[k]) if acc[k] else 0.0
        if M > 0:
            cm = sum(a[0] * a[1] for a in acc[k]) / M
            I = np.zeros((3, 3))
            for m, c, Ig in acc[k]:
                r = c - cm
                I += Ig + m * (np.dot(r, r) * np.eye(3) - np.outer(r, r))
            mass[k] = M; com[k] = cm
            inertia[k] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
    # dofs
    dn, db, dt, dax, dan, dlo, dup, dlim, darm, ddamp, dst, dref, deff, dvel = ([] for _ in range(14))
    for k, i in enumerate(dyn_bodies):
        for j in bodies[i].joints:
            lo, up = j.lower, j.upper
            dn.append(j.name); db.append(k); dt.append(j.jtype); dax.append(j.axis); dan.append(j.anchor)
            locked = bool(j.limited) and lo == up
            if locked:
                # A limited joint whose bounds coincide (MJCF range="0 0": the physics rotors of Ingenuity) is a weld that keeps its dof
                # slot.  As a limit row it is the worst case for the Gauss-Seidel sweeps -- two such joints on one light parent give a
                # constraint matrix with 0.975 off-diagonal correlation (6 sweeps remove a quarter of the error) and, on the GPU, a
                # growing oscillation.  It is welded in the mass matrix instead: a large armature on the joint coordinate (the child
                # cannot move relative to its parent, its inertia still loads the parent) plus a stiff, overdamped centering spring;
                # no limit row.  H stays well conditioned for the sparse factorisation (diagonal 1e2 against inertias of 1e-2).
                dlim.append(0)
                darm.append(j.armature + LOCKED_ARMATURE); ddamp.append(j.damping + LOCKED_DAMPING); dst.append(j.stiffness + LOCKED_STIFFNESS)
                dref.append(lo)
                dlo.append(lo); dup.append(up)
                deff.append(j.effort); dvel.append(j.velocity)
                continue
            dlo.append(lo); dup.append(up); dlim.append(1 if j.limited else 0)
            darm.append(j.armature); ddamp.append(j.damping); dst.append(j.stiffness); dref.append(j.springref)
            deff.append(j.effort); dvel.append(j.velocity)
    # geoms
    gn, gb, gt, gp, gq, gs, gf = ([] for _ in range(7))
    for i, b in enumerate(bodies):
        if collide_body_filter is not None and not collide_body_filter(b.name):
            continue
        for g in b.geoms:
            gn.append(g.name or b.name); gb.append(dyn_of[i]); gt.append(g.gtype)
            gp.append(off_p[i] + off_R[i] @ g.pos)
            gq.append(mat_to_quat(off_R[i] @ quat_to_mat(g.quat)))
            gs.append(g.size); gf.append(g.friction)
    geom_body = np.array(gb, np.int32); geom_type = np.array(gt, np.int32)
    geom_pos = np.array(gp, float).reshape(-1, 3); geom_quat = np.array(gq, float).reshape(-1, 4)
    geom_size = np.array(gs, float).reshape(-1, 3); geom_friction = np.array(gf, float)
    sb, sp, sr, sf, sg = _contact_spheres(geom_body, geom_type, geom_pos, geom_quat, geom_size, geom_friction)
    dof_body = np.array(db, np.int32)
    dof_anchor = np.array(dan, float).reshape(-1, 3)
    if dedupe_spheres and len(sb):
        sb, sp, sr, sf, sg = _dedupe_spheres(sb, sp, sr, sf, sg, parent, bpos, bquat, dof_body, dof_anchor,
                                             np.array(dt, np.int32))
    dof_index = {nme: k for k, nme in enumerate(dn)}
    an, ad, ag = [], [], []
    for a in acts:
        if a["joint"] in dof_index:
            an.append(a["name"]); ad.append(dof_index[a["joint"]]); ag.append(a["gear"])
    return ModelSpec(
        name=name, fixed_base=fixed_base, body_names=[bodies[i].name for i in dyn_bodies], parent=parent, bpos=bpos,
        bquat=bquat, mass=mass, com=com, inertia=inertia, dof_names=dn, dof_body=dof_body,
        dof_type=np.array(dt, np.int32), dof_axis=np.array(dax, float).reshape(-1, 3), dof_anchor=dof_anchor,
        dof_lower=np.array(dlo, float), dof_upper=np.array(dup, float), dof_limited=np.array(dlim, np.int32),
        dof_armature=np.array(darm, float), dof_damping=np.array(ddamp, float), dof_stiffness=np.array(dst, float),
        dof_springref=np.array(dref, float), dof_effort=np.array(deff, float), dof_velocity=np.array(dvel, float),
        geom_names=gn, geom_body=geom_body, geom_type=geom_type, geom_pos=geom_pos, geom_quat=geom_quat,
        geom_size=geom_size, geom_friction=geom_friction, sph_body=sb, sph_pos=sp, sph_rad=sr, sph_friction=sf,
        sph_geom=sg, act_names=an, act_dof=np.array(ad, np.int32), act_gear=np.array(ag, float),
        api_body_names=[b.name for b in bodies], api_body_dyn=np.array(dyn_of, np.int32),
        api_body_pos=np.array(off_p, float).reshape(-1, 3),
        api_body_quat=np.array([mat_to_quat(R) for R in off_R], float).reshape(-1, 4))


def _dedupe_spheres(sb, sp, sr, sf, sg, parent, bpos, bquat, dof_body, dof_anchor, dof_type):
    """Drop a contact sphere that coincides (same world point at every configuration, same radius) with one on
    the parent body: that is the case when it sits exactly on the hinge anchor(s) connecting the two bodies, where
    both bodies have identical point velocity, so the second contact row would be a duplicate constraint."""
    keep = np.ones(len(sb), bool)
    # a sphere wholly inside another sphere of the same body can never be the first to touch anything
    for i in range(len(sb)):
        for k in range(len(sb)):
            if i != k and keep[k] and sb[i] == sb[k] and np.linalg.norm(sp[i] - sp[k]) + sr[i] <= sr[k] + 1e-12 \
                    and (sr[i] < sr[k] or k < i):
                keep[i] = False
                break
    for i in range(len(sb)):
        if not keep[i]:
            continue
        b = sb[i]
        p = parent[b]
        if p < 0:
            continue
        jd = [d for d in range(len(dof_body)) if dof_body[d] == b]
        if not jd or any(dof_type[d] != JOINT_HINGE for d in jd):
            continue
        if not all(np.linalg.norm(dof_anchor[d] - sp[i]) < 1e-9 for d in jd):
            continue
        # position of this point in the parent frame (independent of q because it lies on every hinge axis)
        pp = bpos[b] + quat_to_mat(bquat[b]) @ sp[i]
        for k in range(len(sb)):
            if keep[k] and sb[k] == p and abs(sr[k] - sr[i]) < 1e-9 and np.linalg.norm(sp[k] - pp) < 1e-9:
                keep[i] = False
                break
    return sb[keep], sp[keep], sr[keep], sf[keep], sg[keep]


def load_asset(path, name=None, fix_base_link=False, density=None, replace_cylinder_with_capsule=False, **kw):
    """Front door mirroring ``gym.load_asset(sim, root, file, AssetOptions)`` for the options the five tasks use."""
    name = name or os.path.splitext(os.path.basename(path))[0]
    if path.endswith(".urdf"):
        bodies, acts = parse_urdf(path, default_density=density if density is not None else 1000.0,
                                  replace_cylinder_with_capsule=replace_cylinder_with_capsule, mesh_spheres=kw.pop("mesh_spheres", False),
                                  mesh_root=kw.pop("mesh_root", None), mesh_options=kw.pop("mesh_options", None),
                                  mesh_link_filter=kw.pop("mesh_link_filter", None))
        return build_model(name, bodies, acts, fix_base_link=fix_base_link, density=density, **kw)
    bodies, acts = parse_mjcf(path)
    return build_model(name, bodies, acts, fix_base_link=fix_base_link, density=density, **kw)


# ----------------------------------------------------------------------------
# self-collision (actors created with collision filter 0: reference humanoid.py:194)
# ----------------------------------------------------------------------------
def collision_capsules(spec: ModelSpec):
    """Sphere / capsule collision geoms as segments in their (dynamic) body frame:
    -> (body [G], p0 [G,3], p1 [G,3], radius [G], friction [G], geom index [G]).  A sphere is a zero-length segment."""
    gb, p0, p1, rad, mu, gi = [], [], [], [], [], []
    for g in range(len(spec.geom_body)):
        t = int(spec.geom_type[g])
        if t not in (GEOM_SPHERE, GEOM_CAPSULE):
            continue
        c = np.asarray(spec.geom_pos[g], float)
        half = np.zeros(3)
        if t == GEOM_CAPSULE:
            half = quat_to_mat(np.asarray(spec.geom_quat[g], float)) @ np.array([0.0, 0.0, float(spec.geom_size[g][1])])
        gb.append(int(spec.geom_body[g])); p0.append(c + half); p1.append(c - half)
        rad.append(float(spec.geom_size[g][0])); mu.append(float(spec.geom_friction[g])); gi.append(g)
    return (np.array(gb, np.int32), np.array(p0, float).reshape(-1, 3), np.array(p1, float).reshape(-1, 3),
            np.array(rad, float), np.array(mu, float), np.array(gi, np.int32))


def _axis_angle_mats(a, th):
    """Rodrigues, batched: a [N,3] unit axes, th [N] -> [N,3,3]."""
    c, s = np.cos(th)[:, None, None], np.sin(th)[:, None, None]
    K = np.zeros((len(th), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -a[:, 2], a[:, 1], a[:, 2], -a[:, 0], -a[:, 1], a[:, 0]
    return c * np.eye(3)[None] + s * K + (1 - c) * a[:, :, None] * a[:, None, :]


def body_frames(spec: ModelSpec, q):
    """Batched forward kinematics with the root at the origin: q [N, nd] -> (R [N, nb, 3, 3], r [N, nb, 3]).
    Same convention as the engine's tree pass (joints rotate / slide the child about anchors given in the child frame)."""
    q = np.atleast_2d(np.asarray(q, float))
    n = len(q)
    R = np.zeros((n, spec.nb, 3, 3)); r = np.zeros((n, spec.nb, 3))
    for b in range(spec.nb):
        p = int(spec.parent[b])
        if p < 0:
            Rb = np.tile(np.eye(3), (n, 1, 1)); rb = np.zeros((n, 3))
        else:
            Rb = R[:, p] @ quat_to_mat(np.asarray(spec.bquat[b], float))
            rb = r[:, p] + R[:, p] @ np.asarray(spec.bpos[b], float)
        for d in range(spec.nd):
            if int(spec.dof_body[d]) != b:
                continue
            a = Rb @ np.asarray(spec.dof_axis[d], float)
            pt = rb + Rb @ np.asarray(spec.dof_anchor[d], float)
            if int(spec.dof_type[d]) == JOINT_HINGE:
                Q = _axis_angle_mats(a, q[:, d])
                Rb = Q @ Rb
                rb = pt + np.einsum("nij,nj->ni", Q, rb - pt)
            else:
                rb = rb + a * q[:, d:d + 1]
        R[:, b], r[:, b] = Rb, rb
    return R, r


def segment_distance(a0, a1, b0, b1):
    """Closest points of segments [a0,a1], [b0,b1], batched [N,3] -> (ca, cb).  Clamped closest-point construction; a zero-length
    segment degenerates to a point.  The engine (csrc/core/engine.hpp `seg_seg_closest`) and the oracle use the same
    sequence of operations."""
    d1, d2, rr = a1 - a0, b1 - b0, a0 - b0
    A = (d1 * d1).sum(-1); E = (d2 * d2).sum(-1); F = (d2 * rr).sum(-1); Cc = (d1 * rr).sum(-1); Bb = (d1 * d2).sum(-1)
    eps = 1e-12
    den = A * E - Bb * Bb
    s = np.where(den > eps, np.clip((Bb * F - Cc * E) / np.where(den > eps, den, 1.0), 0.0, 1.0), 0.0)
    s = np.where(A > eps, s, 0.0)
    t = np.where(E > eps, (Bb * s + F) / np.where(E > eps, E, 1.0), 0.0)
    t_c = np.clip(t, 0.0, 1.0)
    s2 = np.where(A > eps, np.clip((Bb * t_c - Cc) / np.where(A > eps, A, 1.0), 0.0, 1.0), 0.0)
    s = np.where((t != t_c) | ~(E > eps), s2, s)
    return a0 + d1 * s[:, None], b0 + d2 * t_c[:, None]


def self_collision_pairs(spec: ModelSpec, n_samples=40000, margin=0.03, limit_slack=0.15, seed=0):
    """Static broad phase of an actor that collides with itself: the body pairs (a < b) that are not joined by a joint and whose
    collision capsules come within `margin` of each other somewhere inside the joint limits (widened by `limit_slack` rad, since
    limits are soft constraints), found by sampling configurations.  Pairs that can never touch -- most of them -- cost nothing
    at run time.  -> list of (body_a, body_b, [(capsule_i, capsule_j), ...])."""
    gb, p0, p1, rad, _, _ = collision_capsules(spec)
    if len(gb) < 2:
        return []
    rng = np.random.default_rng(seed)
    lo = np.minimum(spec.dof_lower, spec.dof_upper) - limit_slack
    up = np.maximum(spec.dof_lower, spec.dof_upper) + limit_slack
    lim = np.asarray(spec.dof_limited, bool)
    lo = np.where(lim, lo, -np.pi); up = np.where(lim, up, np.pi)
    q = rng.uniform(lo, up, (n_samples, spec.nd))
    R, r = body_frames(spec, q)
    w0 = r[:, gb] + np.einsum("ngij,gj->ngi", R[:, gb], p0)
    w1 = r[:, gb] + np.einsum("ngij,gj->ngi", R[:, gb], p1)
    out = {}
    for i in range(len(gb)):
        for j in range(i + 1, len(gb)):
            a, b = int(gb[i]), int(gb[j])
            if a == b or int(spec.parent[a]) == b or int(spec.parent[b]) == a:
                continue
            ca, cb = segment_distance(w0[:, i], w1[:, i], w0[:, j], w1[:, j])
            dmin = (np.linalg.norm(ca - cb, axis=1) - rad[i] - rad[j]).min()
            if dmin < margin:
                key = (min(a, b), max(a, b))
                out.setdefault(key, []).append((i, j) if a < b else (j, i))
    return [(a, b, gp) for (a, b), gp in sorted(out.items())]


def limb_paths(spec: ModelSpec):
    """Partition of the bodies into ancestor->descendant paths ("limbs"): a body continues its parent's limb when it is the only child or
    the child with the strictly largest subtree, and starts a new limb otherwise.  Humanoid: {torso, lower_waist, pelvis}, two legs,
    two arms.  -> (limb id per body [nb], list of body lists)."""
    nb = spec.nb
    children = [[] for _ in range(nb)]
    for b in range(1, nb):
        children[int(spec.parent[b])].append(b)
    size = [1] * nb
    for b in range(nb - 1, 0, -1):
        size[int(spec.parent[b])] += size[b]
    limb = [-1] * nb
    limbs = []
    for b in range(nb):
        p = int(spec.parent[b])
        cont = False
        if p >= 0:
            sib = children[p]
            cont = len(sib) == 1 or all(size[b] > size[c] for c in sib if c != b)
        if cont:
            limb[b] = limb[p]
            limbs[limb[b]].append(b)
        else:
            limb[b] = len(limbs)
            limbs.append([b])
    return limb, limbs


def _has_selfcol(spec: ModelSpec) -> bool:
    try:
        from ..registry import load_selfcol
        return bool(load_selfcol(spec.name))
    except Exception:      # noqa: BLE001 -- models outside the registry (tests build specs by hand)
        return False


# wavefronts per env of the multi-wave sub-step (default 4: one per SIMD of a CU)
WAVE_ROLES = {"shadow_hand": 4}
# role of every limb where the greedy deal below is not the measured optimum: (role per limb, trunk role)
WAVE_ROLE_OVERRIDE = {}     # (shadow_hand ([-1, 2, 1, 2, 0, 1], 3) -- thumb + middle finger on one wave, the trunk rows alone -- measured: slower, profiles/r3k_*)


def wave_roles(spec: ModelSpec, nrole=None, pair_role=None):
    """Limbs dealt to the `nrole` wavefronts of the multi-wave sub-step (csrc/core/engine_mw.hpp, engine_mwc.hpp): the limb of the root
    body is the shared trunk (role -1, every wave recomputes it), the others go heaviest first to the least loaded wave, where a limb
    weighs the summed chain lengths of its dofs (what its rows cost); the trunk's own constraint rows go to the wave with the lightest
    limbs.  pair_role (default: the model has self-collision tables): the LAST wave owns no limb -- it runs the self-collision narrow
    phase and sweeps the self-contact rows (engine_mwc.hpp).  -> (limb per body, body lists, role per limb, trunk role, nrole)."""
    limb, limbs = limb_paths(spec)
    if nrole is None:
        nrole = WAVE_ROLES.get(getattr(spec, "name", ""), 4)
    if pair_role is None:
        pair_role = _has_selfcol(spec)
    off = 0 if spec.fixed_base else 6
    depth = [0] * spec.nb                     # dofs on the path root .. body (incl. the floating base)
    cnt = [0] * spec.nb
    for d in range(spec.nd):
        cnt[int(spec.dof_body[d])] += 1
    for b in range(spec.nb):
        p = int(spec.parent[b])
        depth[b] = (depth[p] if p >= 0 else off) + cnt[b]
    weight = [sum(sum(depth[b] - cnt[b] + k + 1 for k in range(cnt[b])) for b in limbs[l]) for l in range(len(limbs))]
    nlr = nrole - 1 if pair_role else nrole
    load = [0.0] * nlr
    role_of_limb = [-1] * len(limbs)
    order = sorted(range(1, len(limbs)), key=lambda l: (-weight[l], l))
    for l in order:
        r = min(range(nlr), key=lambda k: (load[k], k))
        role_of_limb[l] = r
        load[r] += weight[l] + 0.01 * len(limbs[l])
    trunk_role = min(range(nlr), key=lambda k: (load[k], k))
    if getattr(spec, "name", "") in WAVE_ROLE_OVERRIDE:
        role_of_limb, trunk_role = WAVE_ROLE_OVERRIDE[spec.name]
        assert len(role_of_limb) == len(limbs) and role_of_limb[0] == -1
        role_of_limb = list(role_of_limb)
    return limb, limbs, role_of_limb, trunk_role, nrole


def wave_contact_caps(spec: ModelSpec):
    """Ground contacts each wave of the compact-store limb-per-wave sub-step keeps per env (csrc/core/engine_mwc.hpp gives every role its
    own contact slots in LDS): 4 for a role whose limb has at least 4 dofs (a leg: one flat foot is 4 spheres), 3 for the role that also
    sweeps the trunk's rows (what the LDS of a CU still has room for at 32 envs), 1 otherwise, 0 for a role without bodies.  -> [nrole]"""
    limb, limbs, role_of_limb, trunk_role, nrole = wave_roles(spec)
    maxd = [0] * nrole                        # dofs of the role's longest limb
    for l, bodies in enumerate(limbs):
        r = role_of_limb[l]
        if r >= 0:
            maxd[r] = max(maxd[r], sum(1 for d in range(spec.nd) if int(spec.dof_body[d]) in bodies))
    owns = [any(role_of_limb[l] == r for l in range(len(limbs))) or r == trunk_role for r in range(nrole)]
    return [0 if not owns[r] else (4 if maxd[r] >= 4 else (3 if r == trunk_role else 1)) for r in range(nrole)]


def solver_blocks(spec: ModelSpec, self_collision: bool = False, wave_caps: bool = False):
    """What the block solver order needs to know about a model (oracle/physics.c OrModel.solver = 1, the engine's multi-wave kernels):
    gi_group [nv] -- coordinate group of every generalised velocity index (0 = trunk incl. the floating base, l = limb l),
    body_block [nb] -- the block (wavefront) that sweeps the limit rows / ground contacts of each body, nblk (the self contacts are
    block nblk - 1 when self_collision); wave_caps: also kmax_blk, the per-block caps of the ground contacts."""
    limb, limbs, role_of_limb, trunk_role, nrole = wave_roles(spec)
    off = 0 if spec.fixed_base else 6
    gi_group = [0] * (off + spec.nd)
    for d in range(spec.nd):
        gi_group[off + d] = limb[int(spec.dof_body[d])]
    body_block = [trunk_role if role_of_limb[limb[b]] < 0 else role_of_limb[limb[b]] for b in range(spec.nb)]
    # a model with self-collision tables reserves its last wave for the self contacts (block nrole - 1, whether or not they are switched on)
    out = dict(gi_group=gi_group, body_block=body_block, nblk=nrole if _has_selfcol(spec) else nrole + (1 if self_collision else 0))
    if wave_caps:       # the compact-store form keeps its ground contacts per wave
        out["kmax_blk"] = wave_contact_caps(spec)
    return out


def hand_limb_caps(spec: ModelSpec):
    """Object contacts each LIMB of a manipulator keeps per env in the finger-per-wave sub-step (csrc/core/hand_engine_mw.hpp gives every
    limb its own contact slots in LDS, rows in the fixed shape [limb dofs | wrist dofs]).  What 80 KB of LDS per 32 envs hold (a slot
    is 3 (n_limb + 2) + 10 floats), dealt by where contacts were measured on random-policy states (tools/hand_solver_study.py: the palm
    limb never holds more than 5, the little finger -- with its metacarpal it lies under the cube -- more than 3 in 10 % and more than 5
    in 0.7 % of the sub-steps, the thumb more than 3 in 1.6 %, the other fingers more than 2 in 0.1 %): 5 for the limb of the root
    body (forearm, wrist, palm: the palm alone may hold a manifold of 4); five-joint fingers 5, then 4; four-joint fingers 3, then 2.
    -> [nlimb]"""
    _, limbs = limb_paths(spec)
    nd = [sum(1 for d in range(spec.nd) if int(spec.dof_body[d]) in bodies) for bodies in limbs]
    caps = [5] + [0] * (len(limbs) - 1)
    n5 = n4 = 0
    for l in range(1, len(limbs)):
        if nd[l] >= 5:
            caps[l] = 5 if n5 == 0 else 4
            n5 += 1
        else:
            caps[l] = 3 if n4 == 0 else 2
            n4 += 1
    return caps


def hand_solver_blocks(spec: ModelSpec):
    """solver_blocks() for a fixed-base manipulator + free object (oracle/hand.c solver 1): additionally limb_of_body [nb] and
    limb_cap [nlimb] (hand_limb_caps)."""
    limb, _ = limb_paths(spec)
    out = solver_blocks(spec)
    out.update(limb_of_body=[int(x) for x in limb], limb_cap=hand_limb_caps(spec))
    return out


def self_collision_groups(spec: ModelSpec, pairs=None):
    """Run-time organisation of the self-collision pairs: one *group* per pair of limbs (and per limb with non-adjacent bodies of
    its own).  A group carries at most one contact per sub-step -- its deepest capsule pair -- and all its body pairs share one
    constraint-row code path whose chain is the union of the two limb tips' kinematic chains.
    -> list of dicts {tip_a, tip_b, pairs: [(capsule_a, capsule_b)]} with body(capsule_a) in limb A, body(capsule_b) in limb B."""
    pairs = self_collision_pairs(spec) if pairs is None else pairs
    gb = collision_capsules(spec)[0]
    limb, limbs = limb_paths(spec)
    groups = {}
    for a, b, gps in pairs:
        la, lb = limb[a], limb[b]
        for (i, j) in gps:
            # order the two sides by limb id so that a group has one orientation (intra-limb: ancestor first)
            if (la, a) <= (lb, b):
                groups.setdefault((la, lb), []).append((int(i), int(j)))
            else:
                groups.setdefault((lb, la), []).append((int(j), int(i)))
    # Order: by limb pair, but the groups whose two limbs BOTH belong to the trunk role of the limb-per-wave form (the trunk itself, or a limb that
    # role owns: the Humanoid's arms) come LAST.  That role examines them itself, from the sphere centres its own tree pass leaves behind, while
    # the pair role works through the others (csrc/core/engine_mwc.hpp, round 6); slots are dealt in table order by every form and by the oracle.
    _, _, role_of_limb, trunk_role, _ = wave_roles(spec, pair_role=True)
    own = lambda l: l == 0 or role_of_limb[l] == trunk_role     # noqa: E731
    out = []
    for (la, lb), gps in sorted(groups.items(), key=lambda kv: ((own(kv[0][0]) and own(kv[0][1])), kv[0])):
        out.append(dict(limb_a=la, limb_b=lb, tip_a=limbs[la][-1], tip_b=limbs[lb][-1], pairs=sorted(gps)))
    return out


def self_collision_tables(spec: ModelSpec, **kw):
    """Everything the engine and the oracle need for self-collision, as plain lists (stored next to the model as <name>_selfcol.json
    by tools/compile_models.py): the collision capsules and, grouped by limb pair, the capsule pairs that can touch."""
    gb, p0, p1, rad, mu, gi = collision_capsules(spec)
    groups = self_collision_groups(spec, self_collision_pairs(spec, **kw))
    return dict(cap_body=[int(x) for x in gb], cap_p0=[[float(v) for v in p] for p in p0], cap_p1=[[float(v) for v in p] for p in p1],
                cap_rad=[float(x) for x in rad], cap_mu=[float(x) for x in mu], cap_geom=[int(x) for x in gi],
                groups=[dict(tip_a=int(g["tip_a"]), tip_b=int(g["tip_b"]), pairs=[[int(i), int(j)] for i, j in g["pairs"]]) for g in groups])

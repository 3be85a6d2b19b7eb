"""`isaacgym.gymapi` stand-in on the MI355X-native engine.

Replaces the closed `isaacgym.gymapi` for the call sites of the reference's base class and locomotion / Cartpole tasks
(reference isaacgymenvs/tasks/base/vec_task.py:247-262,337-341,379-386,514-562; tasks/ant.py:77-95,116-212,233-285;
tasks/cartpole.py:48-163; tasks/humanoid.py).  What the calls mean here:

  acquire_gym / create_sim / add_ground / load_asset / create_env / create_actor ...   record what the task asks for;
  prepare_sim                   creates the native engine (isaacgymenvs_amd.native.Engine) for the recorded actor on `num_envs` envs;
  acquire_*_tensor              returns an AoS torch tensor ([num_actors, 13], [num_dofs, 2], [num_sensors, 6]) -- the layout the
                                tasks `.view()`; the engine's own arena is SoA, so these are copies with gym's semantics:
  refresh_*_tensor              engine -> tensor;   set_*_tensor(_indexed)   tensor -> engine (immediately, as gym's CPU pipeline does);
  simulate                      mi_engine_simulate (one dt of `substeps` sub-steps);   fetch_results   nothing to fetch;
  viewer / camera / colour calls do nothing (headless engine).

Assets are resolved to the models compiled into the engine by file name (nv_ant.xml, nv_humanoid.xml, cartpole.urdf, anymal_minimal.urdf,
shadow_hand.xml; the ShadowHand task's free objects cube_multicolor.urdf / egg.xml / pen.xml): the engine is specialised per robot at build
time (isaacgymenvs_amd/codegen.py), it does not load arbitrary files at run time.

Envs with several actors (ShadowHand: hand, object, goal object -- reference shadow_hand.py:356-381): the sim-domain actor index of actor k
of env i is A i + k (A actors per env), the root tensor is [A N, 13], the rigid-body tensor lists the articulation's bodies, then one
body per free object.  The object's root state is the engine's `object_state`; the goal object has no physics (its own collision group,
gravity off): its root state lives in the shim's tensor only.
Terrain (AnymalTerrain, reference anymal_terrain.py:203-215): `add_triangle_mesh` receives the vertices `isaacgym.terrain_utils.
convert_heightfield_to_trimesh` made, checks that they are those, and hands the engine the height field they came from.
"""
from __future__ import annotations

import os

import numpy as np
import torch

# ---------------------------------------------------------------------------------------------------------------- enums / constants
SIM_PHYSX, SIM_FLEX = 0, 1
UP_AXIS_Y, UP_AXIS_Z = 0, 1
DOF_MODE_NONE, DOF_MODE_POS, DOF_MODE_VEL, DOF_MODE_EFFORT = 0, 1, 2, 3
MESH_NONE, MESH_COLLISION, MESH_VISUAL, MESH_VISUAL_AND_COLLISION = 0, 1, 2, 3
DOMAIN_SIM, DOMAIN_ENV, DOMAIN_ACTOR = 0, 1, 2
INVALID_HANDLE = -1      # what the find_* calls return for an unknown name (trifinger.py:1146,1158 compare against it)
ENV_SPACE, LOCAL_SPACE, GLOBAL_SPACE = 0, 1, 2
KEY_ESCAPE, KEY_V, KEY_R = 256, 86, 82
STATE_NONE, STATE_POS, STATE_VEL, STATE_ALL = 0, 1, 2, 3
_MI_SHIM = True


class ContactCollection(int):
    """CC_NEVER 0, CC_LAST_SUBSTEP 1, CC_ALL_SUBSTEPS 2 (vec_task.py:549)"""


class Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __add__(self, o):
        return Vec3(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __neg__(self):
        return Vec3(-self.x, -self.y, -self.z)

    def __mul__(self, k):            # scalar (ball_balance.py:180 `(upper_leg_from + upper_leg_to) * 0.5`, ingenuity.py:191 `rotor_separation * i`)
        return Vec3(self.x * float(k), self.y * float(k), self.z * float(k))

    __rmul__ = __mul__

    def __truediv__(self, k):
        return Vec3(self.x / float(k), self.y / float(k), self.z / float(k))

    def dot(self, o):
        return self.x * o.x + self.y * o.y + self.z * o.z

    def cross(self, o):
        return Vec3(self.y * o.z - self.z * o.y, self.z * o.x - self.x * o.z, self.x * o.y - self.y * o.x)

    def length(self):
        return float(np.sqrt(self.dot(self)))

    def length_sq(self):
        return float(self.dot(self))

    def normalize(self):
        n = self.length() or 1.0
        return Vec3(self.x / n, self.y / n, self.z / n)

    def __repr__(self):
        return f"Vec3({self.x}, {self.y}, {self.z})"


class Quat:
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = float(x), float(y), float(z), float(w)

    @staticmethod
    def from_axis_angle(axis, angle):
        s = np.sin(0.5 * angle)
        n = np.sqrt(axis.x ** 2 + axis.y ** 2 + axis.z ** 2) or 1.0
        return Quat(axis.x / n * s, axis.y / n * s, axis.z / n * s, np.cos(0.5 * angle))

    @staticmethod
    def from_euler_zyx(x, y, z):
        """the rotation Rz(z) Ry(y) Rx(x): the arguments are the angles about x, y and z in that order (only this reading puts the feet of the
        BallBalance legs on their attractor targets, ball_balance.py:181,293-297; assets/procedural.py restates the same)"""
        cz, sz, cy, sy, cx, sx = np.cos(z / 2), np.sin(z / 2), np.cos(y / 2), np.sin(y / 2), np.cos(x / 2), np.sin(x / 2)
        return Quat(sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz, cx * cy * cz + sx * sy * sz)

    def to_euler_zyx(self):
        """(x, y, z) such that from_euler_zyx(x, y, z) is this rotation"""
        x, y, z, w = self.x, self.y, self.z, self.w
        return (float(np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))), float(np.arcsin(np.clip(2 * (w * y - z * x), -1.0, 1.0))),
                float(np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))))

    def __mul__(self, o):          # Hamilton product (allegro_hand.py:283 composes the hand's start rotation from three axis-angle quats)
        return Quat(self.w * o.x + self.x * o.w + self.y * o.z - self.z * o.y, self.w * o.y - self.x * o.z + self.y * o.w + self.z * o.x,
                    self.w * o.z + self.x * o.y - self.y * o.x + self.z * o.w, self.w * o.w - self.x * o.x - self.y * o.y - self.z * o.z)

    def rotate(self, v):           # quadcopter.py:160 `rotor_arm_quat.rotate(rotor_arm_offset)`: v + 2 w (q x v) + 2 q x (q x v)
        q = Vec3(self.x, self.y, self.z)
        t = q.cross(v) * 2.0
        return v + t * self.w + q.cross(t)

    def inverse(self):
        n = self.x ** 2 + self.y ** 2 + self.z ** 2 + self.w ** 2 or 1.0
        return Quat(-self.x / n, -self.y / n, -self.z / n, self.w / n)

    def normalize(self):
        n = np.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2 + self.w ** 2) or 1.0
        return Quat(self.x / n, self.y / n, self.z / n, self.w / n)

    def __repr__(self):
        return f"Quat({self.x}, {self.y}, {self.z}, {self.w})"


class Transform:
    def __init__(self, p=None, r=None):
        self.p = p if p is not None else Vec3()
        self.r = r if r is not None else Quat()

    def transform_point(self, v):
        return self.r.rotate(v) + self.p

    def transform_vector(self, v):
        return self.r.rotate(v)

    def inverse(self):
        ri = self.r.inverse()
        return Transform(-ri.rotate(self.p), ri)

    def __mul__(self, o):
        return Transform(self.transform_point(o.p), self.r * o.r)


# rigid-body attractors (ball_balance.py:285-300): a spring-damper that pulls a point of a body to a world target
AXIS_NONE, AXIS_X, AXIS_Y, AXIS_Z, AXIS_TWIST, AXIS_SWING_1, AXIS_SWING_2 = 0, 1, 2, 4, 8, 16, 32
AXIS_TRANSLATION, AXIS_ROTATION, AXIS_ALL = 7, 56, 63


class AttractorProperties:
    def __init__(self):
        self.stiffness, self.damping, self.axes, self.rigid_handle = 0.0, 0.0, AXIS_NONE, -1
        self.target, self.offset = Transform(), Transform()


class VhacdParams:        # convex decomposition settings of mesh assets (trifinger.py:1120): recorded, the engine samples meshes by spheres
    def __init__(self):
        self.resolution, self.max_convex_hulls, self.max_num_vertices_per_ch = 100000, 64, 64


class _Bag:
    """attribute bag with defaults: AssetOptions, PlaneParams, CameraProperties, the physx / flex sub-structs of SimParams"""

    def __init__(self, **defaults):
        self.__dict__.update(defaults)


class AssetOptions(_Bag):
    def __init__(self):
        super().__init__(default_dof_drive_mode=DOF_MODE_NONE, angular_damping=0.5, linear_damping=0.0, fix_base_link=False,
                         collapse_fixed_joints=False, density=1000.0, armature=0.0, thickness=0.02, disable_gravity=False,
                         replace_cylinder_with_capsule=False, flip_visual_attachments=False, max_angular_velocity=64.0,
                         max_linear_velocity=1000.0, use_mesh_materials=False, slices_per_cylinder=20, vhacd_enabled=False,
                         vhacd_params=VhacdParams())


class PlaneParams(_Bag):
    def __init__(self):
        super().__init__(normal=Vec3(0.0, 0.0, 1.0), distance=0.0, static_friction=1.0, dynamic_friction=1.0, restitution=0.0)


class TriangleMeshParams(_Bag):
    def __init__(self):
        super().__init__(nb_vertices=0, nb_triangles=0, transform=Transform(), static_friction=1.0, dynamic_friction=1.0, restitution=0.0)


class _Scalars(_Bag):
    """a property struct whose numeric fields are C floats in the simulator's bindings: a one-element array assigned to one (what
    utils/dr_utils.py:195-208 apply_random_samples does: `setattr(prop, attr, og * sample)` with a sample of shape (1,)) reads back as a float"""

    def __setattr__(self, k, val):
        if isinstance(val, np.ndarray) and val.size == 1:
            val = float(val.reshape(-1)[0])
        elif torch.is_tensor(val) and val.numel() == 1:
            val = float(val)
        object.__setattr__(self, k, val)


class RigidShapeProperties(_Scalars):
    def __init__(self, friction=1.0):
        super().__init__(friction=friction, rolling_friction=0.0, torsion_friction=0.0, restitution=0.0, compliance=0.0, thickness=0.0)


class RigidBodyProperties(_Scalars):
    def __init__(self, mass=0.0):
        super().__init__(mass=mass, invMass=(1.0 / mass if mass > 0 else 0.0), com=Vec3(), inertia=None)


class TendonProperties(_Scalars):
    def __init__(self, stiffness=0.0, damping=0.0, limit_stiffness=0.0):
        super().__init__(type=0, stiffness=stiffness, damping=damping, fixed_spring_rest_length=0.0, fixed_lower_limit=0.0, fixed_upper_limit=0.0,
                         limit_stiffness=limit_stiffness, is_fixed_limited=True, num_attachments=2)


class CameraProperties(_Bag):
    def __init__(self):
        super().__init__(width=1600, height=900)


class SimParams:
    def __init__(self):
        self.dt = 1.0 / 60.0
        self.substeps = 2
        self.up_axis = UP_AXIS_Y
        self.gravity = Vec3(0.0, -9.8, 0.0)
        self.use_gpu_pipeline = False
        self.num_client_threads = 0
        self.physx = _Bag(num_threads=4, solver_type=1, use_gpu=False, num_position_iterations=4, num_velocity_iterations=0,
                          contact_offset=0.02, rest_offset=0.0, bounce_threshold_velocity=0.2, max_depenetration_velocity=100.0,
                          default_buffer_size_multiplier=2.0, max_gpu_contact_pairs=1024 * 1024, num_subscenes=0,
                          contact_collection=ContactCollection(2), friction_offset_threshold=0.04, friction_correlation_distance=0.025)
        self.flex = _Bag()


# ---------------------------------------------------------------------------------------------------------------- recorded objects
_MODEL_OF_FILE = {"nv_ant.xml": ("ant", "Ant"), "nv_humanoid.xml": ("humanoid", "Humanoid"), "cartpole.urdf": ("cartpole", "Cartpole"),
                  "anymal_minimal.urdf": ("anymal", "AnymalTerrain"), "shadow_hand.xml": ("shadow_hand", "ShadowHand"),
                  # anymal.py:168 loads anymal.urdf: the same robot (13 bodies, 12 dofs, same names and order) with more collision shapes
                  # (boxes / cylinders on base, hips, thighs, shanks) -- the engine's flat Anymal runs the compiled model's contact set (feet,
                  # knees, base capsule); _Asset checks the kinematic tree of the file against it (tests/test_gymapi_shim.py)
                  "anymal.urdf": ("anymal", "Anymal"),
                  # the files the tasks write themselves before loading them (quadcopter.py:198, ingenuity.py:231, ball_balance.py:218);
                  # assets/procedural.py restates the generators, the compiled models come from those restatements
                  "quadcopter.xml": ("quadcopter", "Quadcopter"), "ingenuity.xml": ("ingenuity", "Ingenuity"),
                  "balance_bot.xml": ("balance_bot", "BallBalance"),
                  # cfg/task/AllegroHand.yaml asset.assetFileName (allegro_hand.py:212-214)
                  "allegro_touch_sensor.urdf": ("allegro_hand", "AllegroHand")}
# the hand tasks' free objects (shadow_hand.py:86-96, allegro_hand.py:88-92 + AllegroHand.yaml assetFileNameBlock): one body, no dof
_OBJECT_OF_FILE = {"cube_multicolor.urdf": "block", "cube_multicolor_allegro.urdf": "block", "egg.xml": "egg", "pen.xml": "pen"}
_HAND_TASKS = ("ShadowHand", "AllegroHand")
# files that describe a compiled robot with a richer collision set: accepted when the kinematic tree is the compiled model's (anymal.py:168)
_SAME_ROBOT_MORE_SHAPES = ("anymal.urdf",)


class _ActuatorProps:
    def __init__(self, gear):
        self.motor_effort = float(gear)
        self.kp = self.kv = 0.0


def _single_box_urdf(path):
    """A URDF that is ONE rigid body -- no joint that moves, one link with collision geometry -- read as a box: what the reference's table-top
    tasks load their objects and stages from (trifinger.py:1169-1255: table_without_border.urdf, cube_multicolor_rrc.urdf; `create_box` is the
    other way to get one).  A <box> is itself; a <mesh> becomes the box around its vertices (a stated approximation: exact for the plates and
    cubes these files hold, wrong for a concave shape such as a ring wall -- the caller warns).  Returns None for anything else, else
    dict(link, dims[3], pos[3], quat[4] -- the box's centre and axes in the link frame --, mass or None, inertia[3] or None, mesh: bool,
    hollow: bool -- a mesh whose bounding box's vertical axis crosses none of its triangles: a ring; such a body gets NO collision shape)."""
    import xml.etree.ElementTree as ET
    from ...assets.model import _floats, _resolve_mesh, rpy_to_quat
    if not (os.path.isfile(path) and path.lower().endswith(".urdf")):
        return None
    try:
        root = ET.parse(path).getroot()
    except ET.ParseError:
        return None
    if any(j.get("type", "fixed") != "fixed" for j in root.findall("joint")):
        return None
    links = [ln for ln in root.findall("link") if ln.find("collision") is not None]
    if len(links) != 1 or len(links[0].findall("collision")) != 1:
        return None
    ln = links[0]
    col = ln.find("collision")
    org, geo = col.find("origin"), col.find("geometry")
    if geo is None or len(geo) == 0:
        return None
    pos = _floats(org.get("xyz", "0 0 0"), 3) if org is not None else np.zeros(3)
    quat = rpy_to_quat(*_floats(org.get("rpy", "0 0 0"), 3)) if org is not None else np.array([0.0, 0.0, 0.0, 1.0])
    g = geo[0]
    hollow = False
    if g.tag == "box":
        dims, mesh = _floats(g.get("size"), 3), False
    elif g.tag == "mesh":
        from ...assets.mesh import load_mesh
        from ...assets.model import quat_to_mat
        mpath = _resolve_mesh(g.get("filename"), path)
        if mpath is None:
            raise FileNotFoundError(f"{path}: collision mesh {g.get('filename')} not found")
        V, F = load_mesh(mpath)
        V = np.asarray(V, float) * (_floats(g.get("scale"), 3) if g.get("scale") else 1.0)
        lo, hi = V.min(0), V.max(0)
        dims, mesh = hi - lo, True
        # hollow (a ring wall such as trifinger's high_table_boundary.stl: the arena is INSIDE its bounding box): the line through the middle of the
        # bounding box along its shortest horizontal... along z crosses no triangle of the mesh
        c2 = 0.5 * (lo + hi)[:2]
        T = V[np.asarray(F, int)][:, :, :2] - c2                     # [faces, 3, 2]: the triangles' xy corners relative to that line
        d0 = T[:, 0, 0] * T[:, 1, 1] - T[:, 0, 1] * T[:, 1, 0]
        d1 = T[:, 1, 0] * T[:, 2, 1] - T[:, 1, 1] * T[:, 2, 0]
        d2 = T[:, 2, 0] * T[:, 0, 1] - T[:, 2, 1] * T[:, 0, 0]
        hollow = not bool((((d0 >= 0) & (d1 >= 0) & (d2 >= 0)) | ((d0 <= 0) & (d1 <= 0) & (d2 <= 0))).any())
        pos = pos + quat_to_mat(quat) @ (0.5 * (lo + hi))
    else:
        return None
    mass = inertia = None
    ine = ln.find("inertial")
    if ine is not None and ine.find("mass") is not None:
        mass = float(ine.find("mass").get("value"))
        it = ine.find("inertia")
        if it is not None and all(abs(float(it.get(k, 0.0))) < 1e-12 for k in ("ixy", "ixz", "iyz")):
            inertia = [float(it.get(k)) for k in ("ixx", "iyy", "izz")]
    return dict(link=ln.get("name"), dims=[float(d) for d in dims], pos=[float(x) for x in pos], quat=[float(x) for x in quat],
                mass=mass if (mass is not None and mass > 0) else None, inertia=inertia, mesh=mesh, hollow=hollow)


class _Asset:
    @classmethod
    def primitive(cls, shape, dims, options):
        """gym.create_sphere / create_box / create_capsule: a one-body asset without dofs (ball_balance.py:277 the ball, ingenuity.py:254 the
        target marker).  Which of the engine's free objects it is follows from the task of the articulated actor it shares its envs with."""
        a = cls.__new__(cls)
        a.options, a.sensors, a.shape_friction, a.tendon_props = options, [], None, None
        a.spec, a.object_type, a.dims, a.generic = None, shape, tuple(float(d) for d in dims), False
        a.body_names, a.body_dyn, a.nshapes = [shape], np.zeros(1, np.int64), 1
        a.path = None
        a.box_pos, a.box_quat, a.mass, a.inertia = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), None, None
        return a

    def __init__(self, path, options):
        from ...registry import load_extras, load_model, load_selfcol, sensor_bodies
        key = os.path.basename(path)
        self.options = options
        self.path = path
        self.file_spec = None
        self.sensors = []                      # rigid-body indices in the order create_asset_force_sensor was called
        self.shape_friction = None             # set_asset_rigid_shape_properties: friction of the asset's shapes for the actors created next
        self.tendon_props = None
        self.spec, self.object_type = None, None
        self.generic = False                   # a robot no task of the engine was written for (the Articulation task)
        if key in _OBJECT_OF_FILE:
            self.object_type = _OBJECT_OF_FILE[key]
            self.body_names, self.body_dyn = ["object"], np.zeros(1, np.int64)
            self.nshapes = 1
            return
        from ...assets import runtime
        self.variant = False                   # a file that differs from the compiled model: its own library (assets/runtime.py)
        box = None if key in _MODEL_OF_FILE else _single_box_urdf(path)
        if box is not None:
            # one rigid body: a box actor of a scene, like gym.create_box's (static with fix_base_link, else free)
            import warnings
            if box["hollow"]:
                warnings.warn(f"gym.load_asset: {key}: the collision mesh of the single body is a ring (its bounding box would swallow what stands inside): "
                              f"the body is created WITHOUT a collision shape")
            elif box["mesh"]:
                warnings.warn(f"gym.load_asset: {key}: the collision mesh of the single body is simulated as its bounding box "
                              f"({box['dims'][0]:.3f} x {box['dims'][1]:.3f} x {box['dims'][2]:.3f} m)")
            self.hollow = bool(box["hollow"])
            self.object_type, self.dims, self.generic = "box", tuple(box["dims"]), False
            self.box_pos, self.box_quat, self.mass, self.inertia = tuple(box["pos"]), tuple(box["quat"]), box["mass"], box["inertia"]
            self.body_names, self.body_dyn, self.nshapes = [box["link"]], np.zeros(1, np.int64), 1
            return
        if key in _MODEL_OF_FILE:
            self.model_name, self.task = _MODEL_OF_FILE[key]
            self.spec = load_model(self.model_name)
            if os.path.isfile(path) and self.model_name in runtime.ASSET_OPTIONS and runtime.load_selfcol(self.model_name) is None:
                spec = runtime.parse(path, self.model_name)          # the file itself, not the copy compiled at build time
                if runtime.header_text(self.model_name, spec) != runtime.header_text(self.model_name, self.spec):
                    if key in _SAME_ROBOT_MORE_SHAPES and runtime.same_tree(spec, self.spec):
                        self.file_spec = spec        # the file as parsed; the engine runs the compiled model's contact set and constants
                    elif not runtime.same_topology(spec, self.spec):
                        raise NotImplementedError(f"gym.load_asset: {path} does not have the kinematic tree of the compiled {self.model_name} model")
                    else:
                        self.spec, self.variant = spec, True
        elif os.path.isfile(path):
            try:
                self.model_name, self.spec = runtime.match_model(path)   # a file of another name with the tree of a compiled model
                self.task = runtime.TASK_OF_MODEL[self.model_name]
                self.variant = runtime.header_text(self.model_name, self.spec) != runtime.header_text(self.model_name, load_model(self.model_name))
            except NotImplementedError:
                # a kinematic tree of its own: the engine's Articulation task (gym.simulate and the state tensors, no fused task kernels), compiled
                # for this robot when prepare_sim knows its drives and sensors (assets/runtime.py variant_library; the stock library carries
                # mjcf/amp_humanoid.xml as HumanoidAMP configures it)
                self.model_name, self.task, self.generic = runtime.GENERIC_MODEL, "Articulation", True
                self.spec = runtime.parse_generic(path, options)
        else:
            raise NotImplementedError(f"gym.load_asset: {key} has no model compiled into the engine (available: {sorted(_MODEL_OF_FILE)}) and "
                                      f"is not a readable file; robots are specialised per model (isaacgymenvs_amd/codegen.py, assets/runtime.py)")
        # the bodies gym lists: with collapse_fixed_joints the welded links are gone (= the engine's bodies), otherwise every link of the
        # file, each riding on the engine body it is welded to (api_body_dyn) at a fixed offset (api_body_pos / api_body_quat)
        if getattr(options, "collapse_fixed_joints", False):
            self.body_names = list(self.spec.body_names)
            self.body_dyn = np.arange(self.spec.nb)
            self.body_off_p = np.zeros((self.spec.nb, 3)); self.body_off_q = np.tile([0.0, 0.0, 0.0, 1.0], (self.spec.nb, 1))
        else:
            self.body_names = list(self.spec.api_body_names)
            self.body_dyn = np.asarray(self.spec.api_body_dyn, np.int64)
            self.body_off_p = np.asarray(self.spec.api_body_pos, float); self.body_off_q = np.asarray(self.spec.api_body_quat, float)
        self.nshapes = len(self.spec.geom_body)
        self.engine_sensor_bodies = [] if self.generic else [self.body_names.index(self.spec.body_names[b]) for b in sensor_bodies(self.model_name, self.spec)]
        self.has_self_collision = load_selfcol(self.model_name) is not None
        self.extras = load_extras(self.model_name) if self.model_name in ("shadow_hand", "allegro_hand") else None


class _ActorProps:
    """Per-env actor properties as the reference's domain randomisation writes them: VecTask.apply_randomizations (vec_task.py:752-828) walks
    every env and calls gym.get_actor_*_properties / set_actor_*_properties / set_actor_scale on it.  PhysX keeps one property struct per actor; the
    engine keeps per-env FACTORS of the compiled model's constants in tensors the sub-step reads (`actor_scale`, `dof_limit_shift`, `friction`:
    csrc/core/engine.hpp AS_*, csrc/core/hand_engine.hpp HS_*).  The setters turn what they are given into factors relative to the actor's own
    values, staged here on the host (one row per env: the reference calls them env by env); Gym._flush_props writes the rows that changed to the
    engine before the next simulate()."""

    def __init__(self, sim):
        robot = sim.slots[sim.robot]["asset"]
        self.n, self.nb, self.nd = 0, robot.spec.nb, robot.spec.nd
        self.body = np.ones((0, self.nb))                          # mass (and with it inertia) factor per engine body
        self.damp, self.stiff, self.arm = (np.ones((0, self.nd)) for _ in range(3))
        self.lower, self.upper = np.zeros((0, self.nd)), np.zeros((0, self.nd))       # shifts of the joint limits
        self.tendon_k, self.tendon_d, self.obj_mass, self.obj_scale = (np.ones(0) for _ in range(4))
        self.mu = {}                                               # actor slot -> [n] shape friction (nan: the asset's own)
        self.dirty = False
        self.grow(len(sim.envs))

    def grow(self, n):
        if n <= self.n:
            return
        pad = n - self.n
        for k in ("body", "damp", "stiff", "arm"):
            a = getattr(self, k)
            setattr(self, k, np.concatenate([a, np.ones((pad, a.shape[1]))]))
        for k in ("lower", "upper"):
            a = getattr(self, k)
            setattr(self, k, np.concatenate([a, np.zeros((pad, a.shape[1]))]))
        for k in ("tendon_k", "tendon_d", "obj_mass", "obj_scale"):
            setattr(self, k, np.concatenate([getattr(self, k), np.ones(pad)]))
        for k in self.mu:
            self.mu[k] = np.concatenate([self.mu[k], np.full(pad, np.nan)])
        self.n = n

    def friction(self, slot):
        if slot not in self.mu:
            self.mu[slot] = np.full(self.n, np.nan)
        return self.mu[slot]


def _ratio(new, base):
    """element-wise new / base where the base value is positive, 1 elsewhere (a factor cannot express a change of a zero)"""
    new, base = np.asarray(new, np.float64), np.asarray(base, np.float64)
    return np.where(base > 0, new / np.where(base > 0, base, 1.0), 1.0)


class _Env:
    def __init__(self, sim, index):
        self.sim, self.index = sim, index
        self.actors = []                       # names, in creation order (the same for every env)


class _Sim:
    def __init__(self, compute_device, graphics_device, physics_engine, params):
        self.compute_device, self.params = compute_device, params
        self.device = f"cuda:{compute_device}" if params.use_gpu_pipeline else "cpu"
        self.plane = None
        self.terrain = None                    # add_triangle_mesh: the height field behind the mesh
        self.envs = []
        self.slots = []                        # per actor of an env: dict(asset, name, filter, poses [N][7], friction {env: mu})
        self.engine = None
        self.frame = 0
        self.bufs = {}
        self.one_shot_force = False
        self.attractors = []                   # create_rigid_body_attractor, env 0
        self.walked = False                    # a per-env property call has reached an env other than the newest: creation is over
        self.scene = None                      # prepare_sim: slot -> ("free" | "static", index) of the box actors the engine simulates beside a fixed-base actor
        self.props = None                      # _ActorProps: what the per-env property setters wrote (domain randomisation), staged on the host

    # the articulated actor (the engine's robot) and the free objects
    @property
    def robot(self):
        for k, sl in enumerate(self.slots):
            if sl["asset"].spec is not None:
                return k
        raise RuntimeError("no articulated actor was created")

    @property
    def asset(self):
        return self.slots[self.robot]["asset"]

    @property
    def nactors(self):
        return len(self.slots)


class _Terrain:
    """what isaacgymenvs_amd.native.Engine takes as `terrain` (isaacgymenvs_amd/tasks/terrain.py attribute names)"""

    def __init__(self, hf, hscale, vscale, border, slope_threshold):
        self.heightsamples, self.horizontal_scale, self.vertical_scale, self.border_size = hf, hscale, vscale, border
        self.slope_threshold = slope_threshold
        self.env_origins = np.zeros((1, 1, 3))
        self.env_length, self.max_init_level = 8.0, 0


class Gym:
    """the object `gymapi.acquire_gym()` returns"""

    # ------------------------------------------------------------------ setup
    def create_sim(self, compute_device=0, graphics_device=0, physics_engine=SIM_PHYSX, params=None):
        return _Sim(compute_device, graphics_device, physics_engine, params or SimParams())

    def add_ground(self, sim, plane_params):
        sim.plane = plane_params

    def add_triangle_mesh(self, sim, vertices, triangles, params):
        """anymal_terrain.py:203-215.  The engine walks on the height field the mesh was generated from (terrain_utils remembers it); the
        vertices handed in must be that conversion's (same surface), the transform a pure shift by -border in x and y."""
        from . import terrain_utils
        conv = terrain_utils._last_conversion
        v = np.asarray(vertices, np.float32).reshape(-1, 3)
        if conv is None or conv["vertices"].shape != v.shape or int(params.nb_vertices) != v.shape[0]:
            raise NotImplementedError("gym.add_triangle_mesh: only meshes made by isaacgym.terrain_utils.convert_heightfield_to_trimesh (the engine's "
                                      "ground is a height field)")
        if not np.array_equal(conv["vertices"], v):
            raise ValueError("gym.add_triangle_mesh: the vertices differ from the height field's conversion")
        if int(params.nb_triangles) != np.asarray(triangles).size // 3 or params.transform.p.x != params.transform.p.y or params.transform.p.z != 0.0:
            raise NotImplementedError("gym.add_triangle_mesh: transform must be (-border, -border, 0)")
        sim.terrain = _Terrain(conv["height_field"], conv["horizontal_scale"], conv["vertical_scale"], -float(params.transform.p.x),
                               conv["slope_threshold"])
        sim.plane = _Bag(static_friction=float(params.static_friction), dynamic_friction=float(params.dynamic_friction),
                         restitution=float(params.restitution))

    def load_asset(self, sim, rootpath, filename, options=None):
        import copy
        return _Asset(os.path.join(rootpath, filename), copy.copy(options) if options is not None else AssetOptions())

    def get_asset_dof_count(self, asset):
        return asset.spec.nd if asset.spec is not None else 0

    def get_asset_rigid_body_count(self, asset):
        return len(asset.body_names)

    def get_asset_joint_count(self, asset):
        # joints of the file incl. the fixed ones that weld bodies: one per non-root body (nv_humanoid.xml: 15 for 16 bodies)
        return len(asset.body_names) - 1

    def get_asset_rigid_shape_count(self, asset):
        return asset.nshapes

    def get_asset_dof_names(self, asset):
        return list(asset.spec.dof_names)

    def get_asset_rigid_body_names(self, asset):
        return list(asset.body_names)

    def get_asset_rigid_body_name(self, asset, index):
        return asset.body_names[index]

    def find_asset_rigid_body_index(self, asset, name):
        return asset.body_names.index(name)

    def find_asset_dof_index(self, asset, name):
        return list(asset.spec.dof_names).index(name)

    def get_asset_actuator_count(self, asset):
        if asset.extras is not None:
            return len(asset.extras["actuated_dofs"])
        return len(asset.spec.act_gear)

    def get_asset_actuator_properties(self, asset):
        return [_ActuatorProps(g) for g in asset.spec.act_gear]

    def get_asset_actuator_joint_name(self, asset, index):
        if asset.extras is not None:
            return asset.spec.dof_names[int(asset.extras["actuated_dofs"][index])]
        return asset.spec.dof_names[int(asset.spec.act_dof[index])]

    def get_asset_dof_properties(self, asset):
        out = _dof_properties(asset.spec, _drive_mode(asset))
        if asset.extras is not None and "dof_kp" in asset.extras:       # the hands' position actuators: the dof's stiffness is the drive's kp
            out["stiffness"] = np.asarray(asset.extras["dof_kp"], np.float32)
        return out

    def get_asset_rigid_shape_properties(self, asset):
        mu = asset.spec.geom_friction if asset.spec is not None else [1.0]
        return [RigidShapeProperties(float(np.ravel(mu[i])[0]) if i < len(mu) else 1.0) for i in range(asset.nshapes)]

    def set_asset_rigid_shape_properties(self, asset, props):
        asset.shape_friction = float(np.ravel(np.asarray(props[0].friction.cpu() if hasattr(props[0].friction, "cpu") else props[0].friction))[0])
        return True

    # fixed tendons of the Shadow Hand (shadow_hand.py:253-266)
    def get_asset_tendon_count(self, asset):
        return len(asset.extras["tendons"]) if asset.extras is not None else 0

    def get_asset_tendon_name(self, asset, index):
        return asset.extras["tendons"][index]["name"]

    def get_asset_tendon_properties(self, asset):
        return [TendonProperties() for _ in range(self.get_asset_tendon_count(asset))]

    def set_asset_tendon_properties(self, asset, props):
        asset.tendon_props = [(float(pr.limit_stiffness), float(pr.damping)) for pr in props]
        return True

    def create_asset_force_sensor(self, asset, body_idx, local_pose, props=None):
        asset.sensors.append(int(body_idx))
        if not hasattr(asset, "sensor_poses"):
            asset.sensor_poses = []
        asset.sensor_poses.append(local_pose)
        return len(asset.sensors) - 1

    # primitive one-body assets (ball_balance.py:274-277, ingenuity.py:253-254)
    def create_sphere(self, sim, radius, options=None):
        return _Asset.primitive("sphere", (radius,), options if options is not None else AssetOptions())

    def create_box(self, sim, width, height, depth, options=None):
        return _Asset.primitive("box", (width, height, depth), options if options is not None else AssetOptions())

    def create_capsule(self, sim, radius, length, options=None):
        return _Asset.primitive("capsule", (radius, length), options if options is not None else AssetOptions())

    def create_rigid_body_attractor(self, env, props):
        """ball_balance.py:285-300: recorded (for env 0; every env must ask for the same); prepare_sim turns them into the engine's pins"""
        if env.index == 0:
            env.sim.attractors.append(dict(body=int(props.rigid_handle), stiffness=float(props.stiffness), damping=float(props.damping), axes=int(props.axes),
                                           target=(props.target.p.x, props.target.p.y, props.target.p.z),
                                           offset=(props.offset.p.x, props.offset.p.y, props.offset.p.z)))
        return len(env.sim.attractors) - 1

    def get_rigid_transform(self, env, handle):
        """pose of a rigid body of the env (env-domain handle: find_actor_rigid_body_handle) in the env's frame -- franka_cabinet.py:262,308-310 asks for
        link poses while the envs are being created.  With the engine alive: the body's row of the rigid-body state tensor; before prepare_sim: the
        actor's start pose carried down the kinematic tree at zero joint positions."""
        from ...assets.model import mat_to_quat, quat_to_mat
        sim = env.sim
        k, b = 0, int(handle)
        while k < len(sim.slots) and b >= len(sim.slots[k]["asset"].body_names):
            b -= len(sim.slots[k]["asset"].body_names)
            k += 1
        if k >= len(sim.slots):
            raise IndexError(f"gym.get_rigid_transform: the env has no rigid body {handle}")
        a, e = sim.slots[k]["asset"], min(env.index, len(sim.slots[k]["poses"]) - 1)
        if sim.engine is not None:
            self.refresh_rigid_body_state_tensor(sim)
            row = sim.bufs["rb"].view(len(sim.envs), -1, 13)[env.index, int(handle)].cpu().numpy()
            return Transform(Vec3(*row[0:3]), Quat(*row[3:7]))
        ps = np.asarray(sim.slots[k]["poses"][e], float)
        R, p = quat_to_mat(ps[3:7]), ps[0:3].copy()
        if a.spec is not None:
            sp, dyn = a.spec, int(a.body_dyn[b])
            chain = []
            j = dyn
            while j > 0:
                chain.append(j)
                j = int(sp.parent[j])
            for j in reversed(chain):                  # zero joint positions: a body sits at its rest offset in its parent's frame
                p = p + R @ np.asarray(sp.bpos[j], float)
                R = R @ quat_to_mat(np.asarray(sp.bquat[j], float))
            p = p + R @ np.asarray(a.body_off_p[b], float)
            R = R @ quat_to_mat(np.asarray(a.body_off_q[b], float))
        return Transform(Vec3(*p), Quat(*mat_to_quat(R)))

    def debug_print_asset(self, asset):
        print(f"asset: bodies {asset.body_names}")

    def create_env(self, sim, lower, upper, num_per_row):
        env = _Env(sim, len(sim.envs))
        sim.envs.append(env)
        return env

    def create_actor(self, env, asset, pose, name="", group=-1, filter=-1, seg_id=0):
        sim = env.sim
        k = len(env.actors)
        if env.index == 0:
            if asset.spec is not None and any(sl["asset"].spec is not None for sl in sim.slots):
                raise NotImplementedError("the shim runs one articulated actor per env (plus free objects)")
            sim.slots.append(dict(asset=asset, name=name, filter=int(filter), group=int(group), poses=[], friction={}))
        elif k >= len(sim.slots) or sim.slots[k]["asset"] is not asset:
            raise NotImplementedError("every env must create the same actors in the same order")
        sl = sim.slots[k]
        sl["poses"].append([pose.p.x, pose.p.y, pose.p.z, pose.r.x, pose.r.y, pose.r.z, pose.r.w])
        if asset.shape_friction is not None:
            sl["friction"][env.index] = asset.shape_friction
        env.actors.append(name)
        return k

    def begin_aggregate(self, *a, **k):
        return True

    def end_aggregate(self, *a, **k):
        return True

    # ---- per-env actor properties (the reference's domain randomisation: vec_task.py:752-828 through utils/dr_utils.py:34-56)
    def _props(self, sim):
        if sim.props is None:
            sim.props = _ActorProps(sim)
        sim.props.grow(len(sim.envs))
        return sim.props

    @staticmethod
    def _touch(env):
        """a per-env property call on an env that is not the newest one: the envs are all there and somebody is walking them"""
        if env.index < len(env.sim.envs) - 1:
            env.sim.walked = True

    def _base_dof_props(self, sim, actor):
        """the actor's dof properties before any per-env change: what the task wrote while it created the actor, else the asset's"""
        sl = sim.slots[actor]
        if sl.get("dof_props") is not None:
            return sl["dof_props"]
        return self.get_asset_dof_properties(sl["asset"])

    def get_actor_dof_properties(self, env, actor):
        self._touch(env)
        sim = env.sim
        sl = sim.slots[actor]
        out = np.array(self._base_dof_props(sim, actor), copy=True)
        if sim.props is not None and actor == sim.robot and env.index < sim.props.n:
            pr, e = sim.props, env.index
            out["stiffness"] *= pr.stiff[e]; out["damping"] *= pr.damp[e]; out["armature"] *= pr.arm[e]
            out["lower"] += pr.lower[e]; out["upper"] += pr.upper[e]
        return out

    def set_actor_dof_properties(self, env, actor, props):
        """While the actor is being created (the env is the newest one, no engine yet): drive modes and gains are task parameters of the
        engine -- kept (env 0's; every env sets the same) and read by prepare_sim for the tasks whose drives the engine implements (Anymal,
        Quadcopter, Ingenuity, BallBalance, Articulation); the passive stiffness / damping of the other compiled models are fixed at build
        time.  Afterwards (domain randomisation, vec_task.py:783-828): stiffness / damping / armature become this env's factors of those
        values, lower / upper its shifts of the joint limits."""
        self._touch(env)
        sim = env.sim
        sl = sim.slots[actor]
        made = sl.setdefault("dof_props_made", set())
        # (a randomisation pass walks the envs from the first: by the time it reaches the newest env it has touched an older one)
        if sim.engine is None and not sim.walked and env.index == len(sim.envs) - 1 and env.index not in made:
            made.add(env.index)
            if env.index == 0:
                sl["dof_props"] = np.array(props, copy=True)
            elif "dof_props" in sl and not sl.get("dof_props_warned"):
                ref = sl["dof_props"]
                if any(not np.array_equal(np.asarray(props[k]), np.asarray(ref[k])) for k in ("driveMode", "stiffness", "damping")):
                    # (only env 0's creation-time properties reach the engine's task parameters -- say so when another env's differ)
                    import warnings
                    warnings.warn(f"gym.set_actor_dof_properties: env {env.index} sets other drive modes / gains than env 0 while the actor is being "
                                  "created; the engine's drives are task parameters shared by all envs (env 0's are used)")
                    sl["dof_props_warned"] = True
            return True
        if actor != sim.robot:
            return True
        base, pr, e = self._base_dof_props(sim, actor), self._props(sim), env.index
        pr.stiff[e], pr.damp[e], pr.arm[e] = _ratio(props["stiffness"], base["stiffness"]), _ratio(props["damping"], base["damping"]), \
            _ratio(props["armature"], base["armature"])
        pr.lower[e] = np.asarray(props["lower"], np.float64) - base["lower"]
        pr.upper[e] = np.asarray(props["upper"], np.float64) - base["upper"]
        pr.dirty = True
        return True

    def _base_masses(self, sim, actor):
        a = sim.slots[actor]["asset"]
        if a.spec is None and a.object_type == "box" and getattr(a, "dims", None) is not None and len(a.dims) == 3 and (getattr(sim, "scene", None) is not None or getattr(sim.asset, "generic", False)):
            # a box actor of a scene (create_box or a one-body URDF): the file's <inertial> mass, else density x volume (prepare_sim's rule)
            dens = float(getattr(a.options, "density", 1000.0) or 1000.0)
            return np.array([float(a.mass) if getattr(a, "mass", None) else dens * a.dims[0] * a.dims[1] * a.dims[2]])
        if a.spec is None:
            return np.array([_object_mass(a.object_type, sim.asset.task if sim.slots and any(sl["asset"].spec is not None for sl in sim.slots) else "ShadowHand")])
        return np.asarray([float(a.spec.mass[int(d)]) for d in a.body_dyn])

    def get_actor_rigid_body_properties(self, env, actor):
        self._touch(env)
        sim = env.sim
        a = sim.slots[actor]["asset"]
        m = self._base_masses(sim, actor)
        if sim.props is not None and env.index < sim.props.n:
            if a.spec is not None:
                m = m * sim.props.body[env.index][np.asarray(a.body_dyn, int)]
            elif actor == [k for k, sl in enumerate(sim.slots) if sl["asset"].spec is None][0]:
                m = m * sim.props.obj_mass[env.index]
        return [RigidBodyProperties(float(x)) for x in m]

    def set_actor_rigid_body_properties(self, env, actor, props, recompute_inertia=False):
        """rigid_body_properties.mass (vec_task.py:783-828 with dr_utils.py:63 recomputeInertia = True): the body's mass -- and with it its
        inertia -- times new / own.  Links welded to one engine body share its factor (the mean of theirs)."""
        self._touch(env)
        sim = env.sim
        a = sim.slots[actor]["asset"]
        pr, e = self._props(sim), env.index
        f = _ratio([float(p_.mass) for p_ in props], self._base_masses(sim, actor))
        if a.spec is None and getattr(sim, "scene", None) is not None:
            if abs(float(f[0]) - 1.0) > 1e-6 and not getattr(sim, "_warned_scene_mass", False):
                import warnings
                warnings.warn("set_actor_rigid_body_properties: the boxes of a scene have ONE mass for all envs (MiScene, include/mi_engine.h): per-env masses are not applied")
                sim._warned_scene_mass = True
        elif a.spec is None:
            ks = [k for k, sl in enumerate(sim.slots) if sl["asset"].spec is None]
            if actor == ks[0]:                 # the engine's object; the goal copy has no dynamics
                pr.obj_mass[e] = float(f[0])
        elif actor == sim.robot:
            dyn = np.asarray(a.body_dyn, int)
            pr.body[e] = [float(np.mean(f[dyn == b])) if np.any(dyn == b) else 1.0 for b in range(pr.nb)]
        pr.dirty = True
        return True

    def get_actor_rigid_shape_properties(self, env, actor):
        self._touch(env)
        sim = env.sim
        sl = sim.slots[actor]
        out = self.get_asset_rigid_shape_properties(sl["asset"]) or [RigidShapeProperties(1.0)]     # (a mesh-only asset lists no primitive shapes)
        mu = sl["friction"].get(env.index)
        if sim.props is not None and actor in sim.props.mu and env.index < sim.props.n and not np.isnan(sim.props.mu[actor][env.index]):
            mu = float(sim.props.mu[actor][env.index])
        if mu is not None:
            for p_ in out:
                p_.friction = mu
        return out

    def set_actor_rigid_shape_properties(self, env, actor, props):
        """rigid_shape_properties.friction: the engine has one coefficient per env and contact pair type, the first shape's (the reference
        draws one bucketed value for all shapes of an actor, vec_task.py:800-808); restitution has no counterpart (no bounce in the solver)"""
        self._touch(env)
        if not props:
            return True
        pr = self._props(env.sim)
        f = props[0].friction
        pr.friction(actor)[env.index] = float(np.ravel(np.asarray(f.cpu() if hasattr(f, "cpu") else f))[0])
        pr.dirty = True
        return True

    def get_actor_tendon_properties(self, env, actor):
        self._touch(env)
        sim = env.sim
        a = sim.slots[actor]["asset"]
        n = self.get_asset_tendon_count(a)
        tp = a.tendon_props or [(float(a.extras["tendon_limit_stiffness"]), float(a.extras["tendon_damping"]))] * n
        fk, fd = (sim.props.tendon_k[env.index], sim.props.tendon_d[env.index]) if sim.props is not None and env.index < sim.props.n else (1.0, 1.0)
        return [TendonProperties(stiffness=0.0, damping=tp[i][1] * fd, limit_stiffness=tp[i][0] * fk) for i in range(n)]

    def set_actor_tendon_properties(self, env, actor, props):
        """tendon_properties (ShadowHand.yaml:118-131): the coupling tendons' limit stiffness and damping as this env's factors of the asset's
        (a fixed tendon of the MJCF has no spring of its own: `stiffness` is 0 and scaling it changes nothing, in PhysX as here)"""
        self._touch(env)
        sim = env.sim
        a = sim.slots[actor]["asset"]
        n = self.get_asset_tendon_count(a)
        if n == 0:
            return True
        tp = a.tendon_props or [(float(a.extras["tendon_limit_stiffness"]), float(a.extras["tendon_damping"]))] * n
        live = [i for i in range(n) if tp[i] != (0.0, 0.0)]
        pr, e = self._props(sim), env.index
        if live:
            pr.tendon_k[e] = float(np.mean(_ratio([props[i].limit_stiffness for i in live], [tp[i][0] for i in live])))
            pr.tendon_d[e] = float(np.mean(_ratio([props[i].damping for i in live], [tp[i][1] for i in live])))
            pr.dirty = True
        return True

    def set_actor_scale(self, env, actor, scale):
        """vec_task.py:760-775: the free object of a hand task changes size (its mass stays); the articulated actor cannot"""
        self._touch(env)
        sim = env.sim
        if sim.slots[actor]["asset"].spec is not None:
            return False
        if actor != [k for k, sl in enumerate(sim.slots) if sl["asset"].spec is None][0]:
            return True                        # the goal copy is only drawn
        pr = self._props(sim)
        pr.obj_scale[env.index] = float(np.ravel(np.asarray(scale))[0])
        pr.dirty = True
        return True

    def _engine_object_slot(self, sim):
        live = [k for k, lives in self._object_slots(sim) if lives]
        return live[0] if live else -1

    def _flush_props(self, sim):
        """staged per-env properties -> the engine's tensors (csrc/core/engine.hpp AS_*: [nb] body factors, then [nd] damping, stiffness,
        armature; csrc/core/hand_engine.hpp HS_*: one factor per env for each of hand mass, dof damping, drive stiffness, tendon stiffness,
        tendon damping, object mass, object scale)"""
        pr = sim.props
        if pr is None or not pr.dirty or sim.engine is None:
            return
        pr.dirty = False
        t, n, dev = sim.engine.tensors, len(sim.envs), sim.device
        a = sim.asset
        put = lambda x: torch.as_tensor(np.asarray(x, np.float32), device=dev)
        sc = t.get("actor_scale")
        unsupported = []
        if sc is not None and sc.shape[1] == pr.nb + 3 * pr.nd and a.task not in _HAND_TASKS:
            sc[:] = put(np.concatenate([pr.body[:n], pr.damp[:n], pr.stiff[:n], pr.arm[:n]], axis=1))
            sim.engine.set_option("actor_tensors", 1)
        elif sc is not None and a.task in _HAND_TASKS:
            m = np.asarray(a.spec.mass, np.float64)
            driven = np.asarray(self._base_dof_props(sim, sim.robot)["stiffness"]) > 0
            damped = np.asarray(self._base_dof_props(sim, sim.robot)["damping"]) > 0
            bm = t.get("hand_body_mass_scale")
            if bm is not None and a.task == "ShadowHand":
                # per BODY, as the setter was given them (round 5); the kernels that read the tensor are switched in with the first factor != 1
                bm[:] = put(pr.body[:n])
                sc[:, 0] = 1.0
                if np.any(pr.body[:n] != 1.0):
                    sim.engine.set_option("hand_body_mass", 1)
            else:
                sc[:, 0] = put(pr.body[:n] @ m / m.sum())
            sc[:, 1] = put(pr.damp[:n][:, damped].mean(1) if damped.any() else np.ones(n))
            sc[:, 2] = put(pr.stiff[:n][:, driven].mean(1) if driven.any() else np.ones(n))
            if self.get_asset_tendon_count(a) > 0 and (a.tendon_props is None or any(tp_ != (0.0, 0.0) for tp_ in a.tendon_props)):
                k0 = [tp_ for tp_ in (a.tendon_props or []) if tp_ != (0.0, 0.0)]
                ls, dm = k0[0] if k0 else (float(a.extras["tendon_limit_stiffness"]), float(a.extras["tendon_damping"]))
                sc[:, 3] = put(pr.tendon_k[:n] * ls / float(a.extras["tendon_limit_stiffness"]))
                sc[:, 4] = put(pr.tendon_d[:n] * dm / float(a.extras["tendon_damping"]))
            sc[:, 5] = put(pr.obj_mass[:n]); sc[:, 6] = put(pr.obj_scale[:n])
        elif any(np.any(x[:n] != 1.0) for x in (pr.body, pr.damp, pr.stiff, pr.arm)):
            unsupported.append("mass / dof stiffness / damping / armature")
        sh = t.get("dof_limit_shift")
        if sh is not None and sh.shape[1] == 2 * pr.nd:
            sh[:] = put(np.concatenate([pr.lower[:n], pr.upper[:n]], axis=1))
            if a.task not in _HAND_TASKS:
                sim.engine.set_option("actor_tensors", 1)
        elif np.any(pr.lower[:n] != 0.0) or np.any(pr.upper[:n] != 0.0):
            unsupported.append("joint limits")
        if pr.mu and "friction" in t:
            # one contact coefficient per env: the articulated actor's shape friction against the ground, or -- a hand task -- the mean of the
            # hand's and the object's (the value the engine's own task classes use, tasks/base/vec_task.py); nan / -1: the model's own
            cols = []
            for k, sl in enumerate(sim.slots):
                if k == sim.robot or (a.task in _HAND_TASKS and k == self._engine_object_slot(sim)):
                    base = np.array([sl["friction"].get(e, np.nan) for e in range(n)])
                    cols.append(np.where(np.isnan(pr.mu[k][:n]), base, pr.mu[k][:n]) if k in pr.mu else base)
            mu = np.stack(cols, 1)
            some = ~np.isnan(mu).all(1)
            mu = np.where(np.isnan(mu), 1.0, mu).mean(1)
            t["friction"][:] = put(np.where(some, mu, -1.0))
        elif pr.mu:
            unsupported.append("shape friction")
        if unsupported and not getattr(sim, "_props_warned", False):
            import warnings
            sim._props_warned = True
            warnings.warn(f"per-env actor properties without a counterpart in the {a.task} kernels are ignored: " + ", ".join(unsupported))

    def get_actor_rigid_body_count(self, env, actor):
        return len(env.sim.slots[actor]["asset"].body_names)

    def get_actor_dof_count(self, env, actor):
        return self.get_asset_dof_count(env.sim.slots[actor]["asset"])

    def get_actor_count(self, env):
        return len(env.actors)

    def find_actor_handle(self, env, name):
        return env.actors.index(name)

    def get_actor_handle(self, env, index):      # dr_utils.py:233-236 (check_buckets) walks an env's actors by index
        return index

    def get_actor_name(self, env, actor):
        return env.actors[actor]

    def get_actor_rigid_shape_count(self, env, actor):
        return env.sim.slots[actor]["asset"].nshapes

    def get_actor_index(self, env, actor, domain=DOMAIN_SIM):
        # (a one-actor env: the env index, as before; called while the envs are being filled, so the per-env actor count is that of env 0)
        return env.index * max(len(env.sim.slots), len(env.actors)) + actor if domain == DOMAIN_SIM else actor

    def find_actor_rigid_body_handle(self, env, actor, name):
        # env domain: the bodies of the env's actors in creation order (franka_cube_stack.py:364-371 indexes the [N, bodies, 13] view of the
        # rigid-body state tensor with it -- the arm's links first, then one body per box actor)
        sl = env.sim.slots
        return sum(len(sl[k]["asset"].body_names) for k in range(actor)) + sl[actor]["asset"].body_names.index(name)

    def get_actor_joint_dict(self, env, actor):
        """joint name -> joint index; gym numbers an actor's joints like its links: joint k is the one link k + 1 hangs on, fixed joints included
        while collapse_fixed_joints is off (franka_cube_stack.py:390 looks up 'panda_hand_joint' to pick the hand link's rows of the Jacobian)"""
        a = env.sim.slots[actor]["asset"]
        out = {}
        path = getattr(a, "path", None)
        if path and path.endswith(".urdf") and os.path.isfile(path):
            import xml.etree.ElementTree as ET
            for j in ET.parse(path).getroot().findall("joint"):
                child = j.find("child").get("link")
                if child in a.body_names:
                    out[j.get("name")] = a.body_names.index(child) - 1
        elif a.spec is not None:
            for d, nm in enumerate(a.spec.dof_names):      # MJCF: the movable joints; a body with several hinges lists each of them
                out[nm] = max(a.body_names.index(a.spec.body_names[int(a.spec.dof_body[d])]) - 1, 0)
        return out

    def get_actor_rigid_body_dict(self, env, actor):
        return {nm: i for i, nm in enumerate(env.sim.slots[actor]["asset"].body_names)}

    def get_actor_dof_dict(self, env, actor):
        a = env.sim.slots[actor]["asset"]
        return {nm: i for i, nm in enumerate(a.spec.dof_names)} if a.spec is not None else {}

    def find_actor_dof_handle(self, env, actor, name):
        return list(env.sim.slots[actor]["asset"].spec.dof_names).index(name)

    def set_rigid_body_color(self, *a, **k):
        pass

    def get_env_origin(self, env):
        return Vec3(0.0, 0.0, 0.0)

    def get_sim_params(self, sim):
        return sim.params

    def set_sim_params(self, sim, params):
        sim.params = params
        if sim.engine is not None:
            for i, k in enumerate(("gravity_x", "gravity_y", "gravity_z")):
                sim.engine.set_option(k, list(params.gravity)[i])

    def get_frame_count(self, sim):
        return sim.frame

    def get_sim_dof_count(self, sim):
        return len(sim.envs) * sim.asset.spec.nd

    def get_sim_actor_count(self, sim):
        return len(sim.envs) * sim.nactors

    # ------------------------------------------------------------------ the engine comes to life
    def prepare_sim(self, sim):
        from ... import native
        from ...utils.config import compose
        if not sim.slots or not sim.envs:
            raise RuntimeError("gym.prepare_sim: no actor was created")
        if sim.params.up_axis != UP_AXIS_Z:
            raise ValueError("only up_axis 'z' is implemented")
        n = len(sim.envs)
        for sl in sim.slots:
            if len(sl["poses"]) != n:
                raise NotImplementedError("every env must create the same actors")
        asset, p, px = sim.asset, native.MiSimParams(), sim.params.physx
        rslot = sim.slots[sim.robot]
        p.dt, p.substeps = float(sim.params.dt), int(sim.params.substeps)
        for i, g in enumerate(sim.params.gravity):
            p.gravity[i] = float(g)
        p.iters = int(px.num_position_iterations) + int(px.num_velocity_iterations)
        p.contact_offset, p.rest_offset = float(px.contact_offset), float(px.rest_offset)
        p.max_depen_vel = float(px.max_depenetration_velocity)
        p.erp, p.cfm, p.warm, p.ground_z = 0.5, 1e-6, 1.0, 0.0
        p.plane_mu = float(sim.plane.static_friction) if sim.plane is not None else 1.0
        cfg = None if asset.generic else compose(overrides=[f"task={asset.task}"])["task"]        # the fused kernels' own parameters: unused by simulate()
        poses = torch.tensor(rslot["poses"], dtype=torch.float32)
        terrain = None
        lib_path = None
        if asset.generic:
            from ...assets import runtime
            spec, nd = asset.spec, asset.spec.nd
            dp = rslot.get("dof_props")
            if dp is None:
                dp = _dof_properties(spec, _drive_mode(asset))
            modes = [int(m) for m in dp["driveMode"]]
            if any(m not in (DOF_MODE_NONE, DOF_MODE_EFFORT, DOF_MODE_POS) for m in modes):
                raise NotImplementedError("Articulation: dofs are position drives (DOF_MODE_POS) or effort / undriven; no velocity drives")
            import copy
            sp = copy.deepcopy(spec)            # the actor as its dof properties now describe it
            sp.dof_stiffness, sp.dof_damping = np.array(dp["stiffness"], float), np.array(dp["damping"], float)
            sp.dof_armature = np.array(dp["armature"], float)
            sp.dof_lower, sp.dof_upper = np.array(dp["lower"], float), np.array(dp["upper"], float)
            sp, kp, kd = runtime.drive_split(sp, [d for d in range(nd) if modes[d] == DOF_MODE_POS])
            sens = [int(asset.body_dyn[b]) for b in asset.sensors] or [0]
            tp = native.MiArticulationParams()
            for d in range(nd):
                tp.kp[d], tp.kd[d] = float(kp[d]), float(kd[d])
            tp.max_angular_velocity = float(getattr(asset.options, "max_angular_velocity", 0.0) or 0.0)
            for k in range(7):
                tp.init_root[k] = float(poses[0, k])
            for d in range(nd):          # the asset's joint velocity limits (scenes; include/mi_engine.h drive_vmax)
                vm = float(dp["velocity"][d]) if "velocity" in dp.dtype.names else 0.0
                tp.drive_vmax[d] = vm if 0.0 < vm < 1e6 else 0.0
            boxes = [(k, sl) for k, sl in enumerate(sim.slots) if sl["asset"].spec is None]
            sim.scene = None
            if boxes and spec.fixed_base:
                # the other actors of the env (franka_cube_stack.py:212-233,330-339: gym.create_box assets): the engine's scene of free and
                # static boxes beside the fixed-base actor (include/mi_engine.h MiScene, csrc/core/scene_engine.hpp)
                sc, sim.scene = tp.scene, {}
                sc.arm_gravity = 0 if getattr(asset.options, "disable_gravity", False) else 1
                mu_arm = rslot["friction"].get(0) if rslot["friction"] else None
                sc.arm_mu = float(mu_arm if mu_arm is not None else (np.mean(spec.sph_friction) if len(spec.sph_friction) else 1.0))
                from ...assets.model import quat_mul, quat_to_mat
                for k, sl in boxes:
                    a = sl["asset"]
                    if a.object_type != "box":
                        raise NotImplementedError(f"scene actors are gym.create_box assets (got a {a.object_type})")
                    g_r, g_k = rslot.get("group", -1), sl.get("group", -1)
                    if getattr(a, "hollow", False):
                        continue       # a ring wall without a collision shape (load_asset warned): lives in the stand-in
                    if g_r != -1 and g_k != -1 and g_r != g_k:
                        continue       # another collision group than the robot's in its own env (trifinger.py:561-563: the goal marker): touches nothing, lives in the stand-in
                    ps = np.asarray(sl["poses"], float).copy()
                    fixed = bool(getattr(a.options, "fix_base_link", False))
                    off_p, off_q = np.asarray(getattr(a, "box_pos", (0.0, 0.0, 0.0)), float), np.asarray(getattr(a, "box_quat", (0.0, 0.0, 0.0, 1.0)), float)
                    if np.abs(off_p).max() > 1e-9 or abs(abs(off_q[3]) - 1.0) > 1e-9:
                        if not fixed:
                            raise NotImplementedError("a free box of the scene is centred on its link frame (the file's collision origin is not)")
                        for e_ in range(len(ps)):       # the static box sits at the actor's pose carried to the collision geometry's frame
                            ps[e_, 0:3] = ps[e_, 0:3] + quat_to_mat(ps[e_, 3:7]) @ off_p
                            ps[e_, 3:7] = quat_mul(ps[e_, 3:7], off_q)
                    if fixed and np.abs(ps - ps[0]).max() > 1e-9:
                        raise NotImplementedError("a static box of the scene stands at the same env-local pose in every env")
                    half = [0.5 * d for d in a.dims]
                    mu = float(sl["friction"].get(0, 1.0)) if sl["friction"] else 1.0
                    if fixed:
                        j = sc.n_static
                        if j >= native.MI_SCENE_MAX_STATIC:
                            raise NotImplementedError(f"a scene holds at most {native.MI_SCENE_MAX_STATIC} static boxes")
                        for c in range(3):
                            sc.static_pos[j][c], sc.static_half[j][c] = float(ps[0, c]), float(half[c])
                        for c in range(4):
                            sc.static_quat[j][c] = float(ps[0, 3 + c])
                        sc.static_mu[j] = mu
                        sim.scene[k] = ("static", j)
                        sc.n_static = j + 1
                    else:
                        j = sc.n_free
                        if j >= native.MI_SCENE_MAX_FREE:
                            raise NotImplementedError(f"a scene holds at most {native.MI_SCENE_MAX_FREE} free boxes")
                        dens = float(getattr(a.options, "density", 1000.0) or 1000.0)
                        m = float(a.mass) if getattr(a, "mass", None) else dens * a.dims[0] * a.dims[1] * a.dims[2]      # (a URDF's <inertial> wins over the density)
                        sc.free_mass[j], sc.free_mu[j] = m, mu
                        for c in range(3):
                            o1, o2 = a.dims[(c + 1) % 3], a.dims[(c + 2) % 3]
                            ine = a.inertia[c] if getattr(a, "inertia", None) else m * (o1 * o1 + o2 * o2) / 12.0
                            sc.free_half[j][c], sc.free_inertia[j][c] = float(half[c]), float(ine)
                        for c in range(7):
                            sc.free_init[j][c] = float(ps[0, c])
                        sim.scene[k] = ("free", j)
                        sc.n_free = j + 1
            elif getattr(asset.options, "disable_gravity", False):      # franka_cube_stack.py:186: the only articulated actor of the env does not feel gravity
                for i in range(3):
                    p.gravity[i] = 0.0
            lib_path = runtime.variant_library(asset.model_name, sp, sim.device, sensors=sens)
            asset.engine_spec = sp
        elif asset.task == "Cartpole":
            from ...tasks.cartpole import cartpole_params_from_cfg
            tp = cartpole_params_from_cfg(cfg)
        elif asset.task == "AnymalTerrain":
            from ...tasks.anymal_terrain import _FlatTerrain, anymal_params_from_cfg
            tp = anymal_params_from_cfg(cfg, list(asset.spec.dof_names))
            terrain = sim.terrain
            if terrain is None:
                terrain = _FlatTerrain()
                terrain.max_init_level = 0
        elif asset.task in _HAND_TASKS:
            objs = [sl["asset"].object_type for sl in sim.slots if sl["asset"].spec is None]
            cfg["env"]["objectType"] = objs[0] if objs else "block"
            if asset.task == "ShadowHand":
                from ...tasks.shadow_hand import hand_params_from_cfg
                tp = hand_params_from_cfg(cfg)
            else:
                from ...tasks.allegro_hand import allegro_params_from_cfg
                tp = allegro_params_from_cfg(cfg)
                for k in range(4):       # the actor's orientation is a task parameter of the engine (allegro_hand.py:283)
                    if abs(abs(tp.hand_quat[k]) - abs(float(poses[0, 3 + k]))) > 1e-5:
                        raise NotImplementedError("the Allegro hand is created with the start rotation of allegro_hand.py:283")
            for k in range(3):
                if abs(tp.hand_pos[k] - float(poses[0, k])) > 1e-6:
                    raise NotImplementedError("the hand is mounted at (0, 0, 0.5) (shadow_hand.py:306-307, allegro_hand.py:282)")
        elif asset.task == "Anymal":
            # anymal.py:186-206: DOF_MODE_POS drives with env.control.stiffness / damping on every dof
            from ...tasks.anymal import anymal_flat_params_from_cfg
            tp = anymal_flat_params_from_cfg(cfg, list(asset.spec.dof_names))
            dp = rslot.get("dof_props")
            if dp is not None:
                if not (np.all(dp["driveMode"] == DOF_MODE_POS) and np.ptp(dp["stiffness"]) == 0 and np.ptp(dp["damping"]) == 0):
                    raise NotImplementedError("Anymal: one position drive gain pair for all dofs (anymal.py:203-206)")
                tp.kp, tp.kd = float(dp["stiffness"][0]), float(dp["damping"][0])
            for k in range(7):
                tp.base_init_state[k] = float(poses[0, k])
        elif asset.task == "Quadcopter":
            from ...tasks.quadcopter import quadcopter_params_from_cfg
            tp = quadcopter_params_from_cfg(cfg, asset.spec)
            dp = rslot.get("dof_props")
            if dp is not None:             # quadcopter.py:236-239: position drives, stiffness 1000, damping 0
                if np.ptp(dp["stiffness"]) != 0 or np.ptp(dp["damping"]) != 0:
                    raise NotImplementedError("Quadcopter: one drive gain pair for all dofs")
                tp.drive_stiffness, tp.drive_damping = float(dp["stiffness"][0]), float(dp["damping"][0])
            tp.max_angular_velocity = float(getattr(asset.options, "max_angular_velocity", tp.max_angular_velocity))
            tp.init_height = float(poses[0, 2])
        elif asset.task == "Ingenuity":
            from ...tasks.ingenuity import ingenuity_params_from_cfg
            tp = ingenuity_params_from_cfg(cfg)
            dp = rslot.get("dof_props")
            if dp is not None and (np.any(dp["stiffness"] != 0) or np.any(dp["damping"] != 0)):
                raise NotImplementedError("Ingenuity: the rotor joints are undriven (ingenuity.py:265-268)")
            tp.max_angular_velocity = float(getattr(asset.options, "max_angular_velocity", tp.max_angular_velocity))
            tp.init_height = float(poses[0, 2])
        elif asset.task == "BallBalance":
            from ...tasks.ball_balance import ball_balance_params_from_cfg
            tp = ball_balance_params_from_cfg(cfg, asset.spec)
            dp = rslot.get("dof_props")
            if dp is not None:             # ball_balance.py:271-281: the lower-leg joints are position drives, the upper ones free
                act = [d for d in range(asset.spec.nd) if int(dp["driveMode"][d]) == DOF_MODE_POS]
                if not act or np.ptp(dp["stiffness"][act]) != 0 or np.ptp(dp["damping"][act]) != 0:
                    raise NotImplementedError("BallBalance: position drives with one gain pair")
                tp.drive_kp, tp.drive_kd = float(dp["stiffness"][act[0]]), float(dp["damping"][act[0]])
                tp.actuated_mask = sum(1 << d for d in act)
            if sim.attractors:             # :285-300: three translation attractors on the lower legs = the engine's pins
                legs = [asset.body_names.index(f"lower_leg{j}") for j in range(3)]
                if [a["body"] for a in sim.attractors] != legs or any(a["axes"] != AXIS_TRANSLATION for a in sim.attractors):
                    raise NotImplementedError("BallBalance: one translation attractor per lower leg (ball_balance.py:285-300)")
                a0 = sim.attractors[0]
                if any(a["stiffness"] != a0["stiffness"] or a["damping"] != a0["damping"] or a["offset"] != a0["offset"] for a in sim.attractors):
                    raise NotImplementedError("BallBalance: the three attractors share stiffness, damping and body offset")
                tp.pin_stiffness, tp.pin_damping = a0["stiffness"], a0["damping"]
                for k in range(3):
                    tp.pin_offset[k] = a0["offset"][k]
                    for j in range(3):
                        tp.pin_target[j][k] = sim.attractors[j]["target"][k]
            else:
                raise NotImplementedError("BallBalance without attractors: the engine's tray stands on pinned feet")
            balls = [sl for sl in sim.slots if sl["asset"].spec is None]
            if balls:                      # :274-277, 303-306: the ball's radius / density and start pose
                ba = balls[0]["asset"]
                if ba.object_type != "sphere":
                    raise NotImplementedError("BallBalance: the free object is a sphere")
                r, dens = ba.dims[0], float(getattr(ba.options, "density", 1000.0))
                tp.ball_radius, tp.ball_mass = r, dens * 4.0 / 3.0 * np.pi * r ** 3
                tp.ball_inertia = 0.4 * tp.ball_mass * r * r
                for k in range(3):
                    tp.ball_init_pos[k] = float(balls[0]["poses"][0][k])
            tp.tray_height = float(poses[0, 2])
            for j, sp_ in enumerate(getattr(asset, "sensor_poses", [])[:3]):
                tp.sensor_pos[j][0], tp.sensor_pos[j][1], tp.sensor_pos[j][2] = sp_.p.x, sp_.p.y, sp_.p.z
        else:
            from ...tasks.locomotion import loco_params_from_cfg
            tp = loco_params_from_cfg(cfg, asset.model_name, float(poses[0, 2]))
        own_sensors = None
        if asset.task in _HAND_TASKS and asset.sensors and not asset.engine_sensor_bodies:
            # force sensors on a hand whose compiled model has none (the Allegro hand of allegro_hand.py observes no fingertip forces; the
            # dextreme task creates one per fingertip, tasks/dextreme/allegro_hand_dextreme.py:264-269): a variant of the model that carries them
            own_sensors = [int(asset.body_dyn[b]) for b in asset.sensors]
        if asset.variant or own_sensors is not None:      # compiled once per distinct model, cached (isaacgymenvs_amd/_variants/<hash>/)
            from ...assets import runtime
            lib_path = runtime.variant_library(asset.model_name, asset.spec, sim.device, sensors=own_sensors)
        sim.engine = native.Engine(asset.task, p, tp, n, sim.device, terrain=terrain, lib_path=lib_path)
        if sim.device != "cpu":
            native.select_multi_wave(sim.engine, asset.task, n)        # the launch shape (and with it the solver order) make() would pick
        if asset.has_self_collision:
            sim.engine.set_option("self_collision", 1 if rslot["filter"] == 0 else 0)      # create_actor(..., filter): 0 = links collide
        t = sim.engine.tensors
        root = torch.zeros((n, 13), dtype=torch.float32)
        root[:, :7] = poses
        root = root.to(sim.device)
        if asset.spec.fixed_base and not asset.generic:
            root[:, :7] = t["root_states"][:, :7]        # a fixed base stays where the engine mounts it (the rail of the cart-pole, the hand's mount)
        # (a fixed-base Articulation robot stands where each env's create_actor put it: franka_cube_stack.py:314-323 may draw a start pose per env)
        t["root_states"][:] = root
        if "initial_root_states" in t:
            t["initial_root_states"][:] = root
        t["dof_state"].zero_()
        if rslot["friction"] and "friction" in t:          # per-env shape friction (anymal_terrain.py:236-239,279-281)
            mu = torch.full((n,), -1.0)
            for e, val in rslot["friction"].items():
                mu[e] = val
            t["friction"][:] = mu.to(sim.device)
        if getattr(sim, "scene", None):
            for k, (kind, j) in sim.scene.items():
                if kind == "free":
                    o = torch.zeros((n, 13), dtype=torch.float32)
                    o[:, :7] = torch.tensor(sim.slots[k]["poses"], dtype=torch.float32)
                    t["scene_state"][:, j] = o.to(sim.device)
        if self._object_tensor(sim) is not None:
            k_obj = [k for k, sl in enumerate(sim.slots) if sl["asset"].spec is None]
            if k_obj:
                o = torch.zeros((n, 13), dtype=torch.float32)
                o[:, :7] = torch.tensor(sim.slots[k_obj[0]]["poses"], dtype=torch.float32)
                t[self._object_tensor(sim)][:] = o.to(sim.device)
        if asset.task in _HAND_TASKS:
            if asset.tendon_props:                   # limit stiffness / damping of the four coupling tendons as factors of the model's own
                ks = {pr for i, pr in enumerate(asset.tendon_props) if pr != (0.0, 0.0)}
                if len(ks) > 1:
                    raise NotImplementedError("one limit_stiffness / damping for all coupling tendons")
                if ks:
                    ls, dm = next(iter(ks))
                    t["actor_scale"][:, 3] = ls / float(asset.extras["tendon_limit_stiffness"])
                    t["actor_scale"][:, 4] = dm / float(asset.extras["tendon_damping"])
        # (BallBalance: the task's three sensors sit on the tray, :254-260 -- the engine computes exactly those from the tray's momentum balance;
        #  its `sensor` bodies are the lower legs the attractors hold)
        if asset.task not in ("BallBalance", "Articulation") and own_sensors is None and asset.sensors and \
                asset.sensors != asset.engine_sensor_bodies[:len(asset.sensors)]:
            raise NotImplementedError(f"force sensors on bodies {asset.sensors}: the compiled {asset.model_name} model has them on "
                                      f"{asset.engine_sensor_bodies}")
        self._flush_props(sim)                   # what a setup-time randomisation wrote before the engine existed (shadow_hand.py:224-226)
        # tensors acquired while the envs were being created (franka_cube_stack.py:356 acquires inside create_sim): the same buffers, filled now
        for name, fn in (("root", self.refresh_actor_root_state_tensor), ("dof", self.refresh_dof_state_tensor), ("rb", self.refresh_rigid_body_state_tensor),
                         ("jacobian", self.refresh_jacobian_tensors), ("mass_matrix", self.refresh_mass_matrix_tensors)):
            if name in sim.bufs:
                fn(sim)
        return True

    # ------------------------------------------------------------------ tensor API
    def _buf(self, sim, name, shape):
        if name not in sim.bufs:
            sim.bufs[name] = torch.zeros(shape, dtype=torch.float32, device=sim.device)
        return sim.bufs[name]

    def _object_slots(self, sim):
        """[(slot, lives in the engine)] of the free objects: the first one is the engine's object, the others (goal) live in the shim.  In a
        scene (sim.scene: slot -> ("free" | "static", index), prepare_sim) every free box lives in the engine."""
        ks = [k for k, sl in enumerate(sim.slots) if sl["asset"].spec is None]
        if getattr(sim, "scene", None):
            return [(k, sim.scene.get(k, ("", 0))[0] == "free") for k in ks]
        return [(k, i == 0 and self._object_tensor(sim) is not None) for i, k in enumerate(ks)]

    def _object_view(self, sim, k):
        """the engine's [N, 13] root state of the free object in slot k"""
        if getattr(sim, "scene", None):
            return sim.engine.tensors["scene_state"][:, sim.scene[k][1]]
        return sim.engine.tensors[self._object_tensor(sim)]

    _OBJECT_TENSOR = {"ShadowHand": "object_state", "AllegroHand": "object_state", "BallBalance": "ball_states", "Ingenuity": "marker_states"}

    def _object_tensor(self, sim):
        """the engine tensor that holds the root state of the task's free object: the manipulated object of the hand tasks, BallBalance's ball
        (ball_balance.py:303-306), Ingenuity's target marker (ingenuity.py:270; no physics, the engine only keeps its state)"""
        return self._OBJECT_TENSOR.get(sim.asset.task)

    def acquire_actor_root_state_tensor(self, sim):
        n, A = len(sim.envs), sim.nactors
        buf = self._buf(sim, "root", (n * A, 13))
        for k, sl in enumerate(sim.slots):          # start poses of everything (the goal object keeps what the task writes afterwards)
            buf.view(n, A, 13)[:, k, :7] = torch.tensor(sl["poses"], dtype=torch.float32, device=sim.device)
        self.refresh_actor_root_state_tensor(sim)
        return sim.bufs["root"]

    def acquire_dof_state_tensor(self, sim):
        self.refresh_dof_state_tensor(sim)
        return sim.bufs["dof"]

    def acquire_force_sensor_tensor(self, sim):
        self.refresh_force_sensor_tensor(sim)
        return sim.bufs["sensor"]

    def acquire_dof_force_tensor(self, sim):
        self.refresh_dof_force_tensor(sim)
        return sim.bufs["dof_force"]

    def acquire_net_contact_force_tensor(self, sim):
        self.refresh_net_contact_force_tensor(sim)
        return sim.bufs["netf"]

    def acquire_rigid_body_state_tensor(self, sim):
        self.refresh_rigid_body_state_tensor(sim)
        return sim.bufs["rb"]

    # Jacobians / mass matrices of the articulated actor (franka_cube_stack.py:388-392: acquire once, refresh every step, :551-552).  The
    # simulator's layout: [num_envs, links, 6, dofs] with the base link left out and no base columns for a fixed-base actor, all links and
    # 6 leading base columns for a floating one; mass matrix [num_envs, dofs (+ 6), dofs (+ 6)].
    def acquire_jacobian_tensor(self, sim, actor_name=None):
        self.refresh_jacobian_tensors(sim)
        return sim.bufs["jacobian"]

    def refresh_jacobian_tensors(self, sim):
        if sim.engine is None:          # acquired while the envs are being created (franka_cube_stack.py:356,388): filled by prepare_sim
            a = sim.asset
            nl, nv = len(a.body_names) - (1 if a.spec.fixed_base else 0), a.spec.nv
            self._buf(sim, "jacobian", (len(sim.envs), nl, 6, nv))
            return True
        full = sim.engine.compute_jacobians()
        a = sim.asset
        if len(a.body_names) != a.spec.nb or not np.array_equal(np.asarray(a.body_dyn), np.arange(a.spec.nb)):
            # collapse_fixed_joints off: gym lists every link of the file; a welded link moves with its engine body, its origin at a fixed offset
            # r from that body's: v_link = v + w x r  =>  the linear rows pick up  (angular rows) x r
            dyn = torch.as_tensor(np.asarray(a.body_dyn), device=sim.device)
            sim.engine.refresh_rigid_body_states()
            q = sim.engine.tensors["rigid_body_state"][:, dyn, 3:7]
            off_p = torch.tensor(a.body_off_p, dtype=torch.float32, device=sim.device)
            r = _quat_rotate(q, off_p.expand(q.shape[0], len(dyn), 3))                 # [n, links, 3]
            full = full[:, dyn].clone()                                                   # [n, links, 6, nv]
            ang = full[:, :, 3:6].transpose(-1, -2)                                       # [n, links, nv, 3]
            full[:, :, 0:3] += torch.cross(ang, r.unsqueeze(-2).expand_as(ang), dim=-1).transpose(-1, -2)
        view = full[:, 1:] if sim.asset.spec.fixed_base else full
        buf = sim.bufs.get("jacobian")
        if buf is None:
            sim.bufs["jacobian"] = view.contiguous()
        else:
            buf.copy_(view)
        return True

    def acquire_mass_matrix_tensor(self, sim, actor_name=None):
        self.refresh_mass_matrix_tensors(sim)
        return sim.bufs["mass_matrix"]

    def refresh_mass_matrix_tensors(self, sim):
        if sim.engine is None:
            self._buf(sim, "mass_matrix", (len(sim.envs), sim.asset.spec.nv, sim.asset.spec.nv))
            return True
        buf = sim.bufs.get("mass_matrix")
        if buf is None:
            sim.bufs["mass_matrix"] = sim.engine.compute_mass_matrices()
        else:
            sim.engine.compute_mass_matrices(buf)
        return True

    def refresh_actor_root_state_tensor(self, sim):
        n, A = len(sim.envs), sim.nactors
        v = self._buf(sim, "root", (n * A, 13)).view(n, A, 13)
        if sim.engine is None:
            return True
        v[:, sim.robot].copy_(sim.engine.tensors["root_states"])
        for k, phys in self._object_slots(sim):
            if phys:
                v[:, k].copy_(self._object_view(sim, k))
        return True

    def refresh_dof_state_tensor(self, sim):
        n, nd = len(sim.envs), sim.asset.spec.nd
        buf = self._buf(sim, "dof", (n * nd, 2))
        if sim.engine is not None:
            buf.view(n, nd, 2).copy_(sim.engine.tensors["dof_state"])
        return True

    def refresh_force_sensor_tensor(self, sim):
        n, ns = len(sim.envs), max(len(sim.asset.sensors), 1)
        fs = sim.engine.tensors["force_sensor"]
        self._buf(sim, "sensor", (n * ns, 6)).view(n, ns, 6).copy_(fs[:, :ns])
        return True

    def refresh_dof_force_tensor(self, sim):
        n, nd = len(sim.envs), sim.asset.spec.nd
        self._buf(sim, "dof_force", (n * nd,)).view(n, nd).copy_(sim.engine.tensors["dof_force"])
        return True

    def refresh_net_contact_force_tensor(self, sim):
        """per rigid body, world frame, last sub-step (anymal_terrain.py:119,130): bodies welded to an engine body report that body's force on
        the engine body's own row and zero elsewhere"""
        if sim.engine is None or "net_contact_force" not in sim.engine.tensors:
            return True
        a = sim.asset
        n, nb = len(sim.envs), len(a.body_names)
        buf = self._buf(sim, "netf", (n * nb, 3)).view(n, nb, 3)
        src = sim.engine.tensors["net_contact_force"]
        own = [i for i, nm in enumerate(a.body_names) if nm in a.spec.body_names]
        buf.zero_()
        buf[:, own] = src[:, [a.spec.body_names.index(a.body_names[i]) for i in own]]
        return True

    def refresh_rigid_body_state_tensor(self, sim):
        """[N * (bodies of the articulation + one per free object), 13] (shadow_hand.py:150-175): the engine's rigid_body_state tensor, the
        file's welded links riding on their engine body at their fixed offset, then the objects' root states"""
        a = sim.asset
        n, nb = len(sim.envs), len(a.body_names)
        if sim.engine is None:      # acquired before prepare_sim: one row per body of the arm and per other actor; filled by prepare_sim
            self._buf(sim, "rb", (n * (nb + len([1 for sl in sim.slots if sl["asset"].spec is None])), 13))
            return True
        objs = self._object_slots(sim)
        buf = self._buf(sim, "rb", (n * (nb + len(objs)), 13)).view(n, nb + len(objs), 13)
        sim.engine.refresh_rigid_body_states()
        src = sim.engine.tensors["rigid_body_state"][:, torch.as_tensor(a.body_dyn, device=sim.device)]       # [n, nb, 13]
        off_p = torch.tensor(a.body_off_p, dtype=torch.float32, device=sim.device)
        off_q = torch.tensor(a.body_off_q, dtype=torch.float32, device=sim.device)
        if float(off_p.abs().max()) == 0.0 and float((off_q - off_q.new_tensor([0, 0, 0, 1])).abs().max()) == 0.0:
            buf[:, :nb].copy_(src)
        else:
            q = src[..., 3:7]
            r = _quat_rotate(q, off_p.expand(n, nb, 3))
            buf[:, :nb, 0:3] = src[..., 0:3] + r
            buf[:, :nb, 3:7] = _quat_mul(q, off_q.expand(n, nb, 4))
            buf[:, :nb, 7:10] = src[..., 7:10] + torch.cross(src[..., 10:13], r, dim=-1)
            buf[:, :nb, 10:13] = src[..., 10:13]
        root = sim.bufs.get("root")
        for i, (k, phys) in enumerate(objs):
            if phys:
                buf[:, nb + i].copy_(self._object_view(sim, k))
            elif root is not None:
                buf[:, nb + i].copy_(root.view(n, sim.nactors, 13)[:, k])
        return True

    def enable_actor_dof_force_sensors(self, env, actor):
        return True

    def set_dof_actuation_force_tensor(self, sim, forces):
        n, nd = len(sim.envs), sim.asset.spec.nd
        sim.engine.tensors["dof_actuation_force"].copy_(forces.view(n, nd))
        return True

    def _write_targets(self, sim, targets, envs=None):
        """position targets of the drives -> the engine tensor the task's simulate() reads: `cur_targets` (hands), `dof_position_targets`
        (Quadcopter, BallBalance); the flat Anymal's engine keeps the policy's actions and derives target = action_scale * a + default
        (anymal.py:226-229), so the targets are turned back into those"""
        t = sim.engine.tensors
        n, nd = len(sim.envs), sim.asset.spec.nd
        x = targets.view(n, nd)
        if sim.asset.task == "Anymal":
            tp = sim.engine._tp
            dflt = torch.tensor([tp.default_dof_pos[d] for d in range(nd)], dtype=torch.float32, device=sim.device)
            name, x = "actions", (x - dflt) / float(tp.action_scale)
        else:
            name = "cur_targets" if "cur_targets" in t else "dof_position_targets"
        if envs is None:
            t[name].copy_(x)
        else:
            t[name][envs] = x[envs]
        return True

    def set_dof_position_target_tensor(self, sim, targets):
        return self._write_targets(sim, targets)

    def set_dof_position_target_tensor_indexed(self, sim, targets, actor_indices, count):
        return self._write_targets(sim, targets, torch.div(actor_indices[:count].long(), sim.nactors, rounding_mode="floor"))

    def set_dof_actuation_force_tensor_indexed(self, sim, forces, actor_indices, count):
        n, nd = len(sim.envs), sim.asset.spec.nd
        envs = self._envs_of(sim, actor_indices, count, sim.robot)
        sim.engine.tensors["dof_actuation_force"][envs] = forces.view(n, nd)[envs]
        return True

    def apply_rigid_body_force_tensors(self, sim, forces=None, torques=None, space=ENV_SPACE):
        """shadow_hand.py:700-708: random forces on the object's body, in its local frame; held for the next simulate() only"""
        t = sim.engine.tensors
        if forces is not None and "forces" in t and "object_force" not in t:
            # quadcopter.py:290-292 / ingenuity.py:350-352: thrust forces on the rotor bodies, in the bodies' own frames -- the engine's `forces`
            # tensor [N, bodies, 3] (it applies the rows of its rotor bodies; the tasks leave the others zero); held for the next simulate()
            if space != LOCAL_SPACE or torques is not None:
                raise NotImplementedError("apply_rigid_body_force_tensors: forces in LOCAL_SPACE, no torques (quadcopter.py:292, ingenuity.py:352)")
            n = len(sim.envs)
            f = forces.view(n, -1, 3)
            nbe = min(f.shape[1], t["forces"].shape[1])
            t["forces"][:, :nbe].copy_(f[:, :nbe])
            sim.one_shot_force = True
            return True
        objs = [k for k, phys in self._object_slots(sim) if phys]
        if forces is None or not objs or "object_force" not in sim.engine.tensors:
            raise NotImplementedError("apply_rigid_body_force_tensors: forces on the free object of the hand tasks or on the rotor bodies of Quadcopter / Ingenuity")
        n, nb = len(sim.envs), len(sim.asset.body_names)
        f = forces.view(n, -1, 3)[:, nb]
        if space == LOCAL_SPACE:
            f = _quat_rotate(sim.engine.tensors["object_state"][:, 3:7], f)
        sim.engine.tensors["object_force"].copy_(f)
        sim.one_shot_force = True
        return True

    def _clear_warm_start(self, sim, ids):
        t = sim.engine.tensors
        for k in ("contact_impulse", "limit_impulse", "self_contact_impulse", "attractor_impulse", "scene_warm"):
            if k in t:
                t[k][ids] = 0.0

    def _envs_of(self, sim, actor_indices, count, slot):
        """env ids of the sim-domain actor indices that address actor `slot`"""
        ids = actor_indices[:count].long()
        A = sim.nactors
        return torch.div(ids[ids % A == slot], A, rounding_mode="floor")

    def set_dof_state_tensor_indexed(self, sim, dof_state, actor_indices, count):
        n, nd = len(sim.envs), sim.asset.spec.nd
        ids = self._envs_of(sim, actor_indices, count, sim.robot)
        sim.engine.tensors["dof_state"][ids] = dof_state.view(n, nd, 2)[ids]
        self._clear_warm_start(sim, ids)
        return True

    def set_dof_state_tensor(self, sim, dof_state):
        n, nd = len(sim.envs), sim.asset.spec.nd
        sim.engine.tensors["dof_state"].copy_(dof_state.view(n, nd, 2))
        return True

    def set_actor_root_state_tensor_indexed(self, sim, root_states, actor_indices, count):
        n, A = len(sim.envs), sim.nactors
        src = root_states.view(n, A, 13)
        ids = self._envs_of(sim, actor_indices, count, sim.robot)
        if len(ids) and not sim.asset.spec.fixed_base:
            sim.engine.tensors["root_states"][ids] = src[ids, sim.robot]
            self._clear_warm_start(sim, ids)
        for k, phys in self._object_slots(sim):
            ids = self._envs_of(sim, actor_indices, count, k)
            if phys and len(ids):
                self._object_view(sim, k)[ids] = src[ids, k]
                if getattr(sim, "scene", None):        # a teleported box starts without last sub-step's contact impulses
                    sim.engine.tensors["scene_warm"][ids] = 0.0
            # (the goal object: the task's own tensor IS the state)
        if root_states.data_ptr() != self._buf(sim, "root", (n * A, 13)).data_ptr():
            self._buf(sim, "root", (n * A, 13)).view(n, A, 13)[:] = src
        return True

    def set_actor_root_state_tensor(self, sim, root_states):
        n, A = len(sim.envs), sim.nactors
        src = root_states.view(n, A, 13)
        if not sim.asset.spec.fixed_base:
            sim.engine.tensors["root_states"].copy_(src[:, sim.robot])
        for k, phys in self._object_slots(sim):
            if phys:
                self._object_view(sim, k).copy_(src[:, k])
        return True

    # ------------------------------------------------------------------ stepping
    def simulate(self, sim):
        self._flush_props(sim)
        sim.engine.simulate()
        sim.frame += 1
        if sim.one_shot_force:                      # gym applies body forces for one simulate() only
            sim.engine.tensors["object_force" if "object_force" in sim.engine.tensors else "forces"].zero_()
            sim.one_shot_force = False

    def fetch_results(self, sim, wait=True):
        pass

    # ------------------------------------------------------------------ viewer / rendering: headless engine
    def create_viewer(self, *a, **k):
        return None

    def subscribe_viewer_keyboard_event(self, *a, **k):
        pass

    def viewer_camera_look_at(self, *a, **k):
        pass

    def query_viewer_has_closed(self, viewer):
        return False

    def query_viewer_action_events(self, viewer):
        return []

    def step_graphics(self, sim):
        pass

    def draw_viewer(self, *a, **k):
        pass

    def sync_frame_time(self, sim):
        pass

    def poll_viewer_events(self, viewer):
        pass

    def write_viewer_image_to_file(self, *a, **k):
        pass

    def clear_lines(self, viewer):
        pass

    def add_lines(self, *a, **k):
        pass

    def destroy_viewer(self, viewer):
        pass

    def destroy_sim(self, sim):
        if sim.engine is not None:
            sim.engine.close()
            sim.engine = None


def _object_mass(object_type, task="ShadowHand"):
    from ...utils.config import compose
    cfgd = compose(overrides=[f"task={task}"])["task"]
    cfgd["env"]["objectType"] = object_type
    if task == "AllegroHand":
        from ...tasks.allegro_hand import allegro_params_from_cfg
        return float(allegro_params_from_cfg(cfgd).cube_mass)
    from ...tasks import shadow_hand as sh
    return float(sh.hand_params_from_cfg(cfgd).cube_mass)


def _quat_mul(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)


def _quat_rotate(q, v):
    qv, w = q[..., :3], q[..., 3:4]
    t = 2.0 * torch.cross(qv, v, dim=-1)
    return v + w * t + torch.cross(qv, t, dim=-1)


def _drive_mode(asset):
    return int(asset.options.default_dof_drive_mode) if asset.generic else DOF_MODE_EFFORT


def _dof_properties(spec, mode=DOF_MODE_EFFORT):
    dt = np.dtype([("hasLimits", "?"), ("lower", "f4"), ("upper", "f4"), ("driveMode", "i4"), ("velocity", "f4"), ("effort", "f4"),
                   ("stiffness", "f4"), ("damping", "f4"), ("friction", "f4"), ("armature", "f4")])
    out = np.zeros(spec.nd, dt)
    out["hasLimits"] = np.asarray(spec.dof_limited, bool)
    out["lower"], out["upper"] = spec.dof_lower, spec.dof_upper
    # a robot of its own reports the drive mode it was loaded with (AssetOptions.default_dof_drive_mode, amp/humanoid_amp_base.py:187)
    out["driveMode"] = mode
    out["velocity"], out["effort"] = spec.dof_velocity, spec.dof_effort
    out["stiffness"], out["damping"], out["armature"] = spec.dof_stiffness, spec.dof_damping, spec.dof_armature
    return out


_gym = None


def acquire_gym(*args):
    global _gym
    if _gym is None:
        _gym = Gym()
    return _gym

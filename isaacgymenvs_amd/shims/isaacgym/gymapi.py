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

Assets are resolved to the models compiled into the engine by file name (nv_ant.xml, nv_humanoid.xml, cartpole.urdf): the engine is
specialised per robot at build time (isaacgymenvs_amd/codegen.py), it does not load arbitrary files at run time.
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


class Transform:
    def __init__(self, p=None, r=None):
        self.p = p if p is not None else Vec3()
        self.r = r if r is not None else Quat()


class _Bag:
    """attribute bag with defaults: AssetOptions, PlaneParams, CameraProperties, the physx / flex sub-structs of SimParams"""

    def __init__(self, **defaults):
        self.__dict__.update(defaults)


class AssetOptions(_Bag):
    def __init__(self):
        super().__init__(default_dof_drive_mode=DOF_MODE_NONE, angular_damping=0.5, linear_damping=0.0, fix_base_link=False,
                         collapse_fixed_joints=False, density=1000.0, armature=0.0, thickness=0.02, disable_gravity=False,
                         replace_cylinder_with_capsule=False, flip_visual_attachments=False, max_angular_velocity=64.0,
                         max_linear_velocity=1000.0, use_mesh_materials=False)


class PlaneParams(_Bag):
    def __init__(self):
        super().__init__(normal=Vec3(0.0, 0.0, 1.0), distance=0.0, static_friction=1.0, dynamic_friction=1.0, restitution=0.0)


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
_MODEL_OF_FILE = {"nv_ant.xml": ("ant", "Ant"), "nv_humanoid.xml": ("humanoid", "Humanoid"), "cartpole.urdf": ("cartpole", "Cartpole")}


class _ActuatorProps:
    def __init__(self, gear):
        self.motor_effort = float(gear)
        self.kp = self.kv = 0.0


class _Asset:
    def __init__(self, path, options):
        from ...registry import load_model, load_selfcol, sensor_bodies
        key = os.path.basename(path)
        if key not in _MODEL_OF_FILE:
            raise NotImplementedError(f"gym.load_asset: {key} has no model compiled into the engine (available: {sorted(_MODEL_OF_FILE)}); "
                                      f"robots are specialised at build time (isaacgymenvs_amd/codegen.py)")
        self.model_name, self.task = _MODEL_OF_FILE[key]
        self.spec = load_model(self.model_name)
        self.options = options
        self.sensors = []                      # rigid-body indices in the order create_asset_force_sensor was called
        self.engine_sensor_bodies = [self.spec.api_body_names.index(self.spec.body_names[b]) for b in sensor_bodies(self.model_name, self.spec)]
        self.has_self_collision = load_selfcol(self.model_name) is not None


class _Env:
    def __init__(self, sim, index):
        self.sim, self.index = sim, index
        self.actors = []


class _Sim:
    def __init__(self, compute_device, graphics_device, physics_engine, params):
        self.compute_device, self.params = compute_device, params
        self.device = f"cuda:{compute_device}" if params.use_gpu_pipeline else "cpu"
        self.plane = None
        self.envs = []
        self.asset = None
        self.start_pose = None
        self.filter = 0
        self.engine = None
        self.frame = 0
        self.bufs = {}


class Gym:
    """the object `gymapi.acquire_gym()` returns"""

    # ------------------------------------------------------------------ setup
    def create_sim(self, compute_device=0, graphics_device=0, physics_engine=SIM_PHYSX, params=None):
        return _Sim(compute_device, graphics_device, physics_engine, params or SimParams())

    def add_ground(self, sim, plane_params):
        sim.plane = plane_params

    def load_asset(self, sim, rootpath, filename, options=None):
        return _Asset(os.path.join(rootpath, filename), options or AssetOptions())

    def get_asset_dof_count(self, asset):
        return asset.spec.nd

    def get_asset_rigid_body_count(self, asset):
        return len(asset.spec.api_body_names)

    def get_asset_joint_count(self, asset):
        # joints of the file incl. the fixed ones that weld bodies: one per non-root body (nv_humanoid.xml: 15 for 16 bodies)
        return len(asset.spec.api_body_names) - 1

    def get_asset_rigid_shape_count(self, asset):
        return len(asset.spec.geom_body)

    def get_asset_dof_names(self, asset):
        return list(asset.spec.dof_names)

    def get_asset_rigid_body_names(self, asset):
        return list(asset.spec.api_body_names)

    def get_asset_rigid_body_name(self, asset, index):
        return asset.spec.api_body_names[index]

    def find_asset_rigid_body_index(self, asset, name):
        return asset.spec.api_body_names.index(name)

    def find_asset_dof_index(self, asset, name):
        return list(asset.spec.dof_names).index(name)

    def get_asset_actuator_count(self, asset):
        return len(asset.spec.act_gear)

    def get_asset_actuator_properties(self, asset):
        return [_ActuatorProps(g) for g in asset.spec.act_gear]

    def get_asset_dof_properties(self, asset):
        return _dof_properties(asset.spec)

    def create_asset_force_sensor(self, asset, body_idx, local_pose, props=None):
        asset.sensors.append(int(body_idx))
        return len(asset.sensors) - 1

    def create_env(self, sim, lower, upper, num_per_row):
        env = _Env(sim, len(sim.envs))
        sim.envs.append(env)
        return env

    def create_actor(self, env, asset, pose, name="", group=-1, filter=-1, seg_id=0):
        sim = env.sim
        if sim.asset is not None and sim.asset is not asset:
            raise NotImplementedError("the shim runs one articulated actor per env (Cartpole, Ant, Humanoid)")
        sim.asset, sim.start_pose, sim.filter = asset, pose, int(filter)
        env.actors.append(name)
        return len(env.actors) - 1

    def begin_aggregate(self, *a, **k):
        return True

    def end_aggregate(self, *a, **k):
        return True

    def get_actor_dof_properties(self, env, actor):
        return _dof_properties(env.sim.asset.spec)

    def set_actor_dof_properties(self, env, actor, props):
        return True                             # drive modes / gains of the compiled models are fixed at build time

    def get_actor_rigid_body_count(self, env, actor):
        return len(env.sim.asset.spec.api_body_names)

    def get_actor_dof_count(self, env, actor):
        return env.sim.asset.spec.nd

    def get_actor_count(self, env):
        return len(env.actors)

    def find_actor_handle(self, env, name):
        return env.actors.index(name)

    def get_actor_index(self, env, actor, domain=DOMAIN_SIM):
        return env.index if domain == DOMAIN_SIM else actor

    def find_actor_rigid_body_handle(self, env, actor, name):
        return env.sim.asset.spec.api_body_names.index(name)

    def find_actor_dof_handle(self, env, actor, name):
        return list(env.sim.asset.spec.dof_names).index(name)

    def set_rigid_body_color(self, *a, **k):
        pass

    def set_actor_scale(self, *a, **k):
        return False

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
        return len(sim.envs)

    # ------------------------------------------------------------------ the engine comes to life
    def prepare_sim(self, sim):
        from ... import native
        from ...utils.config import compose
        if sim.asset is None or not sim.envs:
            raise RuntimeError("gym.prepare_sim: no actor was created")
        if sim.params.up_axis != UP_AXIS_Z:
            raise ValueError("only up_axis 'z' is implemented")
        asset, p, px = sim.asset, native.MiSimParams(), sim.params.physx
        p.dt, p.substeps = float(sim.params.dt), int(sim.params.substeps)
        for i, g in enumerate(sim.params.gravity):
            p.gravity[i] = float(g)
        p.iters = int(px.num_position_iterations) + int(px.num_velocity_iterations)
        p.contact_offset, p.rest_offset = float(px.contact_offset), float(px.rest_offset)
        p.max_depen_vel = float(px.max_depenetration_velocity)
        p.erp, p.cfm, p.warm, p.ground_z = 0.5, 1e-6, 1.0, 0.0
        p.plane_mu = float(sim.plane.static_friction) if sim.plane is not None else 1.0
        cfg = compose(overrides=[f"task={asset.task}"])["task"]        # the fused kernels' own parameters: unused by simulate()
        if asset.task == "Cartpole":
            from ...tasks.cartpole import cartpole_params_from_cfg
            tp = cartpole_params_from_cfg(cfg)
        else:
            from ...tasks.locomotion import loco_params_from_cfg
            tp = loco_params_from_cfg(cfg, asset.model_name, float(sim.start_pose.p.z))
        n = len(sim.envs)
        sim.engine = native.Engine(asset.task, p, tp, n, sim.device)
        if asset.has_self_collision:
            sim.engine.set_option("self_collision", 1 if sim.filter == 0 else 0)      # create_actor(..., filter): 0 = links collide
        t = sim.engine.tensors
        root = torch.tensor([sim.start_pose.p.x, sim.start_pose.p.y, sim.start_pose.p.z, sim.start_pose.r.x, sim.start_pose.r.y,
                             sim.start_pose.r.z, sim.start_pose.r.w, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=sim.device)
        if asset.spec.fixed_base:
            root[2] = t["root_states"][0, 2]        # a fixed base stays where the engine mounts it (the rail height of the cart-pole)
        t["root_states"][:] = root
        t["dof_state"].zero_()
        if asset.sensors and asset.sensors != asset.engine_sensor_bodies[:len(asset.sensors)]:
            raise NotImplementedError(f"force sensors on bodies {asset.sensors}: the compiled {asset.model_name} model has them on "
                                      f"{asset.engine_sensor_bodies}")
        return True

    # ------------------------------------------------------------------ tensor API
    def _buf(self, sim, name, shape):
        if name not in sim.bufs:
            sim.bufs[name] = torch.zeros(shape, dtype=torch.float32, device=sim.device)
        return sim.bufs[name]

    def acquire_actor_root_state_tensor(self, sim):
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

    def refresh_actor_root_state_tensor(self, sim):
        n = len(sim.envs)
        self._buf(sim, "root", (n, 13)).copy_(sim.engine.tensors["root_states"])
        return True

    def refresh_dof_state_tensor(self, sim):
        n, nd = len(sim.envs), sim.asset.spec.nd
        self._buf(sim, "dof", (n * nd, 2)).view(n, nd, 2).copy_(sim.engine.tensors["dof_state"])
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

    def refresh_rigid_body_state_tensor(self, sim):
        return True

    def refresh_net_contact_force_tensor(self, sim):
        return True

    def enable_actor_dof_force_sensors(self, env, actor):
        return True

    def set_dof_actuation_force_tensor(self, sim, forces):
        n, nd = len(sim.envs), sim.asset.spec.nd
        sim.engine.tensors["dof_actuation_force"].copy_(forces.view(n, nd))
        return True

    def _clear_warm_start(self, sim, ids):
        t = sim.engine.tensors
        for k in ("contact_impulse", "limit_impulse", "self_contact_impulse"):
            if k in t:
                t[k][ids] = 0.0

    def set_dof_state_tensor_indexed(self, sim, dof_state, actor_indices, count):
        n, nd = len(sim.envs), sim.asset.spec.nd
        ids = actor_indices[:count].long()
        sim.engine.tensors["dof_state"][ids] = dof_state.view(n, nd, 2)[ids]
        self._clear_warm_start(sim, ids)
        return True

    def set_dof_state_tensor(self, sim, dof_state):
        n, nd = len(sim.envs), sim.asset.spec.nd
        sim.engine.tensors["dof_state"].copy_(dof_state.view(n, nd, 2))
        return True

    def set_actor_root_state_tensor_indexed(self, sim, root_states, actor_indices, count):
        ids = actor_indices[:count].long()
        sim.engine.tensors["root_states"][ids] = root_states.view(len(sim.envs), 13)[ids]
        self._clear_warm_start(sim, ids)
        return True

    def set_actor_root_state_tensor(self, sim, root_states):
        sim.engine.tensors["root_states"].copy_(root_states.view(len(sim.envs), 13))
        return True

    # ------------------------------------------------------------------ stepping
    def simulate(self, sim):
        sim.engine.simulate()
        sim.frame += 1

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


def _dof_properties(spec):
    dt = np.dtype([("hasLimits", "?"), ("lower", "f4"), ("upper", "f4"), ("driveMode", "i4"), ("velocity", "f4"), ("effort", "f4"),
                   ("stiffness", "f4"), ("damping", "f4"), ("friction", "f4"), ("armature", "f4")])
    out = np.zeros(spec.nd, dt)
    out["hasLimits"] = np.asarray(spec.dof_limited, bool)
    out["lower"], out["upper"] = spec.dof_lower, spec.dof_upper
    out["driveMode"] = DOF_MODE_EFFORT
    out["velocity"], out["effort"] = spec.dof_velocity, spec.dof_effort
    out["stiffness"], out["damping"], out["armature"] = spec.dof_stiffness, spec.dof_damping, spec.dof_armature
    return out


_gym = None


def acquire_gym(*args):
    global _gym
    if _gym is None:
        _gym = Gym()
    return _gym

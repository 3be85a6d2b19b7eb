"""Robot descriptions the reference builds in code instead of loading from its asset tree.

quadcopter_mjcf(): the MJCF document `Quadcopter._create_quadcopter_asset` writes to ./quadcopter.xml before loading it
(reference isaacgymenvs/tasks/quadcopter.py:119-198): a cylindrical chassis with a free joint and four rotor arms at
45/135/225/315 degrees, each arm a small sphere body with a pitch hinge (axis y) carrying a rotor cylinder with a roll
hinge (axis x), both limited to +-30 degrees.  Same dimensions, densities, names and element order (=> same body / dof
order: chassis, rotor_arm0, rotor0, rotor_arm1, ...; rotor_pitch0, rotor_roll0, rotor_pitch1, ...).

ingenuity_mjcf(): the document `Ingenuity._create_ingenuity_asset` writes to ./ingenuity.xml (reference
isaacgymenvs/tasks/ingenuity.py:120-231), minus what cannot be restated: its three GLB meshes (../assets/glb/ingenuity/*.glb) are
not part of the reference tree, so the mesh geoms -- non-colliding (contype = conaffinity = 0), they only add mass -- are left out of
the chassis and each mesh-only "rotor_visual_i" body carries a stand-in blade (a thin box of the rotor's span) so that its free
hinge has an inertia.  Body / dof order as in the reference (6 bodies per env with the marker actor, 4 dofs): chassis,
rotor_physics_0, rotor_visual_0, rotor_physics_1, rotor_visual_1; rotor_roll0 (locked, range 0 0), rotor_roll0 (free, axis z),
rotor_roll1 (locked), rotor_roll1 (free).

balance_bot_mjcf(): the document `BallBalance._create_balance_bot_asset` writes to ./balance_bot.xml (reference
isaacgymenvs/tasks/ball_balance.py:136-224): a free tray (cylinder, radius 0.5) on three two-segment capsule legs at 0 / 120 / 240
degrees, upper_leg_joint_i limited to +-45 and lower_leg_joint_i to -70 .. 90 degrees; bodies tray, upper_leg0, lower_leg0,
upper_leg1, ..., dofs upper_leg_joint0, lower_leg_joint0, upper_leg_joint1, ...  balance_bot_dims() returns the lengths the task
keeps (:220-224).
"""
from __future__ import annotations

import math


def quadcopter_mjcf() -> str:
    chassis_radius, chassis_thickness = 0.1, 0.03                    # quadcopter.py:121-125
    rotor_radius, rotor_thickness, rotor_arm_radius = 0.04, 0.01, 0.01
    arm_offset = chassis_radius + 0.25 * rotor_arm_radius           # :146, along the arm's own x axis
    rotor_offset = rotor_radius + 0.25 * rotor_arm_radius           # :149
    out = ['<mujoco model="Quadcopter">',
           '  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>',
           '  <worldbody>',
           '    <body name="chassis" pos="0 0 0">',
           f'      <geom type="cylinder" size="{chassis_radius:g} {0.5 * chassis_thickness:g}" pos="0 0 0" density="50"/>',
           '      <joint name="root_joint" type="free"/>']
    for i, angle in enumerate((0.25 * math.pi, 0.75 * math.pi, 1.25 * math.pi, 1.75 * math.pi)):   # :151
        # Quat.from_axis_angle(z, angle) and the arm offset rotated by it (:155-156)
        qw, qz = math.cos(0.5 * angle), math.sin(0.5 * angle)
        px, py = arm_offset * math.cos(angle), arm_offset * math.sin(angle)
        out += [f'      <body name="rotor_arm{i}" pos="{px:g} {py:g} 0" quat="{qw:g} 0 0 {qz:g}">',
                f'        <geom type="sphere" size="{rotor_arm_radius:g}" density="200"/>',
                f'        <joint name="rotor_pitch{i}" type="hinge" pos="0 0 0" axis="0 1 0" limited="true" range="-30 30"/>',
                f'        <body name="rotor{i}" pos="{rotor_offset:g} 0 0" quat="1 0 0 0">',
                f'          <geom type="cylinder" size="{rotor_radius:g} {0.5 * rotor_thickness:g}" density="1000"/>',
                f'          <joint name="rotor_roll{i}" type="hinge" pos="0 0 0" axis="1 0 0" limited="true" range="-30 30"/>',
                '        </body>',
                '      </body>']
    out += ['    </body>', '  </worldbody>', '</mujoco>']
    return "\n".join(out) + "\n"


def ingenuity_mjcf() -> str:
    chassis_size = 0.06                                              # ingenuity.py:121-125
    rotor_radius, rotor_thickness = 0.15, 0.01
    blade = (rotor_radius, 0.01, 0.001)                              # stand-in for lower_prop.glb / upper_prop.glb (see the module docstring)
    out = ['<mujoco model="Ingenuity">',
           '  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>',
           '  <worldbody>',
           '    <body name="chassis" pos="0 0 0">',
           f'      <geom type="box" size="{chassis_size:g} {chassis_size:g} {chassis_size:g}" pos="0 0 0" density="50"/>',
           # the reference gives the chassis a hinge "root_joint" with range 0 0 (:176-180); the actor is created without
           # fix_base_link, so the chassis floats and that joint is not one of the 4 dofs (:61, dofs_per_env)
           '      <joint name="root_joint" type="free"/>']
    for i in range(2):                                               # :187-229, rotor_separation = (0, 0, 0.025)
        z = 0.025 * i
        out += [f'      <body name="rotor_physics_{i}" pos="0 0 {z:g}" quat="1 0 0 0">',
                f'        <geom type="cylinder" size="{rotor_radius:g} {0.5 * rotor_thickness:g}" density="1000"/>',
                f'        <joint name="rotor_roll{i}" type="hinge" limited="true" range="0 0" pos="0 0 0"/>',
                '      </body>',
                f'      <body name="rotor_visual_{i}" pos="0 0 {z:g}" quat="1 0 0 0">',
                f'        <geom type="box" size="{blade[0]:g} {blade[1]:g} {blade[2]:g}" density="1000"/>',
                f'        <joint name="rotor_roll{i}" type="hinge" axis="0 0 1" pos="0 0 0"/>',
                '      </body>']
    out += ['    </body>', '  </worldbody>', '</mujoco>']
    return "\n".join(out) + "\n"


def balance_bot_dims() -> dict:
    tray_radius, tray_thickness, leg_radius = 0.5, 0.02, 0.02           # ball_balance.py:139-141
    leg_outer_offset = tray_radius - 0.1
    leg_length = leg_outer_offset - 2 * leg_radius
    leg_inner_offset = leg_outer_offset - leg_length / math.sqrt(2)
    tray_height = leg_length * math.sqrt(2) + 2 * leg_radius + 0.5 * tray_thickness
    return dict(tray_radius=tray_radius, tray_thickness=tray_thickness, leg_radius=leg_radius, leg_outer_offset=leg_outer_offset,
                leg_length=leg_length, leg_inner_offset=leg_inner_offset, tray_height=tray_height,
                leg_angles=[0.0, 2.0 / 3.0 * math.pi, 4.0 / 3.0 * math.pi])


def _quat_from_euler_zyx(x, y, z):
    """gymapi.Quat.from_euler_zyx(x, y, z) as (w, x, y, z): the rotation Rz(z) Ry(y) Rx(x) -- the arguments are the angles about x, y
    and z in that order (only this reading puts the feet of legs 1 and 2 on their attractor targets, ball_balance.py:293-297)."""
    cz, sz, cy, sy, cx, sx = math.cos(z / 2), math.sin(z / 2), math.cos(y / 2), math.sin(y / 2), math.cos(x / 2), math.sin(x / 2)
    return (cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz)


def balance_bot_mjcf() -> str:
    d = balance_bot_dims()
    L, r = d["leg_length"], d["leg_radius"]
    out = ['<mujoco model="BalanceBot">',
           '  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>',
           '  <worldbody>',
           f'    <body name="tray" pos="0 0 {d["tray_height"]:g}">',
           '      <joint name="root_joint" type="free"/>',
           f'      <geom type="cylinder" size="{d["tray_radius"]:g} {0.5 * d["tray_thickness"]:g}" pos="0 0 0" density="100"/>']
    for i, angle in enumerate(d["leg_angles"]):                          # :165-218
        fx, fy, fz = d["leg_outer_offset"] * math.cos(angle), d["leg_outer_offset"] * math.sin(angle), -r - 0.5 * d["tray_thickness"]
        tx, ty, tz = d["leg_inner_offset"] * math.cos(angle), d["leg_inner_offset"] * math.sin(angle), fz - L / math.sqrt(2)
        px, py, pz = 0.5 * (fx + tx), 0.5 * (fy + ty), 0.5 * (fz + tz)
        uq = _quat_from_euler_zyx(0, -0.75 * math.pi, angle)
        lq = _quat_from_euler_zyx(0, -0.5 * math.pi, 0)
        out += [f'      <body name="upper_leg{i}" pos="{px:g} {py:g} {pz:g}" quat="{uq[0]:g} {uq[1]:g} {uq[2]:g} {uq[3]:g}">',
                f'        <geom type="capsule" size="{r:g} {0.5 * L:g}" density="1000"/>',
                f'        <joint name="upper_leg_joint{i}" type="hinge" pos="0 0 {-0.5 * L:g}" axis="0 1 0" limited="true" range="-45 45"/>',
                f'        <body name="lower_leg{i}" pos="{-0.5 * L:g} 0 {0.5 * L:g}" quat="{lq[0]:g} {lq[1]:g} {lq[2]:g} {lq[3]:g}">',
                f'          <geom type="capsule" size="{r:g} {0.5 * L:g}" density="1000"/>',
                f'          <joint name="lower_leg_joint{i}" type="hinge" pos="0 0 {-0.5 * L:g}" axis="0 1 0" limited="true" range="-70 90"/>',
                '        </body>',
                '      </body>']
    out += ['    </body>', '  </worldbody>', '</mujoco>']
    return "\n".join(out) + "\n"

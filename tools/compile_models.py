#!/usr/bin/env python
"""Compile the robot description files of the reference asset tree into flat ModelSpec JSON files.

Run in the development container (needs /root/reference/assets or --asset-root); the resulting
isaacgymenvs_amd/models/*.json are committed so that the GPU box (no /root/reference) can build the kernels.
Asset options mirror the reference task code (file:line cited per entry).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from isaacgymenvs_amd.assets.model import load_asset  # noqa: E402

ENTRIES = {
    # reference cartpole.py:72-88 (fix_base_link=True)
    # the rail sits 2 m above the plane (cartpole.py:93) and self-collision is filtered (:107): no contact geometry
    "cartpole": dict(file="urdf/cartpole.urdf", fix_base_link=True, collide_body_filter=lambda n: False),
    # reference ant.py:139-157
    "ant": dict(file="mjcf/nv_ant.xml"),
    # reference humanoid.py:142-157; the actor is created with collision filter 0 (humanoid.py:194): it collides with itself
    "humanoid": dict(file="mjcf/nv_humanoid.xml", self_collision=True),
    # reference anymal_terrain.py:214-231: collapse_fixed_joints, replace_cylinder_with_capsule, density 0.001 (only used
    # for links without <inertial>), armature 0, fix_base_link False
    "anymal": dict(file="urdf/anymal_c/urdf/anymal_minimal.urdf", density=0.001, replace_cylinder_with_capsule=True),
    # reference shadow_hand.py:234-245: fix_base_link, collapse_fixed_joints; the hand never touches the ground plane (it is
    # mounted 0.5 m above it, shadow_hand.py:303-304): no ground-contact spheres; its collision geometry is used against
    # the manipulated cube instead (extras below)
    "shadow_hand": dict(file="mjcf/open_ai_assets/hand/shadow_hand.xml", fix_base_link=True, collide_body_filter=lambda n: False),
    # reference allegro_hand.py:216-233: fix_base_link, collapse_fixed_joints, no gravity on the hand.  The task names
    # urdf/kuka_allegro_description/allegro.urdf, which the reference tree does not ship; the hand it does ship is
    # allegro_touch_sensor.urdf (the same 16-dof Allegro hand with the touch-sensor fingertips of the AllegroKuka tasks), all mesh collision
    # shapes -> spheres inscribed in the meshes' convex hulls (isaacgymenvs_amd/assets/mesh.py).  The mounting flange `allegro_mount`
    # (a 23 mm plate behind the wrist, 10 cm from the palm's inner face) is left without contact geometry: the cube is reset (fall_dist)
    # long before it could reach it.
    "allegro_hand": dict(file="urdf/kuka_allegro_description/allegro_touch_sensor.urdf", fix_base_link=True, collide_body_filter=lambda n: False),
    # the stock robot of the Articulation task (any other file is compiled at run time, assets/runtime.py): reference amp/humanoid_amp_base.py:177-186
    "articulation": dict(file="mjcf/amp_humanoid.xml"),
}


def allegro_extras(asset_root, spec_with_geoms):
    """The manipulation extras of the Allegro hand (models/allegro_hand_extras.json): object-contact spheres = the sphere geoms the mesh
    sampler produced, per body in farthest-point order (the engine admits a body's first few touching spheres as its manifold); no tendons;
    position drives with the gains the task sets (allegro_hand.py:256-264: stiffness 3, effort 0.5)."""
    import numpy as np
    spec = spec_with_geoms
    by_body = {}
    for g in range(len(spec.geom_body)):
        if int(spec.geom_type[g]) != 0:
            continue
        by_body.setdefault(int(spec.geom_body[g]), []).append((np.asarray(spec.geom_pos[g], float), float(spec.geom_size[g][0])))
    sph = []
    for b in sorted(by_body):
        items = by_body[b]
        P = np.array([q for q, _ in items])
        order = [int(np.argmax(np.linalg.norm(P - P.mean(0), axis=1)))]
        while len(order) < len(items):
            d = np.min(np.linalg.norm(P[:, None, :] - P[None, order, :], axis=2), axis=1)
            d[order] = -1.0
            order.append(int(np.argmax(d)))
        sph += [(b, items[i][0], items[i][1]) for i in order]
    nd = spec.nd
    return dict(os_body=[int(s[0]) for s in sph], os_pos=[[float(x) for x in s[1]] for s in sph], os_rad=[float(s[2]) for s in sph],
                tendons=[], tendon_limit_stiffness=0.0, tendon_damping=0.0, dof_kp=[3.0] * nd, dof_force_limit=[0.5] * nd,
                actuated_dofs=list(range(nd)), mount_quat=[0.0, 0.0, 0.0, 1.0], fingertips=[])


def hand_extras(asset_root, spec):
    """What the generic ModelSpec does not carry for the Shadow Hand (written to models/shadow_hand_extras.json):
      * object-contact spheres: the collision capsules / boxes of the hand sampled by spheres (capsules: along the axis,
        boxes: a grid in the mid-plane of the thin dimension) -- the cube is an exact box, the hand side is spheres;
      * the 4 fixed tendons the task uses (shared.xml:53-69; shadow_hand.py:256-266 sets limit_stiffness 30, damping 0.1);
      * position-actuator gains kp and force ranges (shared.xml:250-269), the mount orientation (robot.xml:3)."""
    import json
    import xml.etree.ElementTree as ET
    import numpy as np
    from isaacgymenvs_amd.assets.model import quat_to_mat
    hand_dir = os.path.join(asset_root, "mjcf/open_ai_assets/hand")
    sph = []
    for g in range(len(spec.geom_body)):
        b, t = int(spec.geom_body[g]), int(spec.geom_type[g])
        pos, R, size = np.asarray(spec.geom_pos[g], float), quat_to_mat(np.asarray(spec.geom_quat[g], float)), np.asarray(spec.geom_size[g], float)
        if t == 1:  # capsule: radius, half length along local z
            r, hl = size[0], size[1]
            n = max(2, int(np.ceil(2 * hl / (1.6 * r))) + 1)
            for k in range(n):
                z = -hl + 2 * hl * k / (n - 1)
                sph.append((b, pos + R @ np.array([0, 0, z]), r))
        elif t == 2 and size.min() > 0.005:  # box (the 1 mm thumb helper boxes are skipped)
            thin = int(np.argmin(size))
            r = size[thin]
            others = [a for a in range(3) if a != thin]
            grids = []
            for a in others:
                ext = size[a] - r
                n = max(1, int(np.ceil(2 * ext / (1.6 * r))) + 1) if ext > 0 else 1
                grids.append(np.linspace(-ext, ext, n) if n > 1 else np.array([0.0]))
            for u in grids[0]:
                for v in grids[1]:
                    p = np.zeros(3); p[others[0]] = u; p[others[1]] = v
                    sph.append((b, pos + R @ p, r))
    # Per body, order the spheres by farthest-point sampling (first the one farthest from the body's sphere centroid, then always
    # the one farthest from those already taken).  The engine admits at most BODY_CONTACT_CAP contacts per body in this order
    # (a contact manifold, like PhysX's <= 4 points per pair): with a spread-out order the first few touching spheres span the
    # contact patch instead of clustering in one corner of the palm's 30-sphere grid.
    by_body = {}
    for b, pos, r in sph:
        by_body.setdefault(b, []).append((pos, r))
    sph = []
    for b in sorted(by_body):
        items = by_body[b]
        P = np.array([q for q, _ in items])
        order = [int(np.argmax(np.linalg.norm(P - P.mean(0), axis=1)))]
        while len(order) < len(items):
            d = np.min(np.linalg.norm(P[:, None, :] - P[None, order, :], axis=2), axis=1)
            d[order] = -1.0
            order.append(int(np.argmax(d)))
        sph += [(b, items[i][0], items[i][1]) for i in order]
    shared = ET.parse(os.path.join(hand_dir, "shared.xml")).getroot()
    dofs = list(spec.dof_names)
    tendons = []
    for f in shared.find("tendon").findall("fixed"):
        if f.get("name") in ("robot0:T_FFJ1c", "robot0:T_MFJ1c", "robot0:T_RFJ1c", "robot0:T_LFJ1c"):
            js = f.findall("joint")
            lo, hi = [float(x) for x in f.get("range").split()]
            tendons.append(dict(name=f.get("name"), dof=[dofs.index(j.get("joint")) for j in js],
                                coef=[float(j.get("coef")) for j in js], range=[lo, hi]))
    # The asset's explicit hand-to-hand contact pairs (shared.xml:31-51; the hand's shapes are contype 1 / conaffinity 0, so these <pair>
    # entries are the ONLY hand-hand contacts MuJoCo -- and an importer that honours them -- generates; condim 1: frictionless).  One entry is
    # listed twice (C_lfdistal / C_rfdistal, :41 and :45): kept once.  Geometry in the body frames: a capsule as its axis segment + radius,
    # the palm box as centre + half sizes (side a).
    names = list(spec.geom_names)
    pairs, seen = [], set()
    for pr in shared.find("contact").findall("pair"):
        ga, gb = pr.get("geom1"), pr.get("geom2")
        if frozenset((ga, gb)) in seen:
            continue
        seen.add(frozenset((ga, gb)))
        if int(spec.geom_type[names.index(ga)]) != 2 and int(spec.geom_type[names.index(gb)]) == 2:
            ga, gb = gb, ga                                   # the box is side a
        side = []
        for gname in (ga, gb):
            g = names.index(gname)
            t, pos = int(spec.geom_type[g]), np.asarray(spec.geom_pos[g], float)
            R, size = quat_to_mat(np.asarray(spec.geom_quat[g], float)), np.asarray(spec.geom_size[g], float)
            if t == 1:
                ax = R @ np.array([0.0, 0.0, size[1]])
                side.append(dict(geom=gname, body=int(spec.geom_body[g]), kind="capsule", p0=[float(x) for x in pos - ax], p1=[float(x) for x in pos + ax],
                                 r=float(size[0])))
            else:
                assert t == 2 and np.allclose(R, np.eye(3)), gname          # C_palm0: axis-aligned in the palm's frame
                side.append(dict(geom=gname, body=int(spec.geom_body[g]), kind="box", p0=[float(x) for x in pos], p1=[float(x) for x in size], r=0.0))
        assert side[1]["kind"] == "capsule" and side[0]["body"] != side[1]["body"]
        pairs.append(dict(a=side[0], b=side[1], condim=int(pr.get("condim", 3))))
    kp = [0.0] * len(dofs)
    frc = [0.0] * len(dofs)
    act_dof = []
    for a in shared.find("actuator").findall("position"):
        d = dofs.index(a.get("joint"))
        kp[d] = float(a.get("kp"))
        frc[d] = float(a.get("forcerange").split()[1])
        act_dof.append(d)
    mount = ET.parse(os.path.join(hand_dir, "robot.xml")).getroot().find("body")
    ex, ey, ez = [float(x) for x in mount.get("euler").split()]
    from isaacgymenvs_amd.assets.model import quat_mul
    qx = np.array([np.sin(ex / 2), 0, 0, np.cos(ex / 2)]); qy = np.array([0, np.sin(ey / 2), 0, np.cos(ey / 2)]); qz = np.array([0, 0, np.sin(ez / 2), np.cos(ez / 2)])
    q = quat_mul(quat_mul(qx, qy), qz)  # MuJoCo default eulerseq "xyz" (intrinsic)
    return dict(os_body=[int(s[0]) for s in sph], os_pos=[[float(x) for x in s[1]] for s in sph], os_rad=[float(s[2]) for s in sph],
                tendons=tendons, tendon_limit_stiffness=30.0, tendon_damping=0.1, dof_kp=kp, dof_force_limit=frc, actuated_dofs=act_dof,
                mount_quat=[float(x) for x in q],
                fingertips=["robot0:ffdistal", "robot0:mfdistal", "robot0:rfdistal", "robot0:lfdistal", "robot0:thdistal"],
                pairs=pairs)


# robots the reference generates in code (isaacgymenvs_amd/assets/procedural.py restates the generators)
PROCEDURAL = {
    # reference quadcopter.py:119-198; the craft is reset below z = 0.3 m (:411) before it can reach the ground plane, so no
    # contact geometry is generated; thrust is applied at the four rotor bodies (:287-292)
    "quadcopter": dict(gen="quadcopter_mjcf", collide_body_filter=lambda n: False),
    # reference ingenuity.py:120-231; reset below z = 0.5 m (:449) with a 6 cm chassis and 15 cm rotors: the ground is out of reach,
    # the marker actor shares the craft's collision filter bit (:262, :268): no contact geometry
    "ingenuity": dict(gen="ingenuity_mjcf", collide_body_filter=lambda n: False),
    # reference ball_balance.py:136-224; the feet are pinned by attractors (:285-300) and the episode ends before the ball can reach
    # the legs or the ground (:473): the only contact is ball <-> tray, handled by the task's own engine (csrc/core/bbot_engine.hpp)
    "balance_bot": dict(gen="balance_bot_mjcf", collide_body_filter=lambda n: False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asset-root", default="/root/reference/assets")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isaacgymenvs_amd", "models"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for name, e in ENTRIES.items():
        kw = {k: v for k, v in e.items() if k not in ("file", "self_collision")}
        spec = load_asset(os.path.join(a.asset_root, e["file"]), name=name, **kw)
        spec.save(os.path.join(a.out, name + ".json"))
        if e.get("self_collision"):
            import json
            from isaacgymenvs_amd.assets.model import self_collision_tables
            with open(os.path.join(a.out, name + "_selfcol.json"), "w") as f:
                json.dump(self_collision_tables(spec), f, indent=1)
        if name == "shadow_hand":
            import json
            with open(os.path.join(a.out, "shadow_hand_extras.json"), "w") as f:
                full = load_asset(os.path.join(a.asset_root, e["file"]), name=name, fix_base_link=True)   # with collision geoms
                extras = hand_extras(a.asset_root, full)
                from isaacgymenvs_amd.assets.model import mjcf_self_collision_filter
                flt = mjcf_self_collision_filter(os.path.join(a.asset_root, e["file"]))
                # collision filter -1 (shadow_hand.py:357-358): which hand shapes the asset lets touch each other
                extras["self_collision_filter"] = dict(collision_geoms=flt["collision_geoms"], accepting=flt["accepting"], n_pairs=len(flt["pairs"]))
                json.dump(extras, f, indent=1)
        if name == "allegro_hand":
            import json
            import numpy as np
            # the joint constants the task writes into the dof properties of every actor (allegro_hand.py:256-264) replace the URDF's
            spec.dof_damping = np.full(spec.nd, 0.1)
            spec.dof_armature = np.full(spec.nd, 0.001)
            spec.save(os.path.join(a.out, name + ".json"))
            full = load_asset(os.path.join(a.asset_root, e["file"]), name=name, fix_base_link=True, mesh_spheres=True,
                              mesh_root=os.path.join(a.asset_root, "urdf"), mesh_link_filter=lambda link: link != "allegro_mount")
            with open(os.path.join(a.out, "allegro_hand_extras.json"), "w") as f:
                json.dump(allegro_extras(a.asset_root, full), f, indent=1)
        if name == "articulation":
            # as its task drives it (HumanoidAMP.yaml pdControl: True -> every dof DOF_MODE_POS, amp/humanoid_amp_base.py:219-222): the joints'
            # stiffness / damping are drive gains (engine parameters), not passive springs of the model -- so that the reference's default
            # configuration runs on the stock library without a run-time compile
            from isaacgymenvs_amd.assets.runtime import drive_split
            spec, _, _ = drive_split(spec, range(spec.nd))
            spec.save(os.path.join(a.out, name + ".json"))
        print(f"{name}: nb={spec.nb} nd={spec.nd} nv={spec.nv} nsph={len(spec.sph_body)} mass={spec.total_mass():.4f}")
    import tempfile
    from isaacgymenvs_amd.assets import procedural
    for name, e in PROCEDURAL.items():
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, name + ".xml")
            with open(path, "w") as f:
                f.write(getattr(procedural, e["gen"])())
            spec = load_asset(path, name=name, **{k: v for k, v in e.items() if k != "gen"})
        spec.save(os.path.join(a.out, name + ".json"))
        print(f"{name}: nb={spec.nb} nd={spec.nd} nv={spec.nv} nsph={len(spec.sph_body)} mass={spec.total_mass():.4f} bodies={list(spec.body_names)} dofs={list(spec.dof_names)}")


if __name__ == "__main__":
    main()

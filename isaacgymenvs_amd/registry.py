"""Which compiled robot models exist, their generated C++ struct names and per-task sensor placement."""
from __future__ import annotations

import os

from .assets.model import ModelSpec

_HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (struct name, sensor body names (reference file:line))
MODELS = {
    "cartpole": dict(struct="ModelCartpole", sensors=[]),
    # reference ant.py:170-178: force sensors on every body whose name contains "foot"
    "ant": dict(struct="ModelAnt", sensors=["front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot"]),
    # reference humanoid.py:162-168: right_foot, left_foot
    # the actor collides with itself (collision filter 0, humanoid.py:194): capsule pairs in models/humanoid_selfcol.json
    "humanoid": dict(struct="ModelHumanoid", sensors=["right_foot", "left_foot"], selfcol="humanoid_selfcol.json"),
    # reference anymal_terrain.py: no force sensors (it reads net contact forces per body, :119)
    "anymal": dict(struct="ModelAnymal", sensors=[]),
    # reference shadow_hand.py:291-297: force sensors on the five fingertips (distal links)
    "shadow_hand": dict(struct="ModelShadowHand", sensors=["robot0:ffdistal", "robot0:mfdistal", "robot0:rfdistal", "robot0:lfdistal",
                                                           "robot0:thdistal"], extras="shadow_hand_extras.json"),
    # reference allegro_hand.py: no force sensors (the acquire call is commented out, :148-150), no fingertip states in the observations
    "allegro_hand": dict(struct="ModelAllegroHand", sensors=[], extras="allegro_hand_extras.json"),
    # reference quadcopter.py:287-292: thrust forces act on the four rotor bodies (bodies 2, 4, 6, 8); they are listed as
    # "sensor" bodies because the engine records the world pose of exactly those bodies during its tree pass
    "quadcopter": dict(struct="ModelQuadcopter", sensors=["rotor0", "rotor1", "rotor2", "rotor3"]),
    # reference ingenuity.py:347-348: the thrust vectors act on bodies 1 and 3 (rotor_physics_0 / _1), in their local frames
    "ingenuity": dict(struct="ModelIngenuity", sensors=["rotor_physics_0", "rotor_physics_1"]),
    # reference ball_balance.py:285-300: attractors hold a point of each lower leg (the "sensor" bodies: the tree pass records their
    # poses); the reference's three force sensors sit on the tray itself (:254-260) and are computed from the tray's momentum balance
    "balance_bot": dict(struct="ModelBalanceBot", sensors=["lower_leg0", "lower_leg1", "lower_leg2"]),
    # the Articulation task's robot: compiled at run time from whatever file gym.load_asset names (assets/runtime.py); the stock build carries the
    # AMP humanoid (reference amp/humanoid_amp_base.py:177 mjcf/amp_humanoid.xml, force sensors on the feet :191-197)
    "articulation": dict(struct="ModelArticulation", sensors=["right_foot", "left_foot"]),
}


def load_extras(name):
    import json
    with open(os.path.join(_HERE, "models", MODELS[name]["extras"])) as f:
        return json.load(f)


def load_selfcol(name):
    """Self-collision tables of a model (assets/model.py::self_collision_tables), or None when the actor does not collide with itself."""
    import json
    f = MODELS.get(name, {}).get("selfcol")
    if not f:
        return None
    with open(os.path.join(_HERE, "models", f)) as fh:
        return json.load(fh)


def load_model(name) -> ModelSpec:
    return ModelSpec.load(os.path.join(_HERE, "models", name + ".json"))


def sensor_bodies(name, spec=None):
    spec = spec or load_model(name)
    return [spec.body_names.index(n) for n in MODELS[name]["sensors"]]


def generate_headers(out_dir=None):
    from .codegen import emit_model_header, write_if_changed
    out_dir = out_dir or os.path.join(_HERE, "csrc", "gen")
    paths = []
    for name, e in MODELS.items():
        spec = load_model(name)
        extras = None
        if e.get("extras"):
            import json
            with open(os.path.join(_HERE, "models", e["extras"])) as f:
                extras = json.load(f)
        txt = emit_model_header(spec, e["struct"], sensor_bodies(name, spec), extras, load_selfcol(name))
        p = os.path.join(out_dir, f"model_{name}.h")
        write_if_changed(p, txt)
        paths.append(p)
    return paths

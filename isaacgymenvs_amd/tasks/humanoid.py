"""Humanoid: 13 dynamic bodies / 21 hinge DoFs, torque control (reference isaacgymenvs/tasks/humanoid.py)."""
from .locomotion import LocomotionTask


class Humanoid(LocomotionTask):
    native_task = "Humanoid"
    model_name = "humanoid"
    start_height = 1.34  # humanoid.py:179

    def __init__(self, cfg, *args, **kw):
        super().__init__(cfg, *args, **kw)
        # The reference creates the actor with collision filter 0 (humanoid.py:194): its links collide with each other.  The engine
        # does the same by default; `env.selfCollision: False` (not a reference key) restores the filter-1 behaviour of the Ant.
        self.self_collision = bool(cfg["env"].get("selfCollision", True))
        if not self.self_collision:
            self.engine.set_option("self_collision", 0)
        t = self.engine.tensors
        self.self_contact_impulse = t["self_contact_impulse"]      # [N, 13 limb pairs, 3] normal + 2 tangent impulses
        self.self_contact_force = t["self_contact_force"]          # [N, 13, 3] world force on the first body of each pair's contact

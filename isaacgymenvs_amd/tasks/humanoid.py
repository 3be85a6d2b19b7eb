"""Humanoid: 13 dynamic bodies / 21 hinge DoFs, torque control (reference isaacgymenvs/tasks/humanoid.py)."""
from .locomotion import LocomotionTask


class Humanoid(LocomotionTask):
    native_task = "Humanoid"
    model_name = "humanoid"
    start_height = 1.34  # humanoid.py:179

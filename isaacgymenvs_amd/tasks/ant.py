"""Ant: 9 bodies / 8 hinge DoFs / 13 contact spheres, torque control (reference isaacgymenvs/tasks/ant.py)."""
from .locomotion import LocomotionTask


class Ant(LocomotionTask):
    native_task = "Ant"
    model_name = "ant"
    start_height = 0.44  # ant.py:164

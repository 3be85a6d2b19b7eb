"""`gym` stand-in: the reference only uses `gym.spaces.Box` (vec_task.py:34-35,104-113)."""
from . import spaces  # noqa: F401

_mi_shim = True
__all__ = ["spaces"]


class Space:
    pass


class Env:
    pass

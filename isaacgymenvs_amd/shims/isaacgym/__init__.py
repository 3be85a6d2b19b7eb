"""`isaacgym` stand-in (see isaacgymenvs_amd/shims/__init__.py)."""
from . import gymapi, gymtorch, gymutil, terrain_utils, torch_utils  # noqa: F401

_mi_shim = True
__all__ = ["gymapi", "gymtorch", "gymutil", "terrain_utils", "torch_utils"]

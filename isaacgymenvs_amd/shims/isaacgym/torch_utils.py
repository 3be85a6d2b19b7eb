"""`isaacgym.torch_utils`: the reference re-implements these in isaacgymenvs/utils/torch_jit_utils.py and its tasks import from there;
this module exists so that `from isaacgym.torch_utils import *` in user code resolves (to nothing)."""
__all__ = []

"""`isaacgym.torch_utils`: Isaac Gym ships these helpers and its examples import them with `from isaacgym.torch_utils import *`
(`to_torch`, `quat_mul`, `quat_rotate_inverse`, `get_axis_params`, `torch_rand_float`, `tensor_clamp`, ...).  The reference restates the same
names in `isaacgymenvs/utils/torch_jit_utils.py`; this repo's restatement lives in `isaacgymenvs_amd/utils/torch_jit_utils.py` and this
module re-exports it, so user code written against either import keeps working on the stand-in."""
from ...utils import torch_jit_utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_") and callable(getattr(_impl, n)) and getattr(getattr(_impl, n), "__module__", "") == _impl.__name__]
globals().update({n: getattr(_impl, n) for n in __all__})
# the two aliases are plain assignments in the implementation module (their __module__ is the target's): exported by name
for _n in ("my_quat_rotate", "saturate"):
    globals()[_n] = getattr(_impl, _n)
    if _n not in __all__:
        __all__.append(_n)

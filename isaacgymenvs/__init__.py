"""`import isaacgymenvs` -- the name user scripts of the reference use (README.md:33-51):

    import isaacgymenvs
    envs = isaacgymenvs.make(seed=0, task="Ant", num_envs=2000, sim_device="cuda:0", rl_device="cuda:0")

This package is only a name: everything resolves to `isaacgymenvs_amd` (the MI355X-native engine), including the sub-packages
`isaacgymenvs.tasks`, `isaacgymenvs.utils` and the task-config directory `isaacgymenvs/cfg` that scripts address by path.
"""
import importlib
import os
import sys

import isaacgymenvs_amd as _impl

make = _impl.make
__version__ = _impl.__version__

for _name in ("tasks", "utils", "registry", "native", "parallel"):
    _mod = importlib.import_module(f"isaacgymenvs_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    globals()[_name] = _mod
for _name in ("tasks.base", "tasks.base.vec_task", "utils.utils", "utils.dr_utils", "utils.rlgames_utils", "utils.config", "utils.torch_jit_utils"):
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"isaacgymenvs_amd.{_name}")

#: where the Hydra-style task configs live (the reference's `isaacgymenvs/cfg`)
CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(_impl.__file__)), "cfg")

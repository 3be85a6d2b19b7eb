"""Import-name shims that let UNMODIFIED task files of the reference run on this engine (SURVEY.md 8b "B-inner").

The reference's tasks do `from isaacgym import gymtorch, gymapi` (the closed Isaac Gym Preview-4 binary, README.md:13-19) and
`import gym` (OpenAI gym, only for `spaces.Box`).  `install()` registers this package's `isaacgym` and `gym` stand-ins under those
names when the real modules are absent:

    import isaacgymenvs_amd.shims as shims
    shims.install()
    from isaacgymenvs.tasks.ant import Ant          # the reference's own file

Scope: the ~60 entry points the reference's base class and the Cartpole / Ant / Humanoid tasks call (setup getters, the tensor
API, simulate / refresh / set_*), backed by `isaacgymenvs_amd.native.Engine`: `simulate` is `mi_engine_simulate`, the `acquire_*`
tensors are AoS copies that `refresh_*` fills from and `set_*` commits to the engine's SoA arena (gym's own refresh / set semantics).
Viewer and camera entry points exist and do nothing.
"""
import importlib
import sys

import numpy as np


def install(force=False):
    """Register the stand-ins as `isaacgym` / `gym` (only where the real packages are missing, unless force)."""
    if not hasattr(np, "Inf"):
        np.Inf = np.inf                       # the reference still spells it np.Inf (vec_task.py:107), removed in NumPy 2
    for name in ("isaacgym", "gym"):
        if not force:
            try:
                if name not in sys.modules:
                    importlib.import_module(name)
                if not getattr(sys.modules[name], "_mi_shim", False):
                    continue                  # a real package is installed: leave it alone
            except ImportError:
                pass
        pkg = importlib.import_module(f"{__name__}.{name}")
        sys.modules[name] = pkg
        for sub in pkg.__all__:
            sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__name__}.{name}.{sub}")

"""Library behaviours the reference's newer task files were written against and this image's versions dropped -- for running those files
UNMODIFIED, beside the `isaacgym` / `gym` stand-ins (shims/__init__.py already restores `np.Inf`, gone in NumPy 2):

* `tkinter`, `omegaconf`: imported by tasks/dextreme (`from tkinter import W`, allegro_hand_dextreme.py:34 -- unused; `from omegaconf import
  ListConfig`, adr_vec_task.py:54 -- an isinstance check) and absent from this image: one-name stand-ins, only while nothing real is installed;
* `torch.where(condition, ...)` with an INTEGER condition: accepted by the torch the reference pins (1.x: uint8 / long masks), an error since
  torch 2.x ("where expected condition to be a boolean tensor").  The dextreme reward passes `reset_goal_buf`-derived Long tensors
  (allegro_hand_dextreme.py:1630).  TorchScript resolves `torch.where` to the builtin, which a Python-level wrapper cannot reach, so inside
  `environment()` `torch.jit.script` is the identity: the task's own jitted functions run eagerly (same arithmetic, same results), with the wrapper.

Everything is undone on exit.  Nothing of the engine depends on this module; tests/test_gymapi_shim.py uses it for the dextreme task only.

NOT THREAD-SAFE, and not for production use: inside `environment()` `torch.where` and `torch.jit.script` are patched PROCESS-WIDE -- a
concurrent thread sees the patched functions, and a module first imported inside the context keeps its functions un-scripted for the life of the
process.  Use it from one thread, around the construction and stepping of a legacy task file only.  It serves `tasks/dextreme/`, which SURVEY 2 #16
marks out of scope: kept for the tests that exist, not developed further."""
import contextlib
import sys
import types

import torch


@contextlib.contextmanager
def environment():
    added = []
    for name, attrs in (("tkinter", {"W": "w"}), ("omegaconf", {"ListConfig": type("ListConfig", (list,), {}), "DictConfig": type("DictConfig", (dict,), {})})):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.__dict__.update(attrs)
                m._mi_shim = True
                sys.modules[name] = m
                added.append(name)
    where, script = torch.where, torch.jit.script

    def where_with_integer_masks(condition, *args, **kwargs):
        if torch.is_tensor(condition) and condition.dtype != torch.bool:
            condition = condition != 0
        return where(condition, *args, **kwargs)

    def script_eagerly(obj=None, *args, **kwargs):
        return obj

    torch.where, torch.jit.script = where_with_integer_masks, script_eagerly
    try:
        yield
    finally:
        torch.where, torch.jit.script = where, script
        for name in added:
            sys.modules.pop(name, None)

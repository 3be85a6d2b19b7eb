"""isaacgymenvs_amd -- MI355X-native vectorised RL-environment engine with the IsaacGymEnvs public API.

`make()` mirrors reference isaacgymenvs/__init__.py:14-55 (same signature and quirks: `seed` is not consumed by
env creation on the reference path -- here it additionally seeds the in-kernel reset RNG; `num_envs` is applied
only when no cfg is passed).
"""
from __future__ import annotations

import os

__version__ = "0.1.0"


def make(seed: int, task: str, num_envs: int, sim_device: str, rl_device: str, graphics_device_id: int = -1,
         headless: bool = False, multi_gpu: bool = False, virtual_screen_capture: bool = False,
         force_render: bool = True, cfg=None):
    from .tasks import isaacgym_task_map
    from .utils.config import compose, omegaconf_to_dict
    if cfg is None:
        root = compose("config", overrides=[f"task={task}"])
        cfg_dict = omegaconf_to_dict(root["task"])
        cfg_dict["env"]["numEnvs"] = num_envs
    else:
        cfg_dict = omegaconf_to_dict(cfg["task"] if "task" in cfg and hasattr(cfg["task"], "items") and "env" in cfg["task"] else cfg)
    if multi_gpu:  # reference utils/rlgames_utils.py:89-107: one process per GPU, device = cuda:LOCAL_RANK
        local_rank = int(os.getenv("LOCAL_RANK", "0"))
        if not str(sim_device).startswith("cpu"):      # (a sharded job on the CPU backend keeps its device: one process per shard, gloo)
            sim_device = f"cuda:{local_rank}"
            rl_device = f"cuda:{local_rank}"
        cfg_dict["_multi_gpu"] = True
    cfg_dict["_seed"] = int(seed) if seed is not None and seed >= 0 else 0
    # Contract under multi_gpu: the caller passes seed + RANK, as the reference's launcher does (rlgames_utils.py:89-107,
    # utils/utils.py:94 `seed += rank`); whatever must be IDENTICAL on every rank -- the AnymalTerrain height field -- is seeded with
    # the job's base seed = seed - RANK.  A caller that passes the same seed on every rank (README-style make() per rank) says so with
    # cfg["_base_seed"]; without it the subtraction is clamped at 0 (np.random.RandomState rejects negative seeds) and the ranks of
    # such a job would build different terrains.
    if "_base_seed" in cfg_dict:
        cfg_dict["_terrain_seed"] = max(0, int(cfg_dict["_base_seed"]))
    else:
        cfg_dict["_terrain_seed"] = max(0, cfg_dict["_seed"] - (int(os.getenv("RANK", "0")) if multi_gpu else 0))
    name = cfg_dict["name"]
    if name not in isaacgym_task_map:
        raise KeyError(name)
    return isaacgym_task_map[name](cfg=cfg_dict, rl_device=rl_device, sim_device=sim_device,
                                   graphics_device_id=graphics_device_id, headless=headless,
                                   virtual_screen_capture=virtual_screen_capture, force_render=force_render)

"""rl_games adapter surface (reference isaacgymenvs/utils/rlgames_utils.py:242-295).

`RLGPUEnv` is what the reference registers with rl_games' vecenv factory: a thin pass-through to the VecTask that adds
`get_env_info`, `get_number_of_agents`, `set_train_info`, `get_env_state` / `set_env_state`.  rl_games is not installable in
this environment, so the class does not inherit from `rl_games.common.vecenv.IVecEnv` unless that import works; the method
set (names, arguments, return values) is the reference's, which is all rl_games' runner relies on (duck typing).

    from isaacgymenvs_amd.utils.rlgames_utils import RLGPUEnv, get_rlgames_env_creator, register_with_rl_games
"""
from __future__ import annotations

try:  # pragma: no cover - rl_games is absent here
    from rl_games.common import env_configurations, vecenv
    _Base = vecenv.IVecEnv
except Exception:  # noqa: BLE001
    env_configurations = vecenv = None
    _Base = object


def get_rlgames_env_creator(seed: int, task_config: dict, task_name: str, sim_device: str, rl_device: str,
                            graphics_device_id: int = -1, headless: bool = True, multi_gpu: bool = False,
                            post_create_hook=None, virtual_screen_capture: bool = False, force_render: bool = False):
    """rlgames_utils.py:53-117: returns a zero-argument creator of the vectorised task.  With multi_gpu every rank builds
    its own shard on cuda:LOCAL_RANK and uses seed + rank (rlgames_utils.py:89-107, utils/utils.py:94)."""
    def create_rlgpu_env():
        import os
        import isaacgymenvs_amd
        _seed, _sim, _rl = seed, sim_device, rl_device
        if multi_gpu:
            local_rank = int(os.getenv("LOCAL_RANK", "0"))
            _sim = _rl = f"cuda:{local_rank}"
            _seed = seed + int(os.getenv("RANK", "0"))
        env = isaacgymenvs_amd.make(seed=_seed, task=task_name, num_envs=task_config["env"]["numEnvs"], sim_device=_sim, rl_device=_rl,
                                    graphics_device_id=graphics_device_id, headless=headless, multi_gpu=multi_gpu,
                                    virtual_screen_capture=virtual_screen_capture, force_render=force_render, cfg=task_config)
        if post_create_hook is not None:
            post_create_hook()
        return env
    return create_rlgpu_env


class RLGPUEnv(_Base):
    """rlgames_utils.py:242-295.  `env_creator` replaces the lookup in rl_games' env_configurations registry when rl_games
    is not importable; with rl_games present the reference's (config_name, num_actors, **kwargs) form works unchanged."""

    def __init__(self, config_name=None, num_actors=None, env_creator=None, **kwargs):
        if env_creator is None:
            if env_configurations is None:
                raise RuntimeError("rl_games is not installed: pass env_creator=...")
            env_creator = env_configurations.configurations[config_name]["env_creator"]
        self.env = env_creator(**kwargs)

    def step(self, actions):
        return self.env.step(actions)

    def reset(self):
        return self.env.reset()

    def reset_done(self):
        return self.env.reset_done()

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if hasattr(self.env, "amp_observation_space"):
            info["amp_observation_space"] = self.env.amp_observation_space
        if self.env.num_states > 0:
            info["state_space"] = self.env.state_space
        return info

    def set_train_info(self, env_frames, *args_, **kwargs_):
        if hasattr(self.env, "set_train_info"):
            self.env.set_train_info(env_frames, *args_, **kwargs_)

    def get_env_state(self):
        return self.env.get_env_state() if hasattr(self.env, "get_env_state") else None

    def set_env_state(self, env_state):
        if hasattr(self.env, "set_env_state"):
            self.env.set_env_state(env_state)


def register_with_rl_games(create_env_thunk):
    """train.py:151-158: vecenv.register('RLGPU', ...) + env_configurations.register('rlgpu', ...)."""
    if vecenv is None:
        raise RuntimeError("rl_games is not installed")
    vecenv.register("RLGPU", lambda config_name, num_actors, **kwargs: RLGPUEnv(config_name, num_actors, **kwargs))
    env_configurations.register("rlgpu", {"vecenv_type": "RLGPU", "env_creator": lambda **kwargs: create_env_thunk(**kwargs)})

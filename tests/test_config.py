"""Config composition: the mini-Hydra composer must resolve the reference's interpolation forms."""
import pytest

from isaacgymenvs_amd.utils.config import compose


def test_ant_matches_hand_resolved_values():
    t = compose(overrides=["task=Ant"])["task"]
    assert t["name"] == "Ant" and t["physics_engine"] == "physx"
    assert t["env"]["numEnvs"] == 4096 and t["env"]["episodeLength"] == 1000
    assert t["env"]["terminationHeight"] == 0.31 and t["env"]["deathCost"] == -2.0
    s = t["sim"]
    assert s["dt"] == 0.0166 and s["substeps"] == 2 and s["use_gpu_pipeline"] is True
    assert s["physx"]["num_threads"] == 4 and s["physx"]["solver_type"] == 1 and s["physx"]["use_gpu"] is True
    assert s["physx"]["num_position_iterations"] == 4 and s["physx"]["max_depenetration_velocity"] == 10.0


def test_overrides_and_resolvers():
    c = compose(overrides=["task=Cartpole", "num_envs=64", "pipeline=cpu", "sim_device=cpu", "seed=7"])
    t = c["task"]
    assert c["seed"] == 7 and c["task_name"] == "Cartpole"
    assert t["env"]["numEnvs"] == 64 and t["sim"]["use_gpu_pipeline"] is False and t["sim"]["physx"]["use_gpu"] is False
    assert t["env"]["clipObservations"] == 5.0 and t["env"]["maxEffort"] == 400.0
    h = compose(overrides=["task=Humanoid", "task.env.powerScale=0.5"])["task"]
    assert h["env"]["powerScale"] == 0.5 and h["env"]["angularVelocityScale"] == 0.25


def test_unknown_task_raises():
    with pytest.raises(KeyError):
        compose(overrides=["task=DoesNotExist"])


# ------------------------------------------------------------------ config variants of the supported tasks (reference cfg/task/*.yaml)
REF_TASK_DIR = "/root/reference/isaacgymenvs/cfg/task"


def _reference_task_yaml(name):
    """The reference YAML with its `defaults: [Base, _self_]` inheritance applied by hand (plain yaml, no Hydra)."""
    import os
    import yaml
    t = yaml.safe_load(open(os.path.join(REF_TASK_DIR, name + ".yaml")))
    base = {}
    for item in t.pop("defaults", None) or []:
        if item != "_self_":
            base = _deep_merge(base, _reference_task_yaml(item))
    return _deep_merge(base, t)


def _deep_merge(a, b):
    out = dict(a)
    for k, v in b.items():
        out[k] = _deep_merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def test_variant_configs_compose_to_the_reference_values():
    f = compose(overrides=["task=ShadowHandOpenAI_FF"])["task"]
    assert f["name"] == "ShadowHand" and f["env"]["numEnvs"] == 16384
    e = f["env"]
    assert (e["resetTime"], e["controlFrequencyInv"], e["actionsMovingAverage"], e["forceScale"], e["fallPenalty"]) == (8, 3, 0.3, 1.0, -50.0)
    assert (e["observationType"], e["asymmetric_observations"], e["successTolerance"], e["maxConsecutiveSuccesses"]) == ("openai", True, 0.4, 50)
    assert f["task"]["randomize"] is True and f["sim"]["physx"]["num_position_iterations"] == 8
    assert compose(overrides=["task=ShadowHandOpenAI_LSTM"])["task"]["env"]["numEnvs"] == 8192
    t = compose(overrides=["task=ShadowHandTest"])["task"]
    assert t["env"]["numEnvs"] == 256 and t["env"]["resetTime"] == 80 and t["task"]["randomization_params"]["frequency"] == 480000
    assert t["task"]["randomization_params"]["actions"]["schedule"] == "constant"
    assert t["task"]["randomization_params"]["actions"]["range_correlated"] == [0, .015]       # inherited through two levels
    for name, base in (("AntSAC", "Ant"), ("HumanoidSAC", "Humanoid")):
        c = compose(overrides=[f"task={name}"])["task"]
        assert c["name"] == base and c["env"]["numEnvs"] == 64
        assert c["sim"] == compose(overrides=[f"task={base}"])["task"]["sim"]
    from isaacgymenvs_amd.tasks.shadow_hand import hand_max_episode_length
    assert hand_max_episode_length(f) == 160 and hand_max_episode_length(t) == 1600               # shadow_hand.py:139-140
    assert hand_max_episode_length(compose(overrides=["task=ShadowHand"])["task"]) == 600


@pytest.mark.parametrize("name", ["Ant", "AntSAC", "Humanoid", "HumanoidSAC", "Cartpole", "Anymal", "AnymalTerrain", "Quadcopter", "Ingenuity", "BallBalance", "ShadowHand", "AllegroHand",
                                  "ShadowHandOpenAI_FF", "ShadowHandOpenAI_LSTM", "ShadowHandTest"])
def test_env_section_equals_the_reference_yaml(name):
    """Every scalar the reference's task YAML sets in `env:` and in the noise part of `task:` has the same value here (the unresolved
    `${...}` interpolations are compared as text).  Needs /root/reference: skipped on the GPU box."""
    import os
    if not os.path.isdir(REF_TASK_DIR):
        pytest.skip("reference tree not present")
    ref = _reference_task_yaml(name)
    ours = compose(overrides=[f"task={name}"], resolve=False)["task"]

    def check(r, o, path):
        for k, v in r.items():
            if k in ("asset", "actor_params", "viewer", "assetRoot"):      # asset paths are resolved by the registry; actor_params: no engine counterpart
                continue
            assert k in o, f"{name}: {path}{k} missing"
            if isinstance(v, dict):
                check(v, o[k], path + k + ".")
            else:
                assert o[k] == v, f"{name}: {path}{k} = {o[k]!r}, reference {v!r}"
    check(ref["env"], ours["env"], "env.")
    check(ref["task"], ours["task"], "task.")
    check(ref["sim"], ours["sim"], "sim.")


def test_non_zero_restitution_is_refused_not_ignored():
    """Every shipped config has restitution 0 and the engine's contact model has no restitution term: a config that asks for one must fail
    loudly instead of running without it (reference plane params: ant.py:137-143, anymal_terrain.py:198-206)."""
    import isaacgymenvs_amd
    from isaacgymenvs_amd.utils.config import omegaconf_to_dict
    for task, path in (("Ant", ("env", "plane", "restitution")), ("AnymalTerrain", ("env", "terrain", "restitution"))):
        cfg = omegaconf_to_dict(compose("config", overrides=[f"task={task}"])["task"])
        node = cfg
        for k in path[:-1]:
            node = node[k]
        node[path[-1]] = 0.3
        cfg["env"]["numEnvs"] = 4
        with pytest.raises(NotImplementedError, match="restitution"):
            isaacgymenvs_amd.make(seed=0, task=task, num_envs=4, sim_device="cpu", rl_device="cpu", headless=True, cfg=cfg)


def test_legacy_environment_restores_what_it_changes():
    """shims/legacy.py: inside environment() torch.where takes integer masks (torch 1.x behaviour the reference's dextreme reward relies on),
    torch.jit.script is the identity and the tkinter / omegaconf names resolve; afterwards everything is as before."""
    import sys
    import torch
    from isaacgymenvs_amd.shims import legacy
    where, script = torch.where, torch.jit.script
    had = {m: m in sys.modules for m in ("tkinter", "omegaconf")}
    mask = torch.tensor([1, 0, 2])
    with legacy.environment():
        assert torch.where(mask, torch.ones(3), torch.zeros(3)).tolist() == [1.0, 0.0, 1.0]
        assert torch.where(mask > 0, torch.ones(3), torch.zeros(3)).tolist() == [1.0, 0.0, 1.0]

        @torch.jit.script
        def f(x):
            return torch.where(x, x, x)                      # (would not even compile for a Long x under TorchScript)
        assert f(mask).tolist() == [1, 0, 2]
        from tkinter import W  # noqa: F401
        from omegaconf import ListConfig
        assert isinstance(ListConfig([1, 2]), list) or hasattr(ListConfig, "__mro__")
    assert torch.where is where and torch.jit.script is script
    assert {m: m in sys.modules for m in had} == had
    with pytest.raises(RuntimeError):
        torch.where(mask, torch.ones(3), torch.zeros(3))


def test_task_config_group_defaults(tmp_path):
    """`defaults: [_self_, {env: reorientation}]` inside a task file (reference cfg/task/AllegroKuka.yaml:1-3 with cfg/task/env/*.yaml): the
    option's content lands under the group's key, entries are merged in list order (what follows `_self_` overrides the file's own keys), and an
    unknown option is an error"""
    from isaacgymenvs_amd.utils.config import _load_task_yaml
    (tmp_path / "task" / "env").mkdir(parents=True)
    (tmp_path / "task" / "env" / "reorientation.yaml").write_text("subtask: reorientation\nepisodeLength: 600\n")
    (tmp_path / "task" / "Kuka.yaml").write_text("defaults:\n  - _self_\n  - env: reorientation\nname: Kuka\nenv:\n  numEnvs: 8192\n  episodeLength: 400\n")
    (tmp_path / "task" / "KukaLSTM.yaml").write_text("defaults:\n  - Kuka\n  - _self_\nenv:\n  numEnvs: 64\n")
    t = _load_task_yaml("Kuka", str(tmp_path))
    assert t["name"] == "Kuka" and t["env"] == {"subtask": "reorientation", "episodeLength": 600, "numEnvs": 8192}      # the option came after _self_
    t2 = _load_task_yaml("KukaLSTM", str(tmp_path))
    assert t2["env"]["numEnvs"] == 64 and t2["env"]["subtask"] == "reorientation"
    (tmp_path / "task" / "Bad.yaml").write_text("defaults:\n  - env: nothing\nname: Bad\n")
    with pytest.raises(KeyError):
        _load_task_yaml("Bad", str(tmp_path))
    import os
    if os.path.isdir(REF_TASK_DIR):          # the reference's own file (development container only)
        ref = _load_task_yaml("AllegroKuka", os.path.dirname(REF_TASK_DIR))
        assert ref["env"]["subtask"] == "reorientation" and ref["name"] == "AllegroKuka"

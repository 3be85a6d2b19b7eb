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

"""GPU kernels against the CPU oracle over LONG random-policy rollouts, statistically.  Contact dynamics is chaotic -- trajectories part
after a few dozen steps, which is why the step-by-step parity tests are short -- but if both run the same physics their reset counts,
mean rewards and speed scales agree.  A defect that needs hundreds of steps to build up (the Ingenuity's locked rotor joints once: 100x
the joint speeds and visibly more resets on the GPU after ~230 steps) shows here and nowhere in the short tests."""
import numpy as np
import pytest
import torch

from isaacgymenvs_amd.registry import load_model, sensor_bodies
from test_gpu_parity import DEV, _make_env, _oracle_kw, _sim_dict

pytestmark = pytest.mark.gpu


def _oracle(task, env, n, seed):
    from isaacgymenvs_amd.assets.procedural import balance_bot_dims
    from oracle import tasks as OT
    sd, p = _sim_dict(env.sim_params), env._task_params_struct
    if task in ("Ant", "Humanoid"):
        return OT.OracleLocomotionEnv(task == "Humanoid", load_model(task.lower()), sensor_bodies(task.lower()), sd, p, n, seed=seed,
                                      precision="f64", **_oracle_kw(task, env))
    if task == "Quadcopter":
        return OT.OracleQuadcopterEnv(load_model("quadcopter"), sensor_bodies("quadcopter"), sd, p, n, seed=seed, precision="f64")
    if task == "Ingenuity":
        return OT.OracleIngenuityEnv(load_model("ingenuity"), sensor_bodies("ingenuity"), sd, p, n, seed=seed, precision="f64")
    return OT.OracleBallBalanceEnv(load_model("balance_bot"), sensor_bodies("balance_bot"), sd, p, balance_bot_dims(), n, seed=seed)


@pytest.mark.parametrize("task,n,steps", [("Ant", 512, 500), ("Humanoid", 128, 250), ("Quadcopter", 128, 500), ("Ingenuity", 128, 600), ("BallBalance", 96, 300)])
def test_long_rollout_statistics_match_the_oracle(task, n, steps):
    seed = 17
    env = _make_env(task, n, seed=seed)
    orc = _oracle(task, env, n, seed)
    g = torch.Generator(device="cpu").manual_seed(3)
    G, O = dict(rew=0.0, resets=0, vmax=0.0), dict(rew=0.0, resets=0, vmax=0.0)
    for i in range(steps):
        a = torch.rand((n, env.num_actions), generator=g) * 2 - 1
        _, rew, reset, _ = env.step(a.to(DEV))
        _, o_rew, o_reset = orc.step(a.numpy())
        G["rew"] += float(rew.mean()); G["resets"] += int(reset.sum())
        O["rew"] += float(np.mean(o_rew)); O["resets"] += int(np.sum(o_reset))
        if i % 10 == 0:
            G["vmax"] = max(G["vmax"], float(env.engine.tensors["dof_state"][..., 1].abs().max()))
            eng = orc.eng.eng if hasattr(orc.eng, "eng") else orc.eng
            O["vmax"] = max(O["vmax"], float(np.abs(eng.qd).max()))
    assert O["resets"] > 0
    # Two chaotic rollouts that have parted are independent samples: their reset counts differ like two Poisson counts, sigma = sqrt(2 N).  The band is
    # 4 % of the count (a systematic difference) plus 2.5 sigma (round 5's flat 8 % + 4 was less than ONE sigma at the Ant's ~180 resets and failed on
    # the first change of the physics that moved the chaos; the Ant now runs 512 envs, ~700 resets, so that the band is 17 % of the count)
    assert abs(G["resets"] - O["resets"]) <= 0.04 * O["resets"] + 2.5 * np.sqrt(2.0 * O["resets"]) + 2, (G, O)
    assert abs(G["rew"] - O["rew"]) <= 0.12 * abs(O["rew"]) + 0.02 * steps, (G, O)
    assert abs(G["vmax"] - O["vmax"]) <= 0.25 * O["vmax"] + 1.0, (G, O)

"""Coulomb friction known answers on the PRODUCT engines: the CPU backend here, the HIP kernels under `-m gpu`.

The scenarios and closed forms are in tests/friction_util.py (shared with tests/test_oracle_physics.py, which runs them on the fp64 oracle):
below the friction angle the robot stays put; above it every contact slides downhill and the body accelerates with g (sin theta - mu cos theta);
pushed along a diagonal of the tangent axes on level ground it stops after v0^2 / (2 mu g) on the line of the push.  Until round 6 the two
tangent rows of a contact kept their own step sizes (and t1 was applied before t2 was looked at): the Humanoid of this test was braked 16
degrees off its sliding direction, lateral acceleration 0.27-0.37 m/s^2 where the rule of core/engine.hpp friction_disc leaves < 0.02.
"""
import numpy as np
import pytest
import torch

import friction_util as F
from isaacgymenvs_amd.registry import load_model

MU = 0.5


def _env(task, n, device, mw=None):
    import isaacgymenvs_amd
    env = isaacgymenvs_amd.make(seed=3, task=task, num_envs=n, sim_device=device, rl_device=device, headless=True)
    if mw is not None:
        env.engine.set_option("multi_wave", mw)
    # contact coefficient = mean of the shapes' (per-env tensor; negative = the model's own) and the plane's (1.0): 0.5
    env.engine.tensors["friction"][:] = 2.0 * MU - float(env.sim_params.plane_mu)
    return env


def _check(out, robot, strict_stick):
    ae, th = out["slide_acc_expected"], out["slide_theta"]
    assert out["settled_speed"] < 5e-3, out
    # static friction holds below the friction angle.  (The one-sequence Gauss-Seidel order lets the PD-held Ant creep at centimetres per second -- its
    # four iterations do not converge the feet of the compliant legs; the block order of the benchmark kernels holds it to micrometres per second)
    assert out["stick_speed"] < (3e-3 if strict_stick else 0.1) and out["stick_shift"] < (0.02 if strict_stick else 0.25), out
    # sliding: a = g (sin - mu cos) along the fall line to 3 %, nothing across it (frictionless would be g sin = 2.5 x that)
    assert np.abs(out["slide_acc"] - ae).max() < 0.03 * ae, out
    assert np.abs(out["slide_acc_lateral"]).max() < 0.02 * F.G * np.sin(th), out
    assert np.ptp(out["slide_acc"]) < 1e-3 * ae                                   # every lane computes the same slide
    # stop distance v0^2 / (2 mu g) (the explicit first step and the settling of the held legs cost up to a sixth of it), on the line of the push
    de = out["stop_dist_expected"]
    assert (out["stop_dist"] > 0.80 * de).all() and (out["stop_dist"] < 1.05 * de).all(), out
    assert np.abs(out["stop_lateral"]).max() < 0.03 * de and out["stop_speed"] < 5e-3, out


@pytest.mark.parametrize("task,robot", [("Ant", "ant"), ("Humanoid", "humanoid")])
def test_friction_known_answers_cpu_backend(task, robot):
    env = _env(task, 4, "cpu")
    out = F.friction_known_answers(F.EngineRig(env), load_model(robot), robot, MU, float(env.sim_params.dt))
    _check(out, robot, strict_stick=(robot == "humanoid"))


@pytest.mark.gpu
@pytest.mark.parametrize("mw", ["auto", 0])
@pytest.mark.parametrize("task,robot", [("Ant", "ant"), ("Humanoid", "humanoid")])
def test_friction_known_answers_hip(task, robot, mw):
    """both kernel forms: the limb-per-wave kernels the benchmark sizes run (block order) and the one-wave kernels (one Gauss-Seidel sequence)"""
    env = _env(task, 64, "cuda:0", None if mw == "auto" else mw)
    out = F.friction_known_answers(F.EngineRig(env), load_model(robot), robot, MU, float(env.sim_params.dt))
    torch.cuda.synchronize()
    _check(out, robot, strict_stick=(robot == "humanoid" or mw == "auto"))

"""isaacgymenvs_amd/utils/torch_jit_utils.py (what `isaacgymenvs.utils.torch_jit_utils` and `isaacgym.torch_utils` resolve to) against
golden vectors made by the REFERENCE's own functions (tools/gen_golden_torch_utils.py, tests/golden/torch_jit_utils.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "tools"))


@pytest.fixture(scope="module")
def golden(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "torch_jit_utils.npz")))


def _names(golden):
    return sorted({k.split("__")[0] for k in golden})


def test_every_public_function_of_the_reference_module_has_a_golden_case(golden):
    from isaacgymenvs_amd.utils import torch_jit_utils as mine
    covered = set(_names(golden))
    public = {n for n in dir(mine) if not n.startswith("_") and callable(getattr(mine, n)) and n not in ("to_torch", "torch_rand_float", "torch_random_dir_2",
                                                                                                          "unscale_np", "normalise_quat_in_pose")}
    assert public <= covered, public - covered


def test_functions_match_the_reference_outputs(golden):
    """fp32: 2e-6 absolute on unit-scale outputs (the reference factorises quat_mul differently: same value, other rounding); angles are
    compared modulo 2 pi where a wrap sits at the boundary"""
    import gen_golden_torch_utils as gen  # the input table; its run() drives either module
    from isaacgymenvs_amd.utils import torch_jit_utils as mine
    out = gen.run(mine)
    assert set(out) == set(golden)
    worst = {}
    for k in sorted(golden):
        a, b = np.asarray(out[k], np.float64), np.asarray(golden[k], np.float64)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        d = np.abs(a - b)
        name = k.split("__")[0]
        if name in ("get_euler_xyz", "normalize_angle", "compute_rot", "calc_heading") and "__out" in k:
            d = np.minimum(d, np.abs(d - 2 * np.pi))
        if name in ("matrix_to_quaternion",) and "__out" in k:
            d = np.minimum(d, np.abs(a + b))        # q and -q are the same rotation
        scale = max(1.0, float(np.abs(b).max()))
        worst[k] = float(d.max()) / scale
        assert worst[k] < 5e-6, (k, worst[k])


def test_batch_dimensions_and_the_isaacgym_import_idiom():
    import isaacgymenvs_amd.shims as shims
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("isaacgym", "gym")}
    try:
        shims.install(force=True)
        ns = {}
        exec("from isaacgym.torch_utils import *", ns)          # the Isaac Gym examples' idiom
        q = ns["quat_from_angle_axis"](torch.tensor([0.5 * np.pi]), torch.tensor([[0.0, 0.0, 1.0]]))
        v = ns["quat_rotate"](q, torch.tensor([[1.0, 0.0, 0.0]]))
        assert torch.allclose(v, torch.tensor([[0.0, 1.0, 0.0]]), atol=1e-6)
        assert ns["get_axis_params"](-9.81, 2) == [0.0, 0.0, -9.81]
        assert ns["to_torch"]([1, 2], device="cpu").dtype == torch.float32
        # any number of leading dimensions
        g = torch.Generator().manual_seed(0)
        qq = torch.nn.functional.normalize(torch.randn(3, 5, 4, generator=g), dim=-1)
        vv = torch.randn(3, 5, 3, generator=g)
        back = ns["quat_rotate_inverse"](qq, ns["quat_rotate"](qq, vv))
        assert torch.allclose(back, vv, atol=1e-5)
        assert torch.allclose(ns["quat_mul"](qq, ns["quat_conjugate"](qq))[..., 3], torch.ones(3, 5), atol=1e-6)
        import isaacgymenvs.utils.torch_jit_utils as aliased
        assert aliased.quat_mul is ns["quat_mul"]
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("isaacgym", "gym")]:
            del sys.modules[k]
        sys.modules.update(saved)

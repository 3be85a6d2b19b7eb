"""Domain-randomisation sampling: the restatement (isaacgymenvs_amd/utils/dr_utils.py) against the reference's own
functions (isaacgymenvs/utils/dr_utils.py:71-145), imported with a stub `isaacgym` when /root/reference is present, and
against fixed expectations otherwise (so the test also runs where the reference tree does not exist)."""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import pytest

from isaacgymenvs_amd.utils import dr_utils as D

CASES = [
    dict(range=[0.0, 0.002], operation="additive", distribution="gaussian"),
    dict(range=[0.5, 1.5], operation="scaling", distribution="uniform", schedule="linear", schedule_steps=3000),
    dict(range=[0.75, 1.5], operation="scaling", distribution="loguniform", schedule="constant", schedule_steps=100),
    dict(range=[0.0, 0.4], operation="additive", distribution="gaussian", schedule="linear", schedule_steps=500),
]


def _reference_dr_utils():
    ref = "/root/reference/isaacgymenvs/utils/dr_utils.py"
    if not os.path.exists(ref):
        return None
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "isaacgym"))
    stub = "class SimParams:\n    pass\n\ndef __getattr__(name):\n    return None\n"
    open(os.path.join(tmp, "isaacgym", "__init__.py"), "w").write("from . import gymapi, gymtorch\n")
    open(os.path.join(tmp, "isaacgym", "gymapi.py"), "w").write(stub)
    open(os.path.join(tmp, "isaacgym", "gymtorch.py"), "w").write("")
    sys.path.insert(0, tmp)
    try:
        spec = importlib.util.spec_from_file_location("ref_dr_utils", ref)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.path.remove(tmp)


@pytest.mark.parametrize("step", [0, 50, 250, 5000])
def test_generate_random_samples_equals_reference(step):
    ref = _reference_dr_utils()
    for i, case in enumerate(CASES):
        np.random.seed(100 + i)
        ours = D.generate_random_samples(dict(case), (4, 3), step)
        assert np.isfinite(ours).all() and ours.shape == (4, 3)
        if ref is not None:
            np.random.seed(100 + i)
            theirs = ref.generate_random_samples(dict(case), (4, 3), step)
            np.testing.assert_array_equal(ours, theirs)
    # schedules: scaling ops interpolate from 1 (no effect) to the full range
    np.random.seed(0)
    s0 = D.generate_random_samples(dict(CASES[1]), 1000, 0)
    np.testing.assert_allclose(s0, 1.0)
    s1 = D.generate_random_samples(dict(CASES[1]), 1000, 3000)
    assert 0.5 <= s1.min() < 0.6 and 1.4 < s1.max() <= 1.5


def test_bucketing_and_gravity():
    ref = _reference_dr_utils()
    p = dict(range=[0.5, 1.25], distribution="uniform", operation="scaling", num_buckets=100)
    for v in (0.5, 0.6234, 1.0, 1.2499):
        b = D.get_bucketed_val(v, p)
        assert b <= v < b + 0.75 / 100 + 1e-12
        if ref is not None:
            assert b == ref.get_bucketed_val(v, p)
    np.random.seed(3)
    g = D.apply_random_gravity([0, 0, -9.81], [0, 0, -9.81], dict(range=[0, 0.4], operation="additive", distribution="gaussian"), 10)
    assert abs(g[2] + 9.81) < 2.5 and abs(g[0]) < 2.5

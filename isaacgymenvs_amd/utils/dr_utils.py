"""Domain-randomisation sampling (host side, NumPy) -- restates reference isaacgymenvs/utils/dr_utils.py:71-208.

Same distributions, schedules and operation semantics, same use of NumPy's global RNG (one draw per call site), so that
under `np.random.seed(s)` the samples equal the reference's (tests/test_dr_utils.py checks this against the reference's
own functions).  Property objects are plain dicts / attribute holders here instead of gymapi structs.
"""
from __future__ import annotations

from bisect import bisect

import numpy as np


def _sched_scaling(p, step):  # dr_utils.py:78-90
    sched_type = p["schedule"] if "schedule" in p else None
    sched_step = p["schedule_steps"] if "schedule" in p else None
    if sched_type == "linear":
        return 1 / sched_step * min(step, sched_step)
    if sched_type == "constant":
        return 0 if step < sched_step else 1
    return 1


def generate_random_samples(attr_randomization_params, shape, curr_gym_step_count, extern_sample=None):
    """dr_utils.py:71-132"""
    rand_range = attr_randomization_params["range"]
    distribution = attr_randomization_params["distribution"]
    operation = attr_randomization_params["operation"]
    s = _sched_scaling(attr_randomization_params, curr_gym_step_count)
    if extern_sample is not None:
        sample = extern_sample
        if operation == "additive":
            sample *= s
        elif operation == "scaling":
            sample = sample * s + 1 * (1 - s)
    elif distribution == "gaussian":
        mu, var = rand_range
        if operation == "additive":
            mu *= s
            var *= s
        elif operation == "scaling":
            var = var * s
            mu = mu * s + 1 * (1 - s)
        sample = np.random.normal(mu, var, shape)
    elif distribution == "loguniform":
        lo, hi = rand_range
        if operation == "additive":
            lo *= s
            hi *= s
        elif operation == "scaling":
            lo = lo * s + 1 * (1 - s)
            hi = hi * s + 1 * (1 - s)
        sample = np.exp(np.random.uniform(np.log(lo), np.log(hi), shape))
    elif distribution == "uniform":
        lo, hi = rand_range
        if operation == "additive":
            lo *= s
            hi *= s
        elif operation == "scaling":
            lo = lo * s + 1 * (1 - s)
            hi = hi * s + 1 * (1 - s)
        sample = np.random.uniform(lo, hi, shape)
    else:
        raise ValueError(f"unknown distribution {distribution}")
    return sample


def get_bucketed_val(new_prop_val, attr_randomization_params):
    """dr_utils.py:135-145"""
    if attr_randomization_params["distribution"] == "uniform":
        lo, hi = attr_randomization_params["range"][0], attr_randomization_params["range"][1]
    else:
        lo = attr_randomization_params["range"][0] - 2 * np.sqrt(attr_randomization_params["range"][1])
        hi = attr_randomization_params["range"][0] + 2 * np.sqrt(attr_randomization_params["range"][1])
    num_buckets = attr_randomization_params["num_buckets"]
    buckets = [(hi - lo) * i / num_buckets + lo for i in range(num_buckets)]
    if np.ndim(new_prop_val) == 0:
        return buckets[bisect(buckets, new_prop_val) - 1]
    # array form of the same lookup (bisect == bisect_right; index -1 wraps to the last bucket exactly like the scalar expression)
    return np.asarray(buckets)[np.searchsorted(buckets, np.asarray(new_prop_val), side="right") - 1]


def apply_random_gravity(gravity, og_gravity, attr_randomization_params, curr_gym_step_count):
    """SimParams branch of apply_random_samples for attr == 'gravity' (dr_utils.py:160-172).  gravity: 3 floats."""
    sample = generate_random_samples(attr_randomization_params, 3, curr_gym_step_count)
    if attr_randomization_params["operation"] == "scaling":
        return [og_gravity[i] * sample[i] for i in range(3)]
    if attr_randomization_params["operation"] == "additive":
        return [og_gravity[i] + sample[i] for i in range(3)]
    return list(gravity)


def apply_random_samples_array(prop, og_prop, attr, attr_randomization_params, curr_gym_step_count, extern_sample=None):
    """ndarray branch of apply_random_samples (dr_utils.py:181-192); prop / og_prop: dict-like of arrays."""
    sample = generate_random_samples(attr_randomization_params, np.shape(prop[attr]), curr_gym_step_count, extern_sample)
    if attr_randomization_params["operation"] == "scaling":
        new_prop_val = og_prop[attr] * sample
    elif attr_randomization_params["operation"] == "additive":
        new_prop_val = og_prop[attr] + sample
    if "num_buckets" in attr_randomization_params and attr_randomization_params["num_buckets"] > 0:
        new_prop_val = get_bucketed_val(new_prop_val, attr_randomization_params)
    prop[attr] = new_prop_val
    return new_prop_val

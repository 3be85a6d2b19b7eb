"""Domain-randomisation sampling on the host (NumPy).

What the reference specifies (isaacgymenvs/utils/dr_utils.py:71-208 with the `randomization_params` blocks of the task YAMLs): a
parameter is re-drawn from `distribution` in {gaussian, uniform, loguniform} over `range`, combined with its original value by
`operation` in {additive, scaling}, faded in by an optional `schedule` (linear over / constant after `schedule_steps`), optionally
snapped to `num_buckets` values.  The draws come from NumPy's global generator, one call per parameter, so a run seeded with
`np.random.seed(s)` reproduces the reference's values (tests/test_dr_utils.py checks that against the reference's own functions).

Organisation here: a distribution is described once (how `range` turns into its two shape parameters, how the schedule blends them
towards "no effect", how to draw), and every entry point goes through `Draw`."""
from __future__ import annotations

import numpy as np


def schedule_weight(params, step):
    """0 -> the randomisation has no effect yet, 1 -> full range (dr_utils.py:78-90)."""
    kind = params.get("schedule")
    if kind == "linear":
        return min(step, params["schedule_steps"]) / params["schedule_steps"]
    if kind == "constant":
        return 1.0 if step >= params["schedule_steps"] else 0.0
    return 1.0


class Draw:
    """One randomised parameter: distribution, operation and the schedule weight resolved into the two numbers NumPy needs."""

    #: distribution -> (which of the two `range` entries shift with a scaling schedule, NumPy draw)
    _KINDS = {
        "gaussian": ((0,), lambda a, b, shape: np.random.normal(a, b, shape)),              # range = (mean, std)
        "uniform": ((0, 1), lambda a, b, shape: np.random.uniform(a, b, shape)),             # range = (low, high)
        "loguniform": ((0, 1), lambda a, b, shape: np.exp(np.random.uniform(np.log(a), np.log(b), shape))),
    }

    def __init__(self, params, step):
        if params["distribution"] not in self._KINDS:
            raise ValueError(f"unknown distribution {params['distribution']}")
        self.op = params["operation"]
        self.weight = schedule_weight(params, step)
        moved, self._draw = self._KINDS[params["distribution"]]
        pair = [params["range"][0], params["range"][1]]
        if self.op == "additive":                      # additive noise fades in from zero: both numbers shrink with the weight
            pair = [v * self.weight for v in pair]
        elif self.op == "scaling":                     # a scale factor fades in from one: locations move towards 1, spreads shrink
            pair = [v * self.weight + (1 - self.weight) if i in moved else v * self.weight for i, v in enumerate(pair)]
        self.a, self.b = pair

    def sample(self, shape):
        return self._draw(self.a, self.b, shape)

    def blend_external(self, sample):
        """an externally supplied sample goes through the same schedule (dr_utils.py:96-101)"""
        if self.op == "additive":
            return sample * self.weight
        if self.op == "scaling":
            return sample * self.weight + (1 - self.weight)
        return sample

    def combine(self, original, sample):
        if self.op == "scaling":
            return original * sample
        if self.op == "additive":
            return original + sample
        raise ValueError(f"unknown operation {self.op}")


def generate_random_samples(attr_randomization_params, shape, curr_gym_step_count, extern_sample=None):
    d = Draw(attr_randomization_params, curr_gym_step_count)
    return d.blend_external(extern_sample) if extern_sample is not None else d.sample(shape)


def bucket_edges(params):
    """`num_buckets` equally spaced values from the low end of the parameter's range: the range itself for uniform draws, mean +- 2
    sqrt(second entry) for gaussian ones (dr_utils.py:135-145)."""
    a, b = params["range"][0], params["range"][1]
    lo, hi = (a, b) if params["distribution"] == "uniform" else (a - 2 * np.sqrt(b), a + 2 * np.sqrt(b))
    n = params["num_buckets"]
    return np.array([(hi - lo) * i / n + lo for i in range(n)])


def get_bucketed_val(new_prop_val, attr_randomization_params):
    """Snap to the largest bucket value not above the argument; below the first bucket this wraps to the LAST one -- the reference's
    `buckets[bisect(buckets, x) - 1]` with index -1 -- which is kept."""
    edges = bucket_edges(attr_randomization_params)
    idx = np.searchsorted(edges, np.asarray(new_prop_val), side="right") - 1
    out = edges[idx]
    return float(out) if np.ndim(new_prop_val) == 0 else out


def apply_random_gravity(gravity, og_gravity, attr_randomization_params, curr_gym_step_count):
    """`sim_params.gravity` (dr_utils.py:160-172): three draws combined with the original vector."""
    d = Draw(attr_randomization_params, curr_gym_step_count)
    if d.op not in ("scaling", "additive"):
        return list(gravity)
    s = d.sample(3)
    return [d.combine(og_gravity[i], s[i]) for i in range(3)]


def apply_random_samples_array(prop, og_prop, attr, attr_randomization_params, curr_gym_step_count, extern_sample=None):
    """Array-valued property (dr_utils.py:181-192): one draw of the property's shape, combined, bucketed, written back."""
    d = Draw(attr_randomization_params, curr_gym_step_count)
    s = d.blend_external(extern_sample) if extern_sample is not None else d.sample(np.shape(prop[attr]))
    val = d.combine(og_prop[attr], s)
    if attr_randomization_params.get("num_buckets", 0) > 0:
        val = get_bucketed_val(val, attr_randomization_params)
    prop[attr] = val
    return val

"""Tiny stand-in for `gym.spaces.Box` (the `gym` package is not part of this image).

The reference only reads `.shape`, `.low`, `.high`, `.dtype` of these (vec_task.py:104-113,
utils/rlgames_utils.py:262-280).  If `gym`/`gymnasium` is importable its Box is used instead.
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    from gym.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001
    try:
        from gymnasium.spaces import Box  # type: ignore
    except Exception:  # noqa: BLE001
        class Box:  # minimal API-compatible box
            def __init__(self, low, high, shape=None, dtype=np.float32):
                low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
                if shape is not None:
                    low, high = np.broadcast_to(low, shape).copy(), np.broadcast_to(high, shape).copy()
                self.low, self.high, self.shape, self.dtype = low, high, low.shape, np.dtype(dtype)

            def sample(self):
                lo = np.where(np.isfinite(self.low), self.low, -1.0)
                hi = np.where(np.isfinite(self.high), self.high, 1.0)
                return np.random.uniform(lo, hi).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

try:  # pragma: no cover - depends on the environment
    from gym.spaces import Dict  # type: ignore
except Exception:  # noqa: BLE001
    try:
        from gymnasium.spaces import Dict  # type: ignore
    except Exception:  # noqa: BLE001
        class Dict:  # the dictionary observation space of the reference's dextreme tasks (tasks/dextreme/adr_vec_task.py:76-84): name -> Box
            def __init__(self, spaces=None, **kw):
                self.spaces = dict(spaces or {}, **kw)

            def __getitem__(self, k):
                return self.spaces[k]

            def __setitem__(self, k, space):
                self.spaces[k] = space

            def __iter__(self):
                return iter(self.spaces)

            def __len__(self):
                return len(self.spaces)

            def keys(self):
                return self.spaces.keys()

            def items(self):
                return self.spaces.items()

            def values(self):
                return self.spaces.values()

            def sample(self):
                return {k: s.sample() for k, s in self.spaces.items()}

            def contains(self, x):
                return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(s.contains(x[k]) for k, s in self.spaces.items())

            def __repr__(self):
                return "Dict(" + ", ".join(f"{k}: {s!r}" for k, s in self.spaces.items()) + ")"

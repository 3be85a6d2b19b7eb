"""Process-wide seeding for a ROCm job (what reference isaacgymenvs/utils/utils.py:87-115 does for CUDA)."""
import os
import random

import numpy as np
import torch


def set_seed(seed, torch_deterministic=False, rank=0):
    """Seed python / NumPy / torch (host and every visible ROCm device) with `seed + rank` and return the value used.

    `seed == -1` asks for a fresh seed (a fixed 42 + rank under `torch_deterministic`, as the reference does).  The engine's own
    reset RNG is not touched here: it is a counter-based hash of (engine seed, global env id, episode, draw) inside the kernels
    and is seeded through `make(seed=...)`.  Deterministic mode only concerns the torch ops around the engine (policy network):
    MIOpen's auto-tuner is switched off and deterministic algorithms are requested; hipBLASLt needs no workspace variable.
    """
    if seed == -1:
        seed = 42 + rank if torch_deterministic else int(np.random.randint(0, 10000))
    else:
        seed = int(seed) + rank
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)                  # seeds the host generator and, lazily, the generator of every ROCm device
    det = bool(torch_deterministic)
    torch.backends.cudnn.benchmark = not det           # MIOpen find-mode auto-tuning (torch keeps the cudnn name on ROCm)
    torch.backends.cudnn.deterministic = det
    torch.use_deterministic_algorithms(det, warn_only=True)
    return seed

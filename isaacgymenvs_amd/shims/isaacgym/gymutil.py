"""`isaacgym.gymutil`: only imported, never used on the paths the shim covers."""


def parse_arguments(*args, **kwargs):
    raise NotImplementedError("gymutil.parse_arguments: the reference's train.py uses Hydra; not part of the shim")

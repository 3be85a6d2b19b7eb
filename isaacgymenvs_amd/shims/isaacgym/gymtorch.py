"""`isaacgym.gymtorch`: tensor descriptors are the torch tensors themselves."""


def wrap_tensor(desc):
    return desc


def unwrap_tensor(tensor):
    return tensor

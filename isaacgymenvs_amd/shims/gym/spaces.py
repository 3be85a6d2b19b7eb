from ...utils.spaces import Box  # noqa: F401

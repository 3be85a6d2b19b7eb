from ...utils.spaces import Box, Dict  # noqa: F401

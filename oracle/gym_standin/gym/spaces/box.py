from . import Box  # noqa: F401

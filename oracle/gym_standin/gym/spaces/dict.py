from . import Dict  # noqa: F401

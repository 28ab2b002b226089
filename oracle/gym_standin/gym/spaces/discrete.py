from . import Discrete  # noqa: F401

from . import functional, pytorch  # noqa: F401

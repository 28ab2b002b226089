"""Minimal stand-in for the ``gym.spaces`` symbols the reference's envs import (mubs_cov.py:1-3, env_wrappers.py:7).
Test infrastructure for golden generation in the build container only; NOT gym."""
from . import spaces  # noqa: F401

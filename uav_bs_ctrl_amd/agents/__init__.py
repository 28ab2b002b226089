"""Agent registry, mirroring /root/reference/algos/madrqn/agents/__init__.py:1-7 (``'gnn'`` entry) and
/root/reference/algos/drqn/agents/__init__.py (``'drqn_gnn'``)."""
REGISTRY = {}

from .gnn_agents import DrqnGnnAgent, GnnAgent  # noqa: E402

REGISTRY["gnn"] = GnnAgent
REGISTRY["drqn_gnn"] = DrqnGnnAgent

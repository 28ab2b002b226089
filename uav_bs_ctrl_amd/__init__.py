"""uav_bs_ctrl_amd - MI355X-native (gfx950) hetero-GNN hot path of zhangxiaochen95/uav_bs_ctrl's MADRQN agent.

Drop-in for the reference's ``REGISTRY['gnn']`` (algos/madrqn/agents/__init__.py:1-7): same constructor, same
``forward(g, h) -> (q, h')``, same ``state_dict`` layout; the graph arithmetic runs in hand-written HIP kernels behind
the C-ABI of ``include/uavgnn.h``.  There is no CPU fallback.
"""
from .agents import REGISTRY, GnnAgent  # noqa: F401
from .tuned import enable_tuned_gemms  # noqa: F401  (opt-in, process-wide: see uav_bs_ctrl_amd/tuned)
from .graph import HeteroBatch, batch, cat, from_obs_dicts, from_padded_obs, heterograph, merge  # noqa: F401

__version__ = "0.1.0"

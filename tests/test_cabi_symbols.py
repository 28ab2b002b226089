"""The C-ABI library loads and exports every symbol include/uavgnn.h (the drop-in boundary) and include/uavgnn_probe.h (probe-only
building blocks of tools/) declare (no compute calls without a GPU)."""
import ctypes
import os
import re

from uav_bs_ctrl_amd import _lib
from uav_bs_ctrl_amd.build import build_lib


def _declared(root, header="uavgnn.h"):
    src = open(os.path.join(root, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uavgnn_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol(repo_root):
    path = build_lib()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    public, probe = _declared(repo_root), _declared(repo_root, "uavgnn_probe.h")
    assert probe and not set(public) & set(probe), "a symbol is declared in both headers"
    assert not any("_dbg" in n for n in public), "debug entries belong to include/uavgnn_probe.h"
    # the product (everything under uav_bs_ctrl_amd/ but the ctypes table) calls the public header only
    pkg = os.path.join(repo_root, "uav_bs_ctrl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "_lib.py":
                text = open(os.path.join(dirpath, f)).read()
                used = [n for n in probe if re.search(r"\b" + n + r"\b", text)]
                assert not used, f"{os.path.join(dirpath, f)} calls probe-only entries {used}"
    names = sorted(public + probe)
    assert "uavgnn_gatv2_fwd" in names and "uavgnn_talk_attn_bwd" in names
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/uavgnn.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table out of sync with the header"
    assert _lib.lib().uavgnn_version() == 100
    assert b"instantiations" in _lib.lib().uavgnn_strerror(-1001)


def test_argument_errors_are_codes_not_crashes():
    L = _lib.lib()
    assert L.uavgnn_gru_gates_fwd(None, None, None, 4, 8, None, None) == -1000
    assert L.uavgnn_gatv2_fwd(None, 0, 4, None, 2, None, None, 3, None, None, None, None, None, None, None, 4, 64, 0.2, None,
                              256, None, None) == -1000
    assert L.uavgnn_gatv2_bwd_workspace_bytes(4, 256) == 1024 * 256 * 12 * 4

"""The plain-C scalar restatement (oracle/c/gatv2_ref.c) agrees with the torch segment restatement and, through it,
with the fixtures generated from the reference's own modules."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch as th

from oracle import restatement as R
from tests.util import assert_close, load_golden

CDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "c")


@pytest.fixture(scope="module")
def cref():
    subprocess.run(["make", "-s", "-C", CDIR], check=True)
    lib = ctypes.CDLL(os.path.join(CDIR, "libgatv2_ref.so"))
    lib.gatv2_ref_forward.restype = ctypes.c_int
    return lib


def _dp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("rel,xk,ok", [("seen", "x_gt", "seen_off"), ("near", "x_ubs", "near_off")])
def test_c_restatement_matches_segment_restatement_on_the_ragged_fixture(cref, rel, xk, ok):
    g, _, p, cfg, _ = load_golden("agent_tarmac")
    pr = R.sub(p, f"enc.f_conv.{rel}")
    x_src, x_dst, off = g[xk].numpy().copy(), g["x_a"].numpy().copy(), g[ok].numpy().astype(np.int32).copy()
    N, nh = x_dst.shape[0], cfg["n_heads"]
    H = pr["fc_src.weight"].shape[0]
    arrs = [pr[k].numpy().reshape(-1).copy() for k in ("fc_src.weight", "fc_src.bias", "fc_dst.weight", "fc_dst.bias",
                                                       "attn", "res_fc.weight", "res_fc.bias")]
    out = np.zeros((N, H))
    rc = cref.gatv2_ref_forward(_dp(x_src), x_src.shape[1], _dp(x_dst), 2, _dp(off), N, *[_dp(a) for a in arrs], nh,
                                H // nh, ctypes.c_double(0.2), _dp(out))
    assert rc == 0
    ref = R.gatv2_conv_seg(g[xk], g["x_a"], g[ok], pr, nh).reshape(N, H)
    assert_close(th.as_tensor(out), ref, 1e-12, f"C vs torch restatement ({rel})")

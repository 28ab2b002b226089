"""Shared helpers for the test-suite (fixture loading, tolerance rule)."""
import ast
import math
import os

import numpy as np
import torch as th

from oracle.closed_form import closed_form_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GRAPH_KEYS = ("x_a", "x_gt", "seen_off", "x_ubs", "near_off", "talk_off", "talk_src", "talk_eid", "x_flat")


def load_golden(name, dtype=th.float64, device="cpu"):
    """-> (graph dict, h, params(state_dict, closed form), cfg, npz)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {}
    for k in GRAPH_KEYS:
        if k in z.files:
            a = th.as_tensor(z[k])
            g[k] = (a.to(dtype) if a.is_floating_point() else a).to(device)
    cfg = ast.literal_eval(str(z["cfg"]))
    params = {}
    for i, (n, s) in enumerate(zip(z["param_names"], z["param_shapes"])):
        shape = ast.literal_eval(str(s))
        amp = 0.1 if len(shape) == 1 else 0.25
        params[str(n)] = closed_form_tensor(shape, 1.0 + i * math.pi / 7, amp, th.float64).to(dtype).to(device)
    h = th.as_tensor(z["h"]).to(dtype).to(device)
    return g, h, params, cfg, z


def assert_close(actual, ref, rel=1e-5, what="", floor=0.0):
    """The parity rule of BASELINE.md section 4: rtol = rel, atol = rel * max|ref|.

    ``floor`` is an absolute floor on atol for tensors that are analytically zero (e.g. d/d f_sign.bias: a constant
    added to every signature cancels in the softmax), where max|ref| is itself rounding noise."""
    actual = actual.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert actual.shape == ref.shape, f"{what}: shape {tuple(actual.shape)} vs {tuple(ref.shape)}"
    atol = max(rel * (float(ref.abs().max()) if ref.numel() else 0.0), floor, 1e-30)
    err = (actual - ref).abs()
    bound = atol + rel * ref.abs()
    bad = err > bound
    if bool(bad.any()):
        i = int(th.argmax(err - bound))
        raise AssertionError(f"{what}: {int(bad.sum())}/{ref.numel()} out of tolerance (rel={rel}); worst err "
                             f"{float(err.flatten()[i]):.3e} at flat {i}: got {float(actual.flatten()[i]):.8e} "
                             f"ref {float(ref.flatten()[i]):.8e}; max|ref|={float(ref.abs().max()):.3e}")


def load_learner_golden(name="learner_update_tarmac", dtype=th.float64, device="cpu"):
    """-> (batch dict with obs as list of segment-array dicts, closed-form params, cfg, npz)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = ast.literal_eval(str(z["cfg"]))
    obs = []
    for t in range(cfg["T"] + 1):
        g = {}
        for k in GRAPH_KEYS:
            key = f"t{t}:{k}"
            if key in z.files:
                a = th.as_tensor(z[key])
                g[k] = (a.to(dtype) if a.is_floating_point() else a).to(device)
        obs.append(g)
    params = {}
    for i, (n, s) in enumerate(zip(z["param_names"], z["param_shapes"])):
        shape = ast.literal_eval(str(s))
        amp = 0.1 if len(shape) == 1 else 0.25
        params[str(n)] = closed_form_tensor(shape, 1.0 + i * math.pi / 7, amp, th.float64).to(dtype).to(device)
    f = lambda k: th.as_tensor(z[k]).to(dtype).to(device)  # noqa: E731
    batch = dict(obs=obs, h0=f("h0"), h1=f("h1"), acts=th.as_tensor(z["acts"]).long().to(device), rews=f("rews"),
                 dones=f("dones"))
    return batch, params, cfg, z

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


# ---------------------------------------------------------------------------------------------------------------------
# Gradient rule (north_star: 1e-5 relative fp32).  A gradient is a long fp32 reduction over edges / agents / time steps,
# so fp32 arithmetic itself - in ANY order - may sit above 1e-5 of the float64 value.  The bound is therefore
#     max(1e-5, 4 x the float32 CPU oracle's own error against the float64 oracle on the same inputs)
# measured in the same mixed norm as assert_close, and every comparison is RECORDED (measured error, the fp32 oracle's
# error, the bound) into gpurun_out/grad_errors.jsonl so that the table under profiles/ says how large the errors are.
GRAD_BASE = 1e-5
GRAD_FACTOR = 4.0
_GRAD_LOG = os.path.join(os.path.dirname(GOLDEN), os.pardir, "gpurun_out", "grad_errors.jsonl")


def needed_rel(actual, ref, floor=0.0):
    """Smallest `rel` for which assert_close(actual, ref, rel, floor=floor) passes: max_i err_i / (max|ref| + |ref_i|)
    over the elements whose error exceeds `floor`."""
    actual = actual.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert actual.shape == ref.shape, f"shape {tuple(actual.shape)} vs {tuple(ref.shape)}"
    if ref.numel() == 0:
        return 0.0
    err = (actual - ref).abs()
    err = th.where(err > floor, err, th.zeros_like(err))
    den = float(ref.abs().max()) + ref.abs()
    if float(den.max()) == 0.0:
        return 0.0 if float(err.max()) == 0.0 else float("inf")
    return float((err / den.clamp_min(1e-300)).max())


def grad_close(actual, ref64, what, ref32=None, floor=0.0, bound=None):
    """Assert a gradient against the float64 oracle under the rule above and record the measurement.  ref32: the same
    gradient from the float32 CPU oracle (None: the bound is the plain 1e-5, or `bound` when the caller states one)."""
    import json
    got = needed_rel(actual, ref64, floor)
    cpu32 = None if ref32 is None else needed_rel(ref32, ref64, floor)
    lim = bound if bound is not None else max(GRAD_BASE, GRAD_FACTOR * (cpu32 or 0.0))
    try:
        os.makedirs(os.path.dirname(_GRAD_LOG), exist_ok=True)
        with open(_GRAD_LOG, "a") as f:
            f.write(json.dumps(dict(test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], what=what,
                                    rel_err=got, cpu_fp32_rel_err=cpu32, bound=lim, floor=floor,
                                    max_abs_ref=float(ref64.detach().abs().max()) if ref64.numel() else 0.0)) + "\n")
    except OSError:
        pass
    assert got <= lim, (f"{what}: gradient rel err {got:.3e} > bound {lim:.3e} "
                        f"(fp32 CPU oracle's own error vs float64: {cpu32 if cpu32 is None else format(cpu32, '.3e')})")
    return got

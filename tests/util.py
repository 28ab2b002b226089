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
    bad = (err > bound) | (th.isnan(err) & ~th.isnan(ref))      # a NaN where the reference is a number is a failure, not "not greater"
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
# Gradient rule (north_star: 1e-5 relative fp32).  A gradient is a long fp32 reduction over edges / agents / time steps
# with cancellation, so fp32 arithmetic itself - in ANY summation order - may sit above 1e-5 of the float64 value (and a
# gradient that is analytically zero, e.g. d/d f_sign.bias, is pure rounding noise on both sides).  Element i passes when
#     |got_i - ref64_i|  <=  max( 1e-5 (max|ref64| + |ref64_i|),  4 E32 ),
# E32 = the largest absolute error of the float32 CPU oracle against the float64 oracle on the same tensor: 1e-5 wherever
# fp32 can hold it, otherwise no worse than 4x what fp32 arithmetic itself delivers.  No blanket floor.  Every comparison
# is RECORDED (measured relative error, the fp32 oracle's, which clause decided) in gpurun_out/grad_errors.jsonl; the table
# under profiles/ is made from it by tools/grad_error_table.py.
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


def grad_close(actual, ref64, what, ref32=None, floor=0.0, base=None):
    """Assert a gradient against the float64 oracle under the rule above and record the measurement.  ref32: the same
    gradient from the float32 CPU oracle; `floor`: an explicit absolute floor for comparisons that have no fp32 oracle; `base`: a
    relative bound other than 1e-5 for a comparison whose call site states why (recorded as such)."""
    import json
    base = GRAD_BASE if base is None else base
    a = actual.detach().double().cpu()
    r = ref64.detach().double().cpu()
    assert a.shape == r.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(r.shape)}"
    raw = needed_rel(a, r)
    e32 = cpu32 = None
    if ref32 is not None:
        r32 = ref32.detach().double().cpu()
        e32 = float((r32 - r).abs().max()) if r.numel() else 0.0
        cpu32 = needed_rel(r32, r)
    abs_floor = max(floor, GRAD_FACTOR * (e32 or 0.0))
    err = (a - r).abs()
    lim = th.clamp(base * (float(r.abs().max()) + r.abs()) if r.numel() else err, min=abs_floor)
    ok = bool((err <= lim).all()) if r.numel() else True
    try:
        os.makedirs(os.path.dirname(_GRAD_LOG), exist_ok=True)
        with open(_GRAD_LOG, "a") as f:
            f.write(json.dumps(dict(test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], what=what, rel_err=raw,
                                    cpu_fp32_rel_err=cpu32, max_abs_err=float(err.max()) if r.numel() else 0.0,
                                    cpu_fp32_max_abs_err=e32, abs_floor=abs_floor,
                                    max_abs_ref=float(r.abs().max()) if r.numel() else 0.0,
                                    decided_by=("1e-5" if raw <= GRAD_BASE else f"stated bound {base:g}" if (ok and raw <= base) else
                                                "4x fp32 oracle error" if ok else "FAIL"))) + "\n")
    except OSError:
        pass
    assert ok, (f"{what}: gradient rel err {raw:.3e} (max abs {float(err.max()):.3e}) exceeds max({base:g} relative, "
                f"{abs_floor:.3e} absolute = 4x the fp32 CPU oracle's own error {e32})")
    return raw


def wait_worker(pr, q, timeout=420.0):
    """Result of a spawned worker that reports through ``q``: returns as soon as it arrives, and FAILS as soon as the worker is gone
    without having reported (a worker killed by a native abort would otherwise keep the caller waiting for the whole timeout)."""
    import queue as _queue
    import time
    deadline = time.monotonic() + timeout
    while True:
        try:
            return q.get(timeout=2.0)
        except _queue.Empty:
            pass
        if not pr.is_alive():
            try:
                return q.get(timeout=2.0)            # it may have reported just before it exited
            except _queue.Empty:
                raise AssertionError(f"worker exited with code {pr.exitcode} without reporting") from None
        if time.monotonic() > deadline:
            raise AssertionError(f"worker did not report within {timeout:.0f} s")

"""``torch.autograd.Function`` wrappers over the C-ABI kernels (include/uavgnn.h).

Each op checks dtype / contiguity / device, takes PyTorch's current HIP stream and calls ``libuavgnn.so`` through
ctypes.  No op has a CPU or eager-PyTorch fallback: CPU tensors raise ``UavGnnError``.
"""
from __future__ import annotations

import os
import torch as th

from . import _lib as L

NEG_SLOPE = 0.2  # DGL GATv2Conv default, not overridden at gnn_agents.py:93-96
K1_IMAGE = os.environ.get("UAVGNN_K1_IMAGE", "1") != "0"   # prepared parameter image inside frozen_weights() scopes (A/B switch)
HETERO_FUSED = True   # K1 forward of both encoder relations in one launch (csrc/gatv2_hetero.hip); False: per relation
# K1's score GEMM on the bf16 matrix cores (exact three-way splits: fp32-level accuracy); False: the fp32-MFMA build of the same
# kernel (csrc/gatv2_hetero_f32.hip) - the A/B reference and the K1 part of bench.py's strict-fp32 leg
K1_BF16Z = os.environ.get("UAVGNN_K1_BF16Z", "1") != "0"
# row maxima out of the time-batched K1 launches (more than K1_ROWMAX_MIN_ROWS destinations) for an f16x2 f_aggr forward (A/B switch)
K1_ROWMAX = os.environ.get("UAVGNN_K1_ROWMAX", "1") != "0"
K1_ROWMAX_MIN_ROWS = 1 << 17
GEMM_H2_RM2 = os.environ.get("UAVGNN_GEMM_H2_RM2", "1") != "0"   # d x of the recurrent step takes the row maxima of d_proj inside its launch (A/B switch: a pass of uavgnn_row_absmax)
# ... and out of ANY launch whose `seen` relation has this mean in-degree (a compute-bound launch) from the row count at which the f_aggr
# product behind it takes the f16x2 kernel (GEMM_X3_SMALL_GRID tiles of 256 x 128 for H = 256 output columns)
K1_ROWMAX_DENSE_DEG = int(os.environ.get("UAVGNN_K1_ROWMAX_DENSE_DEG", "16"))
K1_ROWMAX_DENSE_MIN_ROWS = 16384


class _KernelTimer:
    """Optional HIP-event timing of the C-ABI launches (bench.py's live roofline measurement).  Events are recorded
    on PyTorch's current stream, which is the stream every launch is issued on."""

    def __init__(self):
        self.enabled = False
        self.only = None
        self._spans = {}

    def reset(self, enabled=True, only=None):
        """only: names of the spans to time (None = all).  A span costs two event creations + records on the launch
        thread (~15 us) and two timestamp packets on the stream: timing EVERY launch of a cycle makes its rollout phase
        host-bound (tools/launch_bound_probe.py), so a timed benchmark region instruments the kernel it grades only."""
        self._spans = {}
        self.enabled = enabled
        self.only = None if only is None else frozenset(only)

    class _Span:
        def __init__(self, timer, name, work=None):
            self.t, self.name, self.work = timer, name, work

        def __enter__(self):
            self.on = self.t.enabled and (self.t.only is None or self.name in self.t.only)
            if self.on:
                self.e0 = th.cuda.Event(enable_timing=True)
                self.e1 = th.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.on:
                self.e1.record()
                self.t._spans.setdefault(self.name, []).append((self.e0, self.e1, self.work))
            return False

    def span(self, name, work=None):
        """work: optional tuple describing the launch (edges, destinations, saves-attention flag) for byte accounting."""
        return _KernelTimer._Span(self, name, work)

    def summary(self):
        th.cuda.synchronize()
        out = {}
        for name, pairs in self._spans.items():
            ms = [a.elapsed_time(b) for a, b, _ in pairs]
            out[name] = dict(count=len(ms), avg_ms=sum(ms) / len(ms), total_ms=sum(ms), ms=ms,
                             work=[w for _, _, w in pairs if w is not None])
        return out


KERNEL_TIMER = _KernelTimer()


class _HeteroGATv2(th.autograd.Function):
    """K1 over R relations that share the destination nodes.  Returns [N, R*H]: relation i owns columns [i*H, (i+1)*H)
    (so the th.cat of gnn_agents.py:106 never happens).  Per relation the flat argument list carries
    x_src, seg_off, dst_order, attn, W_s, b_s, W_d, b_d, W_r, b_r  (b_r and dst_order may be None)."""

    PER_REL = 10
    last_rowmax = None      # (address of the output, rowmax_near, rowmax_seen) of the forward that just ran, for hetero_gatv2()

    @staticmethod
    def forward(ctx, x_dst, nh, train, *rel_args):
        R = len(rel_args) // _HeteroGATv2.PER_REL
        L.require_gpu(x_dst, *[t for t in rel_args if isinstance(t, th.Tensor)])
        x_dst = L.f32c(x_dst)
        N = x_dst.shape[0]
        H = rel_args[4].shape[0]
        D = H // nh
        out = th.empty((N, R * H), dtype=th.float32, device=x_dst.device)
        saved, meta, has_order = [x_dst], [], []
        # both relations of the observation encoder in ONE launch when the shape has a fused instantiation
        fused = (HETERO_FUSED and R == 2 and N > 0 and rel_args[4].shape[1] == 4 and rel_args[14].shape[1] == 2 and
                 bool(L.lib().uavgnn_gatv2_hetero_supported(4, 2, x_dst.shape[1], nh, D)))
        prepared = []
        for i in range(R):
            x_src, seg_off, order, attn, W_s, b_s, W_d, b_d, W_r, b_r = rel_args[i * 10:(i + 1) * 10]
            x_src = L.f32c(x_src)
            FS = W_s.shape[1]
            if W_s.shape[0] != H or x_src.shape[0] and x_src.shape[1] != FS:
                raise L.UavGnnError("hetero_gatv2: inconsistent relation shapes")
            p = [L.f32c(t.detach()) for t in (W_s, b_s, W_d, b_d, attn, W_r)]
            b_r_c = None if b_r is None else L.f32c(b_r.detach())
            # ctx.needs_input_grad stays True for parameters under torch.no_grad(): `train` (grad mode at the call site)
            # decides whether the attention weights are saved - a rollout / target-network forward must not pay for them
            need = train and any(ctx.needs_input_grad[3 + i * 10 + 3: 3 + i * 10 + 10])
            a_save = th.empty((max(x_src.shape[0], 1), nh), dtype=th.float32, device=x_dst.device) if need else None
            prepared.append((x_src, seg_off, order, p, b_r_c, need, a_save, FS, b_r is not None))
            fused = fused and all(t.data_ptr() % 16 == 0 for t in p + ([b_r_c] if b_r_c is not None else []))
        if fused:
            (xs, so, oo, pS, brS, needS, aS, _, _), (xn, no, _, pN, brN, needN, aN, _, _) = prepared
            lib, pa_s, pa_n = L.lib(), L.ptr_array(pS + [brS]), L.ptr_array(pN + [brN])
            phases = 3 if K1_BF16Z else 3 | 256
            image = None
            if _PLANES is not None and K1_BF16Z and K1_IMAGE:
                # the parameter image of the kernel's prologue, built once per scope instead of by every workgroup of every launch
                # (csrc/gatv2_hetero.hip, K1Image: -0.85 us of a 20-us rollout launch).  Keyed by storage AND version counters.
                ws = pS + pN + [t for t in (brS, brN) if t is not None]
                key = ("k1img", nh, D) + tuple(t.data_ptr() for t in ws) + tuple(t._version for t in ws)
                image = _cached_planes(key, lib.uavgnn_gatv2_hetero_image_bytes(), x_dst.device,
                                       lambda buf: L.check(lib.uavgnn_gatv2_hetero_prepare(pa_s, pa_n, nh, D, NEG_SLOPE, buf.data_ptr(),
                                                                                           L.stream()), "uavgnn_gatv2_hetero_prepare"),
                                       keep=ws)
            with KERNEL_TIMER.span("gatv2_hetero_fwd", (xs.shape[0], xn.shape[0], N, int(needS), int(needN))):
                head = (L.ptr(xs), xs.shape[0], L.ptr(so), L.ptr(oo), L.ptr(xn), xn.shape[0], L.ptr(no), L.ptr(x_dst), N, pa_s, pa_n,
                        nh, D, NEG_SLOPE)
                tail = (out.data_ptr(), R * H, L.ptr(aS), L.ptr(aN), phases, L.stream())
                _HeteroGATv2.last_rowmax = None
                if K1_ROWMAX and K1_BF16Z and GEMM_H2 and GEMM_X3 and (N > K1_ROWMAX_MIN_ROWS or
                                                                       (N >= K1_ROWMAX_DENSE_MIN_ROWS and xs.shape[0] >= K1_ROWMAX_DENSE_DEG * N)):
                    # time-batched launch: the maxima of the two halves of every output row on the way (+9 % on this launch) - the f_aggr
                    # product behind it runs on the f16x2 kernel (-25 %); the store-bound rollout launch on env-realistic degrees would pay
                    # +1.9 us of 18: plain there.  A rollout launch on DENSE degrees (mean in-degree of `seen` >= 16, known from the shapes:
                    # every `gt` source has one out-edge) is compute-bound at 85-90 us: +2 % there, -17 us on the 58-us GEMM behind it
                    rmA, rmB = th.empty(N, dtype=th.float32, device=x_dst.device), th.empty(N, dtype=th.float32, device=x_dst.device)
                    rc = lib.uavgnn_gatv2_hetero_fwd_rowmax(*head, L.ptr(image), out.data_ptr(), R * H, L.ptr(aS), L.ptr(aN), rmA.data_ptr(),
                                                            rmB.data_ptr(), phases, L.stream())
                    _HeteroGATv2.last_rowmax = (out.data_ptr(), rmA, rmB)
                elif image is not None:
                    rc = lib.uavgnn_gatv2_hetero_fwd_image(*head, image.data_ptr(), *tail)
                else:
                    rc = lib.uavgnn_gatv2_hetero_fwd_phases(*head, *tail)
            if rc == L.UAVGNN_EUNSUPPORTED:
                fused = False
            else:
                L.check(rc, "uavgnn_gatv2_hetero_fwd")
        for i in range(R):
            x_src, seg_off, order, p, b_r_c, need, a_save, FS, has_br = prepared[i]
            if not fused:
                with KERNEL_TIMER.span(f"gatv2_fwd[F={FS}]", (x_src.shape[0], N, int(need))):
                    rc = L.lib().uavgnn_gatv2_fwd(L.ptr(x_src), x_src.shape[0], FS, L.ptr(x_dst), x_dst.shape[1],
                                                  L.ptr(seg_off), L.ptr(order), N, *[L.ptr(t) for t in p], L.ptr(b_r_c), nh,
                                                  D, NEG_SLOPE, out.data_ptr() + 4 * i * H, R * H, L.ptr(a_save), L.stream())
                L.check(rc, "uavgnn_gatv2_fwd")
            saved += [x_src, seg_off, order if order is not None else seg_off, *p,
                      a_save if a_save is not None else x_dst]
            has_order.append(order is not None)
            meta.append((FS, need, has_br, has_order[-1]))
        ctx.nh, ctx.meta, ctx.H = nh, meta, H
        ctx.save_for_backward(*saved, out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        saved = ctx.saved_tensors
        x_dst, out = saved[0], saved[-1]
        nh, H = ctx.nh, ctx.H
        R = len(ctx.meta)
        N, dev = x_dst.shape[0], x_dst.device
        d_out = L.f32c(d_out)
        grads = [None, None, None]
        for i, (FS, need, has_br, has_ord) in enumerate(ctx.meta):
            x_src, seg_off, order, W_s, b_s, W_d, b_d, attn, W_r, a_save = saved[1 + i * 10: 1 + (i + 1) * 10]
            if not need or N == 0:
                grads += [None] * 10
                continue
            g = [th.empty_like(W_s), th.empty_like(b_s), th.empty_like(W_d), th.empty_like(b_d),
                 th.empty_like(attn), th.empty_like(W_r), th.empty(H, dtype=th.float32, device=dev)]
            ws_bytes = L.lib().uavgnn_gatv2_bwd_workspace_bytes(FS, H)
            ws = th.empty(ws_bytes // 4, dtype=th.float32, device=dev)
            with KERNEL_TIMER.span(f"gatv2_bwd[F={FS}]"):
                rc = L.lib().uavgnn_gatv2_bwd(L.ptr(x_src), x_src.shape[0], FS, L.ptr(x_dst), x_dst.shape[1], L.ptr(seg_off),
                                              L.ptr(order) if has_ord else None, N, L.ptr(W_s), L.ptr(b_s), L.ptr(W_d), L.ptr(b_d), L.ptr(attn), nh, H // nh,
                                              NEG_SLOPE, out.data_ptr() + 4 * i * H, d_out.data_ptr() + 4 * i * H,
                                              R * H, L.ptr(a_save), *[L.ptr(t) for t in g], ws.data_ptr(), ws_bytes,
                                              L.stream())
            L.check(rc, "uavgnn_gatv2_bwd")
            dW_s, db_s, dW_d, db_d, dattn, dW_r, db_r = g
            grads += [None, None, None, dattn, dW_s, db_s, dW_d, db_d, dW_r, db_r if has_br else None]
        return tuple(grads)


def hetero_gatv2(x_dst, nh, relations):
    """relations: list of (x_src [E,F], seg_off [N+1] int32, dst_order [N] int32 or None, conv) with conv exposing attn,
    fc_src, fc_dst, res_fc."""
    flat = []
    for x_src, seg_off, order, conv in relations:
        flat += [x_src, seg_off, order, conv.attn, conv.fc_src.weight, conv.fc_src.bias, conv.fc_dst.weight,
                 conv.fc_dst.bias, conv.res_fc.weight, conv.res_fc.bias]
    _HeteroGATv2.last_rowmax = None
    out = _HeteroGATv2.apply(x_dst, nh, th.is_grad_enabled(), *flat)
    rm, _HeteroGATv2.last_rowmax = _HeteroGATv2.last_rowmax, None
    if rm is not None and rm[0] == out.data_ptr():
        out._uavgnn_rowmax = rm[1:]        # (row maxima of the `near` / `seen` halves: read by linear_relu right behind this call)
    return out


def _talk_transpose_if_needed(g, *tensors):
    """The transpose of the talk CSC is a backward-only index: skip building it for no-grad forwards (act, target)."""
    if th.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return g.talk_transpose()
    return None, None, None


def _talk_env(g, M, K):
    """(graph_off, B, n_max) when the batch is made of small graphs the per-graph K3b kernels cover, else None."""
    go, n_max = g.graph_off, g.hints.get("max_graph_agents")
    if go is None or not go.is_cuda or n_max is None or n_max < 1:
        return None
    if not L.lib().uavgnn_talk_attn_env_supported(int(n_max), int(M), int(K)):
        return None
    return go, go.numel() - 1, int(n_max)


def _launch_talk_fwd(env, s, ld_s, q, ld_q, v, ld_v, K, M, talk_off, talk_src, N, scale, c, ld_c, a_save, x_copy, ld_x,
                     n_copy):
    """K3b forward on raw pointers: per-graph kernel when `env` describes the batch, per-destination kernel otherwise."""
    with KERNEL_TIMER.span("talk_attn_fwd"):
        if env is not None:
            go, B, n_max = env
            rc = L.lib().uavgnn_talk_attn_env_fwd(s, ld_s, q, ld_q, v, ld_v, K, M, L.ptr(talk_off), L.ptr(talk_src),
                                                  go.data_ptr(), B, n_max, scale, c, ld_c, a_save, x_copy, ld_x, n_copy,
                                                  L.stream())
        else:
            rc = L.lib().uavgnn_talk_attn_fwd(s, ld_s, q, ld_q, v, ld_v, K, M, L.ptr(talk_off), L.ptr(talk_src), N,
                                              scale, c, ld_c, a_save, x_copy, ld_x, n_copy, L.stream())
    L.check(rc, "uavgnn_talk_attn_fwd")


def _launch_talk_bwd(env, s, ld_s, q, ld_q, v, ld_v, K, M, talk_off, talk_src, transpose, N, scale, a_save, d_c, ld_dc,
                     d_s, ld_ds, d_q, ld_dq, d_v, ld_dv):
    with KERNEL_TIMER.span("talk_attn_bwd"):
        if env is not None:
            go, B, n_max = env
            rc = L.lib().uavgnn_talk_attn_env_bwd(s, ld_s, q, ld_q, v, ld_v, K, M, L.ptr(talk_off), L.ptr(talk_src),
                                                  go.data_ptr(), B, n_max, scale, a_save.data_ptr(), d_c, ld_dc, d_s,
                                                  ld_ds, d_q, ld_dq, d_v, ld_dv, L.stream())
        else:
            t_off, t_dst, t_pos = transpose
            de = th.empty_like(a_save) if s is not None else None
            rc = L.lib().uavgnn_talk_attn_bwd(s, ld_s, q, ld_q, v, ld_v, K, M, L.ptr(talk_off), L.ptr(talk_src),
                                              L.ptr(t_off), L.ptr(t_dst), L.ptr(t_pos), N, scale, a_save.data_ptr(),
                                              d_c, ld_dc, d_s, ld_ds, d_q, ld_dq, d_v, ld_dv, L.ptr(de), L.stream())
    L.check(rc, "uavgnn_talk_attn_bwd")


class _TalkAttention(th.autograd.Function):
    """K3b.  c_v = sum_u softmax_u(<s_u, q_v> * scale) v_u over the talk relation; s = q = None -> mean."""

    @staticmethod
    def forward(ctx, s, q, v, talk_off, talk_src, t_off, t_dst, t_pos, scale, env=None):
        L.require_gpu(v, talk_off, talk_src, s, q)
        N, M = v.shape
        K = 0 if s is None else s.shape[1]
        for t in (s, q, v):
            if t is not None and (t.dtype != th.float32 or t.stride(1) != 1):
                raise L.UavGnnError("talk_attention: float32 row-major inputs required")
        E = talk_src.shape[0]
        c = th.empty((N, M), dtype=th.float32, device=v.device)
        a_save = th.empty(max(E, 1), dtype=th.float32, device=v.device)
        _launch_talk_fwd(env, L.ptr(s), 0 if s is None else s.stride(0), L.ptr(q), 0 if q is None else q.stride(0),
                         L.ptr(v), v.stride(0), K, M, talk_off, talk_src, N, float(scale), c.data_ptr(), c.stride(0),
                         a_save.data_ptr(), None, 0, 0)
        ctx.scale, ctx.uniform, ctx.env = float(scale), s is None, env
        ctx.save_for_backward(*(t for t in (s, q) if t is not None), v, talk_off, talk_src, t_off, t_dst, t_pos, a_save)
        return c

    @staticmethod
    def backward(ctx, d_c):
        if ctx.uniform:
            v, talk_off, talk_src, t_off, t_dst, t_pos, a_save = ctx.saved_tensors
            s = q = None
        else:
            s, q, v, talk_off, talk_src, t_off, t_dst, t_pos, a_save = ctx.saved_tensors
        N, M = v.shape
        K = 0 if s is None else s.shape[1]
        d_c = d_c.contiguous()
        d_v = th.empty((N, M), dtype=th.float32, device=v.device)
        d_s = d_q = None
        if not ctx.uniform:
            d_s = th.empty((N, K), dtype=th.float32, device=v.device)
            d_q = th.empty((N, K), dtype=th.float32, device=v.device)
        _launch_talk_bwd(ctx.env, L.ptr(s), 0 if s is None else s.stride(0), L.ptr(q), 0 if q is None else q.stride(0),
                         L.ptr(v), v.stride(0), K, M, talk_off, talk_src, (t_off, t_dst, t_pos), N, ctx.scale, a_save,
                         d_c.data_ptr(), d_c.stride(0), L.ptr(d_s), K, L.ptr(d_q), K, d_v.data_ptr(), M)
        return d_s, d_q, d_v, None, None, None, None, None, None, None


def talk_attention(s, q, v, g, scale=1.0):
    """g: HeteroBatch carrying the talk relation."""
    off, src = g.talk_csc()
    env = _talk_env(g, v.shape[1], 0 if s is None else s.shape[1])
    t_off, t_dst, t_pos = (None, None, None) if env is not None else _talk_transpose_if_needed(g, s, q, v)
    return _TalkAttention.apply(s, q, v, off, src, t_off, t_dst, t_pos, scale, env)


class _GruGates(th.autograd.Function):
    """K4 pointwise half: h' from the two [N,3H] pre-activation blocks of a GRU cell."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        L.require_gpu(gi, gh, h)
        gi, gh, h = L.f32c(gi), L.f32c(gh), L.f32c(h)
        N, H = h.shape
        h_out = th.empty_like(h)
        with KERNEL_TIMER.span("gru_gates_fwd"):
            rc = L.lib().uavgnn_gru_gates_fwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), N, H, h_out.data_ptr(),
                                              L.stream())
        L.check(rc, "uavgnn_gru_gates_fwd")
        ctx.save_for_backward(gi, gh, h)
        return h_out

    @staticmethod
    def backward(ctx, d_hout):
        gi, gh, h = ctx.saved_tensors
        N, H = h.shape
        d_hout = L.f32c(d_hout)
        d_gi, d_gh, d_h = th.empty_like(gi), th.empty_like(gh), th.empty_like(h)
        with KERNEL_TIMER.span("gru_gates_bwd"):
            rc = L.lib().uavgnn_gru_gates_bwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), d_hout.data_ptr(), N, H,
                                              d_gi.data_ptr(), d_gh.data_ptr(), d_h.data_ptr(), L.stream())
        L.check(rc, "uavgnn_gru_gates_bwd")
        return d_gi, d_gh, d_h


def gru_gates(gi, gh, h):
    return _GruGates.apply(gi, gh, h)


GRU_FUSED = True   # K4 as one kernel (csrc/gru_fused.hip); False: vendor GEMMs + the gate kernel (A/B, tools/gru_probe.py)


GRU_FUSED_MIN_ROWS = 1024   # below: a handful of workgroups walk 18 K slices serially (25 us at 8 rows; vendor GEMMs + gates: 15 us)


def gru_cell_supported(inp, h) -> bool:
    return bool(GRU_FUSED and inp.shape[0] >= GRU_FUSED_MIN_ROWS and inp.is_cuda and inp.dtype == th.float32 and h.dtype == th.float32 and inp.stride(1) == 1
                and inp.stride(0) % 4 == 0 and inp.data_ptr() % 16 == 0
                and L.lib().uavgnn_gru_cell_supported(inp.shape[1], h.shape[1]))


GRU_X3 = os.environ.get("UAVGNN_GRU_X3", "1") != "0"   # the cell's GEMMs as bf16x3 splits on the bf16 matrix cores (csrc/gru_x3.hip)
# ... as exactly scaled two-term f16 splits, three products per fp32 product (csrc/gru_h2.hip), where the producer of the cell's input
# hands over the row maxima (the TarMAC step); UAVGNN_GRU_H2=0: the bf16x3 cell everywhere (A/B)
GRU_H2 = os.environ.get("UAVGNN_GRU_H2", "1") != "0"

# bf16 planes of weight matrices (and the parameter image of the fused K1 forward), reused ONLY inside a `frozen_weights()`
# scope.  A drop-in module's weights may change behind any cache (`.data` writes bump no version counter), so by default every
# call splits its weights again (3-5 us, one launch).  The learner's loss forward + backward is one call during which nobody
# can touch the parameters: 101 recurrent steps and 51 backward steps share their planes there (~250 launches, ~1.1 ms of a C3
# cycle).  A rollout between two optimiser steps is the other such interval: ``MultiAgentQLearner.act`` passes a store that
# lives until the learner itself changes the parameters (``apply`` / ``load_checkpoint`` / ``invalidate_weight_cache``).
_PLANES = None
_PLANES_MAX = 128


class frozen_weights:
    """Scope in which the caller guarantees that no parameter changes: weight planes are built once per (storage, layout).

    ``store``: a dict owned by the caller that outlives the scope (entries are reused by later scopes over the same dict until
    the caller clears it); default: a store that dies with the outermost scope."""

    def __init__(self, store=None):
        self.store = store

    def __enter__(self):
        global _PLANES
        self.prev = _PLANES
        if self.store is not None:
            _PLANES = self.store
        elif _PLANES is None:
            _PLANES = {}
        return self

    def __exit__(self, *exc):
        global _PLANES
        _PLANES = self.prev
        return False


def _cached_planes(key, nbytes, device, build, keep=()):
    """planes tensor for `key`; `build(planes)` launches the split when the scope has not seen the key yet.

    The key holds device ADDRESSES of the weights, so an entry also holds strong references to them (`keep`): while the
    entry lives the caching allocator cannot hand those addresses to another tensor, i.e. a hit IS the same storage.  (A
    temporary weight - TarMAC's stacked projection built by th.cat on every call - freed after a no-grad target-network
    step would otherwise let the next policy step's temporary of the same shape land on the same address and pick up the
    TARGET network's planes.)"""
    if _PLANES is not None:
        hit = _PLANES.pop(key, None)
        if hit is not None:
            _PLANES[key] = hit              # most recently used last (dicts keep insertion order)
            return hit[0]
    planes = th.empty(nbytes, dtype=th.uint8, device=device)
    build(planes)
    if _PLANES is not None:
        # a long-lived store fed with temporaries (weights built by th.cat per call) must not grow without bound: the least
        # recently used entries go, one by one - never the whole store in the middle of a scope, and never the K1 parameter image
        while len(_PLANES) >= _PLANES_MAX:
            victim = next((k for k in _PLANES if k[0] != "k1img"), None)
            if victim is None:
                break
            del _PLANES[victim]
        _PLANES[key] = (planes, tuple(keep))
    return planes


def gru_cell_two_piece_supported(x, c, h) -> bool:
    """The bf16x3 cell takes its input as [x || c] from two buffers (no concatenated copy) when both pieces are multiples of
    32 columns wide, 16-byte aligned with row strides of whole float4s."""
    K1, K2, H = x.shape[1], c.shape[1], h.shape[1]
    return bool(GRU_X3 and K1 >= 32 and K1 % 32 == 0 and K2 % 32 == 0 and x.stride(1) == 1 and c.stride(1) == 1
                and x.stride(0) % 4 == 0 and c.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and c.data_ptr() % 16 == 0
                and L.lib().uavgnn_gru_cell_x3_supported(K1 + K2, H)
                and 4 * x.shape[0] * max(x.stride(0), c.stride(0), H) < 2 ** 32)


def gru_cell_h2_supported(K_in, H, N, ld_max) -> bool:
    """The f16x2 cell (csrc/gru_h2.hip) covers this call - given a `row_absmax` from the kernel that produced the input."""
    return bool(GRU_H2 and GRU_X3 and N >= GRU_FUSED_MIN_ROWS and L.lib().uavgnn_gru_cell_h2_supported(K_in, H)
                and 4 * N * max(ld_max, H) < 2 ** 32)


def _gru_cell_launch(inp, h, W_ih, b_ih, W_hh, b_hh, save, inp2=None, h2_out=None, rowmax=None):
    """h' (and the [N, 4H] pre-activation sets when `save`) of the fused GRU cell.  inp2: second piece of the input
    ([inp || inp2] is what W_ih multiplies; the caller checked gru_cell_two_piece_supported).  h2_out: contiguous [N, H]
    buffer h' is written into (a slot of the time-batched staging of a BPTT sequence).  rowmax [N]: max |.| over every row of
    [inp || inp2 || h], written by the kernel that produced the input (the fused TarMAC message launch) - selects the f16x2 cell
    (csrc/gru_h2.hip: half the matrix-core work of the bf16x3 cell; the caller checked gru_cell_h2_supported)."""
    N, H = h.shape
    h2 = th.empty_like(h) if h2_out is None else h2_out
    pre = th.empty((N, 4 * H), dtype=th.float32, device=h.device) if save else None
    if rowmax is not None:
        lib, K1, K2 = L.lib(), inp.shape[1], (0 if inp2 is None else inp2.shape[1])
        K_in = K1 + K2
        with KERNEL_TIMER.span("gru_cell_fwd", (N, K_in, H, "f16x2")):
            planes = _cached_planes(("gruh2", W_ih.data_ptr(), W_hh.data_ptr(), W_ih._version, W_hh._version, K_in, H),
                                    lib.uavgnn_gru_cell_h2_workspace_bytes(K_in, H), h.device,
                                    lambda p: L.check(lib.uavgnn_gru_split_weights_h2(W_ih.data_ptr(), K_in, W_hh.data_ptr(), H,
                                                                                      p.data_ptr(), L.stream()),
                                                      "uavgnn_gru_split_weights_h2"), keep=(W_ih, W_hh))
            rc = lib.uavgnn_gru_cell_fwd_h2(inp.data_ptr(), inp.stride(0), K1, L.ptr(inp2), 0 if inp2 is None else inp2.stride(0), K2,
                                            h.data_ptr(), N, H, rowmax.data_ptr(), planes.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                                            h2.data_ptr(), L.ptr(pre), L.stream())
        L.check(rc, "uavgnn_gru_cell_fwd_h2")
        return h2, pre
    if inp2 is not None or (GRU_X3 and L.lib().uavgnn_gru_cell_x3_supported(inp.shape[1], H)
                            and 4 * N * max(inp.stride(0), H) < 2 ** 32):   # 32-bit byte offsets inside the kernel (3.3 M rows at K_in = 320)
        lib, K1, K2 = L.lib(), inp.shape[1], (0 if inp2 is None else inp2.shape[1])
        K_in = K1 + K2
        with KERNEL_TIMER.span("gru_cell_fwd", (N, K_in, H, "bf16x3")):
            # the planes are rebuilt on every call outside a frozen_weights() scope: nothing observable tells when a drop-in
            # module's weights changed
            planes = _cached_planes(("gru", W_ih.data_ptr(), W_hh.data_ptr(), W_ih._version, W_hh._version, K_in, H),
                                    lib.uavgnn_gru_cell_x3_workspace_bytes(K_in, H), h.device,
                                    lambda p: L.check(lib.uavgnn_gru_split_weights(W_ih.data_ptr(), K_in, W_hh.data_ptr(), H,
                                                                                   p.data_ptr(), L.stream()),
                                                      "uavgnn_gru_split_weights"), keep=(W_ih, W_hh))
            rc = lib.uavgnn_gru_cell_fwd_x3_opts(inp.data_ptr(), inp.stride(0), K1, L.ptr(inp2),
                                                 0 if inp2 is None else inp2.stride(0), K2, h.data_ptr(), N, H, planes.data_ptr(),
                                                 b_ih.data_ptr(), b_hh.data_ptr(), h2.data_ptr(), L.ptr(pre), GRU_X3_FLAGS,
                                                 L.stream())
        L.check(rc, "uavgnn_gru_cell_fwd_x3_opts")
        return h2, pre
    with KERNEL_TIMER.span("gru_cell_fwd", (N, inp.shape[1], H, "f32")):
        rc = L.lib().uavgnn_gru_cell_fwd(inp.data_ptr(), inp.stride(0), inp.shape[1], h.data_ptr(), N, H, W_ih.data_ptr(),
                                         b_ih.data_ptr(), W_hh.data_ptr(), b_hh.data_ptr(), h2.data_ptr(), L.ptr(pre),
                                         L.stream())
    L.check(rc, "uavgnn_gru_cell_fwd")
    return h2, pre


DINP_SPLIT = os.environ.get("UAVGNN_DINP_SPLIT", "1") != "0"   # _TarmacStep.backward: d x and d c of the GRU input as two products
# A/B switch: UAVGNN_HEAD_FUSED_BWD=0 forms d h' = d_hout + dq W_out with a vendor GEMM in front of the gate kernel
HEAD_FUSED_BWD = os.environ.get("UAVGNN_HEAD_FUSED_BWD", "1") != "0"


def _gru_gates_bwd_from_pre(pre, h, d_hout, d_gi=None, d_gh=None, head=None, sums=None, rowmax=None):
    """Gate gradients from the saved pre-activation sets.  ``head`` = (dq [N, n_out], W_out [n_out, H]): the gradient of h' is
    d_hout (None = 0) + dq W_out, formed inside the kernel (uavgnn_gru_gates_bwd_fused_head).  ``sums``: a
    [uavgnn_gru_gates_bwd_sum_rows(N, H), 4H] buffer that receives the launch's per-workgroup column sums (the bias gradients'
    share of this step)."""
    N, H = h.shape
    if d_gi is None:
        d_gi = th.empty((N, 3 * H), dtype=th.float32, device=h.device)
    if d_gh is None:
        d_gh = th.empty_like(d_gi)
    dh = th.empty_like(h)
    with KERNEL_TIMER.span("gru_gates_bwd"):
        if sums is not None and rowmax is not None:      # ... + the row maxima of d_gi / d_gh (`rowmax` [N], H = 256)
            dq, W_out = head if head is not None else (None, None)
            rc = L.lib().uavgnn_gru_gates_bwd_fused_sums_rowmax(pre.data_ptr(), h.data_ptr(), L.ptr(d_hout), L.ptr(dq),
                                                                0 if dq is None else dq.shape[1], L.ptr(W_out), N, H, d_gi.data_ptr(),
                                                                d_gh.data_ptr(), dh.data_ptr(), sums.data_ptr(), rowmax.data_ptr(),
                                                                L.stream())
        elif sums is not None:
            dq, W_out = head if head is not None else (None, None)
            rc = L.lib().uavgnn_gru_gates_bwd_fused_sums(pre.data_ptr(), h.data_ptr(), L.ptr(d_hout), L.ptr(dq),
                                                         0 if dq is None else dq.shape[1], L.ptr(W_out), N, H, d_gi.data_ptr(),
                                                         d_gh.data_ptr(), dh.data_ptr(), sums.data_ptr(), L.stream())
        elif head is not None:
            dq, W_out = head
            rc = L.lib().uavgnn_gru_gates_bwd_fused_head(pre.data_ptr(), h.data_ptr(), L.ptr(d_hout), dq.data_ptr(), dq.shape[1],
                                                         W_out.data_ptr(), N, H, d_gi.data_ptr(), d_gh.data_ptr(), dh.data_ptr(),
                                                         L.stream())
        else:
            rc = L.lib().uavgnn_gru_gates_bwd_fused(pre.data_ptr(), h.data_ptr(), d_hout.data_ptr(), N, H, d_gi.data_ptr(),
                                                    d_gh.data_ptr(), dh.data_ptr(), L.stream())
    L.check(rc, "uavgnn_gru_gates_bwd_fused")
    return d_gi, d_gh, dh


class _GruCellFused(th.autograd.Function):
    """nn.GRUCell as ONE forward launch (K4); backward = gate kernel on the saved pre-activations + vendor GEMMs."""

    @staticmethod
    def forward(ctx, inp, h, W_ih, b_ih, W_hh, b_hh, train, rowmax=None):
        inp, h = L.f32c(inp), L.f32c(h)
        p = [L.f32c(t.detach()) for t in (W_ih, b_ih, W_hh, b_hh)]
        h2, pre = _gru_cell_launch(inp, h, *p, save=bool(train), rowmax=rowmax)
        ctx.have_pre = bool(train)
        if train:
            ctx.save_for_backward(inp, h, pre, p[0], p[2])
        return h2

    @staticmethod
    def backward(ctx, d_h2):
        if not ctx.have_pre:
            raise L.UavGnnError("gru_cell: backward through a forward that saved no pre-activations (train=False)")
        inp, h, pre, W_ih, W_hh = ctx.saved_tensors
        d_gi, d_gh, dh = _gru_gates_bwd_from_pre(pre, h, L.f32c(d_h2))
        d_inp = _mm_nn(d_gi, W_ih) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1]:
            _mm_nn(d_gh, W_hh, out=dh, accumulate=True)
        else:
            dh = None
        gWih = _wgrad(d_gi, inp) if ctx.needs_input_grad[2] else None
        gbih = _colsum(d_gi) if ctx.needs_input_grad[3] else None
        gWhh = _wgrad(d_gh, h) if ctx.needs_input_grad[4] else None
        gbhh = _colsum(d_gh) if ctx.needs_input_grad[5] else None
        return d_inp, dh, gWih, gbih, gWhh, gbhh, None, None


def row_absmax(*pieces, out=None):
    """[N] = max |.| per row over up to three row-major fp32 matrices with N rows each (Inf for a row that holds Inf / NaN): the
    `rowmax` of the f16x2 GRU cell for callers whose producer does not hand it over (one extra pass over the operand)."""
    ps = [L.f32c(t) for t in pieces]
    N = ps[0].shape[0]
    if out is None:
        out = th.empty(N, dtype=th.float32, device=ps[0].device)
    args = []
    for i in range(3):
        t = ps[i] if i < len(ps) else None
        args += [L.ptr(t), 0 if t is None else t.stride(0), 0 if t is None else t.shape[1]]
    L.check(L.lib().uavgnn_row_absmax(*args, N, out.data_ptr(), L.stream()), "uavgnn_row_absmax")
    return out


def gru_cell(inp, h, cell, rowmax=None):
    """nn.GRUCell(inp, h) with `cell`'s parameters: fused kernel when the shape has an instantiation, else vendor GEMMs +
    the gate kernel.  rowmax [N] (``row_absmax(inp, h)`` or a producer's): the f16x2 cell where it covers the shape."""
    if gru_cell_supported(inp, h):
        # ANY differentiable input makes autograd run the backward, which reads the saved [N, 4H] pre-activation sets
        train = th.is_grad_enabled() and any(t.requires_grad for t in (inp, h, cell.weight_ih, cell.bias_ih,
                                                                       cell.weight_hh, cell.bias_hh))
        if rowmax is not None and not (GRU_H2 and GRU_X3 and L.lib().uavgnn_gru_cell_h2_supported(inp.shape[1], h.shape[1])
                                       and 4 * inp.shape[0] * max(inp.stride(0), h.shape[1]) < 2 ** 32):
            rowmax = None
        return _GruCellFused.apply(inp, h, cell.weight_ih, cell.bias_ih, cell.weight_hh, cell.bias_hh, train, rowmax)
    gi = linear(inp, cell.weight_ih, cell.bias_ih)
    gh = linear(h, cell.weight_hh, cell.bias_hh)
    return gru_gates(gi, gh, h)


def _row_blocks(n, cap=256, min_rows=32):
    """Largest power-of-two block count <= cap that divides n with >= min_rows rows per block."""
    S = 1
    while S < cap and n % (2 * S) == 0 and n // (2 * S) >= min_rows:
        S *= 2
    return S


def _colsum_partial(dy):
    """[S, C] partial column sums of a (possibly column-sliced) [n, C] matrix.  torch's one-pass column reduction
    over 10^4..10^5 rows runs at 0.01-3 TB/s depending on C (48 us for [32768, 9], 34 us for [32768, 768]); blocking
    the rows S = 256 ways first reaches ~5 TB/s for every width (tools/colsum_probe.py), fixed order."""
    n = dy.shape[0]
    S = _row_blocks(n)
    return dy.unflatten(0, (S, n // S)).sum(1)


def _colsum(dy):
    return _colsum_partial(dy).sum(0)


# ---------------------------------------------------------------------------------------------------------------------
# Dense layers on the bf16 matrix cores (csrc/gemm_x3.hip): fp32 in / out, six exact bf16 products per fp32 product
# ---------------------------------------------------------------------------------------------------------------------
GEMM_X3 = os.environ.get("UAVGNN_GEMM_X3", "1") != "0"   # False: vendor fp32 GEMMs (A/B, tools/gemm_x3_probe.py)
# Measured against the recorded vendor solutions at C3 (tools/gemm_x3_probe.py, profiles/r02_gemm_x3_probe.txt): 1.10-1.30x on
# outputs that tile by 128 columns (256, 512), 1.01x on 320 (2.5 tiles), 0.84x on 96: only the first group takes the kernel.


# A/B variants of the bf16x3 kernels are PER-CALL flag words of the C ABI (include/uavgnn.h: the library keeps no process-wide
# state); the probes set these module attributes, the environment seeds them: GEMM 8 (default) / 9 (staging interleaved) / 4
# (128 x 128 tiles), GRU cell 1 (default: staging interleaved) / 0 (staging in blocks)
GEMM_X3_FLAGS = {"8": 0, "9": 4, "4": 8}.get(os.environ.get("UAVGNN_GEMM_X3_VARIANT", "8"), 0)
GRU_X3_FLAGS = 1 if os.environ.get("UAVGNN_GRU_X3_VARIANT", "1") == "0" else 0


GEMM_X3_SMALL_GRID = 128   # fewer 256 x 128 tiles than this: 128 x 128 tiles instead


def gemm_x3_supported(a, n_out, k) -> bool:
    return bool(GEMM_X3 and a.is_cuda and a.dtype == th.float32 and a.dim() == 2 and a.stride(1) == 1
                and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and n_out % 128 == 0 and a.shape[0] >= 4096
                and a.shape[0] * a.stride(0) < 2 ** 31 and L.lib().uavgnn_gemm_x3_supported(a.shape[0], n_out, k))


def gemm_x3(a, W, transpose_w=False, bias=None, out=None, accumulate=False, relu=False, rowmax_out=None):
    """out = a @ W.T (transpose_w=False, W [n_out, k]) or a @ W (transpose_w=True, W [k, n_out]) (+ bias) (+ out) (relu).
    `W` may be a strided view with unit inner stride; its bf16 planes are rebuilt on every call (3-5 us: nothing observable
    tells when a drop-in module's weights changed) unless the caller opened a frozen_weights() scope.  Caller checks gemm_x3_supported()."""
    lib = L.lib()
    M, K = a.shape
    R, C = W.shape
    n_out = C if transpose_w else R
    assert (R if transpose_w else C) == K and W.stride(1) == 1
    if out is None:
        out = th.empty((M, n_out), dtype=th.float32, device=a.device)
    with KERNEL_TIMER.span("gemm_x3", (M, n_out, K)):
        planes = _cached_planes(("mat", W.data_ptr(), W._version, W.stride(0), R, C, bool(transpose_w)), 6 * R * C, a.device,
                                lambda p: L.check(lib.uavgnn_split_bf16x3(W.data_ptr(), W.stride(0), R, C, int(transpose_w),
                                                                          p.data_ptr(), L.stream()), "uavgnn_split_bf16x3"),
                                keep=(W,))
        flags = GEMM_X3_FLAGS
        if not (flags & 8) and ((M + 255) // 256) * ((n_out + 127) // 128) < GEMM_X3_SMALL_GRID:
            # C2-size batches (N_a = 4096): the 256 x 128 tiles of the eight-wave kernel are 32 workgroups on 256 CUs (58 us for
            # [4096, 768] x [768, 256]); the four-wave kernel's 128 x 128 tiles double the workgroups (UAVGNN_GEMM_TILE_128)
            flags |= 8
            if ((M + 127) // 128) * ((n_out + 127) // 128) < GEMM_X3_SMALL_GRID:
                flags = (flags & ~8) | 16         # still under half the CUs: 64 x 128 tiles (UAVGNN_GEMM_TILE_64)
        epi = (1 if accumulate else 0) | (2 if relu else 0) | flags
        if rowmax_out is not None and not (flags & 24):      # ... + max |.| over every row of `a`, a by-product of the staging
            rc = lib.uavgnn_gemm_nt_x3_rowmax(a.data_ptr(), a.stride(0), M, K, planes.data_ptr(), n_out, L.ptr(bias), out.data_ptr(),
                                              out.stride(0), epi, rowmax_out.data_ptr(), L.stream())
        else:
            if rowmax_out is not None:
                rowmax_out.copy_(a.abs().amax(1))
            rc = lib.uavgnn_gemm_nt_x3(a.data_ptr(), a.stride(0), M, K, planes.data_ptr(), n_out, L.ptr(bias), out.data_ptr(),
                                       out.stride(0), epi, L.stream())
    L.check(rc, "uavgnn_gemm_nt_x3")
    return out


def _mm_nt(x, W, b=None, relu=False):
    """x @ W.T (+ b) (relu): bf16x3 kernel when the shape has one, else the vendor GEMM."""
    if gemm_x3_supported(x, W.shape[0], W.shape[1]) and W.stride(1) == 1 and (b is None or b.is_contiguous()):
        return gemm_x3(x, W, False, bias=b, relu=relu)
    if relu:
        return th._addmm_activation(b, x, W.t())
    return th.addmm(b, x, W.t()) if b is not None else th.mm(x, W.t())


def _mm_nn(dy, W, out=None, accumulate=False, rowmax=None):
    """dy @ W (+ out when accumulate).  rowmax: the row maxima of dy from its producer -> the f16x2 kernel where it covers the shape."""
    if rowmax is not None and gemm_h2_supported(dy, W.shape[1], W.shape[0]) and W.stride(1) == 1 and \
            (out is None or (out.stride(1) == 1 and out.dtype == th.float32)):
        return gemm_h2(dy, W, rowmax, True, out=out, accumulate=accumulate)
    if gemm_x3_supported(dy, W.shape[1], W.shape[0]) and W.stride(1) == 1 and \
            (out is None or (out.stride(1) == 1 and out.dtype == th.float32)):
        return gemm_x3(dy, W, True, out=out, accumulate=accumulate)
    if out is None:
        return th.mm(dy, W)
    return out.addmm_(dy, W) if accumulate else th.mm(dy, W, out=out)


GATE_SUMS = os.environ.get("UAVGNN_GATE_SUMS", "1") != "0"   # bias gradients of the cell from the gate kernel's column sums (A/B switch)
DX_CAT = os.environ.get("UAVGNN_DX_CAT", "1") != "0"   # d x of the TarMAC step as ONE product over [d_gi || d_proj] (A/B switch)


def gemm_x3_cat_supported(a1, a2, n_out) -> bool:
    K1, K2 = a1.shape[1], a2.shape[1]
    return bool(DX_CAT and a2.is_cuda and a2.dtype == th.float32 and a2.dim() == 2 and a2.shape[0] == a1.shape[0] and K1 % 32 == 0
                and K2 % 32 == 0 and a2.stride(1) == 1 and a2.stride(0) % 4 == 0 and a2.data_ptr() % 16 == 0
                and a2.shape[0] * a2.stride(0) < 2 ** 31 and gemm_x3_supported(a1, n_out, K1 + K2)
                and ((a1.shape[0] + 255) // 256) * ((n_out + 127) // 128) >= GEMM_X3_SMALL_GRID and not (GEMM_X3_FLAGS & 8))


def gemm_x3_cat(a1, a2, W1, W2, out):
    """out = a1 @ W1 + a2 @ W2 (W1 [K1, n_out], W2 [K2, n_out], unit inner strides) as ONE bf16x3 product over the contraction
    [a1 || a2] (csrc/gemm_x3.hip, two-source loader; the stacked weight is split once per weight version inside a
    frozen_weights() scope).  Caller checks gemm_x3_cat_supported()."""
    lib = L.lib()
    M, K1 = a1.shape
    K2, n_out = a2.shape[1], W1.shape[1]
    K = K1 + K2
    assert W1.shape[0] == K1 and W2.shape == (K2, n_out) and W1.stride(1) == 1 and W2.stride(1) == 1

    def build(p):
        Wc = th.cat((W1, W2), 0)                                  # [K, n_out], contiguous
        L.check(lib.uavgnn_split_bf16x3(Wc.data_ptr(), n_out, K, n_out, 1, p.data_ptr(), L.stream()), "uavgnn_split_bf16x3")
    with KERNEL_TIMER.span("gemm_x3", (M, n_out, K)):
        planes = _cached_planes(("matcat", W1.data_ptr(), W1._version, W1.stride(0), W2.data_ptr(), W2._version, W2.stride(0), K1, K2,
                                 n_out), 6 * K * n_out, a1.device, build, keep=(W1, W2))
        rc = lib.uavgnn_gemm_nt_x3_cat(a1.data_ptr(), a1.stride(0), K1, a2.data_ptr(), a2.stride(0), M, K, planes.data_ptr(), n_out,
                                       None, out.data_ptr(), out.stride(0), GEMM_X3_FLAGS & 4, L.stream())
    L.check(rc, "uavgnn_gemm_nt_x3_cat")
    return out


# The input-gradient products of the recurrent step and of f_aggr on the f16x2 arithmetic (csrc/gemm_h2.hip: three f16 products per fp32
# product) where the kernel that produced the activation operand hands over its row maxima; UAVGNN_GEMM_H2=0: bf16x3 everywhere (A/B)
GEMM_H2 = os.environ.get("UAVGNN_GEMM_H2", "1") != "0"


def gemm_h2_supported(a, n_out, k) -> bool:
    """The eight-wave f16x2 kernel covers y = a B^T: what gemm_x3_supported asks, and a grid that fills the chip with 256 x 128 tiles
    (the bf16x3 path switches to smaller tiles below; the f16x2 kernel has none)."""
    return bool(GEMM_H2 and gemm_x3_supported(a, n_out, k) and L.lib().uavgnn_gemm_h2_supported(a.shape[0], n_out, k)
                and ((a.shape[0] + 255) // 256) * ((n_out + 127) // 128) >= GEMM_X3_SMALL_GRID)


def gemm_h2(a, W, rowmax, transpose_w=False, bias=None, out=None, accumulate=False, relu=False, a2=None, W2=None, rowmax2=None,
            rowmax2_out=None):
    """out = a @ W.T (transpose_w=False) or a @ W (transpose_w=True) (+ bias) (+ out) (relu) on the f16x2 kernel; with a2 / W2:
    [a || a2] @ [W; W2] (W [K1, n_out], W2 [K2, n_out], transposed form only).  rowmax (and rowmax2): per row an upper bound of
    max |.| over the row of the activation operand, from its producer(s); rowmax2_out [M] instead of rowmax2: the launch takes the row
    maxima of a2 itself and leaves them there (uavgnn_gemm_nt_h2_rm2).  Caller checks gemm_h2_supported()."""
    lib = L.lib()
    M, K1 = a.shape
    if a2 is not None:
        assert transpose_w and W2 is not None and W.stride(1) == 1 and W2.stride(1) == 1
        K2, n_out = a2.shape[1], W.shape[1]
        K = K1 + K2
        key = ("h2cat", W.data_ptr(), W._version, W.stride(0), W2.data_ptr(), W2._version, W2.stride(0), K1, K2, n_out)

        def build(p):
            Wc = th.cat((W, W2), 0)                                   # [K, n_out], contiguous
            L.check(lib.uavgnn_split_h2(Wc.data_ptr(), n_out, K, n_out, 1, p.data_ptr(), L.stream()), "uavgnn_split_h2")
        keep = (W, W2)
    else:
        R, C = W.shape
        n_out, K = (C, R) if transpose_w else (R, C)
        assert K == K1 and W.stride(1) == 1
        key = ("h2mat", W.data_ptr(), W._version, W.stride(0), R, C, bool(transpose_w))

        def build(p):
            L.check(lib.uavgnn_split_h2(W.data_ptr(), W.stride(0), R, C, int(transpose_w), p.data_ptr(), L.stream()), "uavgnn_split_h2")
        keep = (W,)
    if out is None:
        out = th.empty((M, n_out), dtype=th.float32, device=a.device)
    with KERNEL_TIMER.span("gemm_h2", (M, n_out, K)):
        planes = _cached_planes(key, lib.uavgnn_split_h2_bytes(n_out, K), a.device, build, keep=keep)
        fn, rm2 = (lib.uavgnn_gemm_nt_h2, rowmax2) if rowmax2_out is None else (lib.uavgnn_gemm_nt_h2_rm2, rowmax2_out)
        rc = fn(a.data_ptr(), a.stride(0), K1, L.ptr(a2), 0 if a2 is None else a2.stride(0), M, K, rowmax.data_ptr(),
                L.ptr(rm2), planes.data_ptr(), n_out, L.ptr(bias), out.data_ptr(), out.stride(0),
                (1 if accumulate else 0) | (2 if relu else 0) | (GEMM_X3_FLAGS & 4), L.stream())
    L.check(rc, "uavgnn_gemm_nt_h2")
    return out


GEMM_TN_X3 = os.environ.get("UAVGNN_GEMM_TN_X3", "1") != "0"   # weight gradients on the bf16 matrix cores (csrc/gemm_tn_x3.hip)
# Measured against the vendor's batched split-K fp32 GEMM (tools/gemm_tn_probe.py, profiles/r03_gemm_tn_probe.txt): BOTH operands
# have to be split and transposed inside the kernel, which bounds it at 95-108 TFLOP/s fp32-equivalent = the vendor's 106-108
# at the per-step shapes (32 768 rows; 0.4-0.6 x on the 96- and 9-row outputs), 142 vs 134 at the time-batched encoder shape
# (1.67 M rows).  Only the latter takes the kernel; its error against float64 is 0.5-0.8 x the vendor's on every shape.
GEMM_TN_MIN_ROWS = 1 << 18
# ... on the f16x2 arithmetic with LDS transposing reads (csrc/gemm_tn_h2.hip): the recurrent weights of a staged BPTT sequence
GEMM_TN_H2 = os.environ.get("UAVGNN_GEMM_TN_H2", "1") != "0"


def _max_two_stage(t):
    """max over all elements as row maxima of a [S, numel / S] view, then the maximum of the S row maxima - neither stage takes torch's
    multi-block semaphore path, whose final write did not always land inside a replayed hipGraph (learner._mse)."""
    n = t.numel()
    S = 1
    while S < 1024 and n % (2 * S) == 0 and n // (2 * S) >= 256:
        S *= 2
    return t.reshape(S, n // S).max(1).values.max()


def gemm_tn_h2_supported(dy, x, min_in=128) -> bool:
    """dy^T x on csrc/gemm_tn_h2.hip (256 x 128 output tiles).  min_in: the narrowest `x` worth a 128-wide tile - 128 by default; the
    96-column projections pass 96 with the operands SWAPPED (x^T d_proj = dWp^T: three quarters of a tile instead of three eighths)."""
    return bool(GEMM_X3 and GEMM_H2 and GEMM_TN_H2 and dy.is_cuda and dy.dtype == th.float32 and x.dtype == th.float32 and dy.dim() == 2
                and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.shape[0] >= GEMM_TN_MIN_ROWS and dy.stride(1) == 1 and x.stride(1) == 1
                and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
                and dy.shape[1] >= 256 and x.shape[1] >= min_in       # (a 9-row output leaves most of a 256 x 128 tile idle: the vendor GEMM)
                and L.lib().uavgnn_gemm_tn_h2_supported(dy.shape[0], dy.shape[1], x.shape[1]))


def gemm_tn_x3_supported(dy, x) -> bool:
    return bool(GEMM_X3 and GEMM_TN_X3 and dy.is_cuda and dy.dtype == th.float32 and x.dtype == th.float32 and dy.dim() == 2
                and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.shape[0] >= GEMM_TN_MIN_ROWS and dy.stride(1) == 1
                and x.stride(1) == 1 and dy.shape[1] > 0 and x.shape[1] > 0)


def gemm_tn_x3(dy, x, partials=None, accumulate=False):
    """partials [S, out, in] (+)= chunked dy^T x (csrc/gemm_tn_x3.hip; the caller sums over S).  Caller checks
    gemm_tn_x3_supported()."""
    lib = L.lib()
    n, Mo, Ko = dy.shape[0], dy.shape[1], x.shape[1]
    if partials is None:
        S = lib.uavgnn_gemm_tn_x3_chunks(n, Mo, Ko)
        partials = th.empty((S, Mo, Ko), dtype=th.float32, device=dy.device)
    with KERNEL_TIMER.span("gemm_tn_x3", (n, Mo, Ko)):
        rc = lib.uavgnn_gemm_tn_x3(dy.data_ptr(), dy.stride(0), Mo, x.data_ptr(), x.stride(0), Ko, n, partials.data_ptr(),
                                   partials.shape[0], int(accumulate), L.stream())
    L.check(rc, "uavgnn_gemm_tn_x3")
    return partials


class _LinearSplitK(th.autograd.Function):
    """y = x W^T (+ b) on the vendor GEMM (hipBLASLt/rocBLAS fp32), with a weight-gradient path shaped for this
    workload: N_a is 10^4..10^5 rows while W is at most 768 x 512, so dW = dY^T X has a tiny output and a huge
    reduction dimension.  A plain GEMM call leaves most CUs idle there (measured: 400 us for 768x320x32768); the
    reduction is split into S row chunks run as ONE batched GEMM [S, out, in] followed by a fixed-order sum over S
    (deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return _mm_nt(x, W, b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        return _LinearSplitK._grads(ctx, x, W, dy, ctx.has_bias)

    @staticmethod
    def _grads(ctx, x, W, dy, has_bias, rowmax=None, x_rowmax=None):
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = _mm_nn(dy, W, rowmax=rowmax)
        if ctx.needs_input_grad[1] and rowmax is not None and x_rowmax is not None and gemm_tn_h2_supported(dy, x):
            # f16x2 weight gradient: the global maxima of the producers' row maxima bound every column (see WeightGradSink.end_sequence)
            lib = L.lib()
            n, Mo, Ko = dy.shape[0], dy.shape[1], x.shape[1]
            S = lib.uavgnn_gemm_tn_h2_chunks(n, Mo, Ko)
            part = th.empty((S, Mo, Ko), dtype=th.float32, device=x.device)
            cy, cx = _max_two_stage(rowmax).expand(Mo).contiguous(), _max_two_stage(x_rowmax).expand(Ko).contiguous()
            with KERNEL_TIMER.span("gemm_tn_h2", (n, Mo, Ko)):
                rc = lib.uavgnn_gemm_tn_h2(dy.data_ptr(), dy.stride(0), Mo, x.data_ptr(), x.stride(0), Ko, n, cy.data_ptr(), cx.data_ptr(),
                                           part.data_ptr(), S, 0, L.stream())
            L.check(rc, "uavgnn_gemm_tn_h2")
            dW = part.sum(0)
        elif ctx.needs_input_grad[1] and gemm_tn_x3_supported(dy, x):
            dW = gemm_tn_x3(dy, x).sum(0)
        elif ctx.needs_input_grad[1]:
            n = x.shape[0]
            S = 1   # row chunks of >= 2048: S = 16 at N_a = 32768 (measured best or within 10 % on every layer shape)
            while S < 64 and n % (2 * S) == 0 and n // (2 * S) >= 2048:
                S *= 2
            if S > 1:
                xc = x if x.is_contiguous() else x.contiguous()
                part = th.bmm(dy.view(S, n // S, -1).transpose(1, 2), xc.view(S, n // S, -1))   # [S, out, in]
                dW = part.sum(0)
            else:
                dW = th.mm(dy.t(), x)
        if has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy)
        return dx, dW, db


def linear(x, W, b=None):
    return _LinearSplitK.apply(x, W, b)


RELU_BWD_FUSED = os.environ.get("UAVGNN_RELU_BWD_FUSED", "1") != "0"   # ReLU mask + bias gradient in one pass (A/B switch)


class _LinearReLU(th.autograd.Function):
    """relu(x W^T + b) with the bias + ReLU applied in the GEMM epilogue (hipBLASLt, via ``torch._addmm_activation``):
    the separate elementwise pass over [N_a, H] (and over [(T+1) N_a, H] in the time-batched encoder) disappears from
    the forward.  Backward masks dy with (y > 0) and reuses the split-K weight-gradient path."""

    @staticmethod
    def forward(ctx, x, W, b, rm_a=None, rm_b=None, train=True):
        ctx.x_rowmax = None
        ctx.n_extra = 0 if rm_a is None else 3
        n_out = W.shape[0]
        rm = None if rm_a is None else (rm_a, rm_b)
        if (rm is not None and b is not None and b.is_contiguous() and W.stride(1) == 1 and gemm_h2_supported(x, n_out, W.shape[1])):
            # the producer (the time-batched K1 launch) left the maxima of the two halves of every row: the layer on the f16x2 kernel
            y = gemm_h2(x, W, rm[0], False, bias=b, relu=True, rowmax2=rm[1])
            if ctx.needs_input_grad[1] and train:      # (needs_input_grad stays True for parameters under no_grad - `train` is the grad mode
                ctx.x_rowmax = th.maximum(rm[0], rm[1])    # at the call site: a rollout step does not pay for the weight gradient's bound)
            ctx.save_for_backward(x, W, y)
            return y
        if (ctx.needs_input_grad[1] and n_out % 4 == 0 and gemm_x3_supported(x, n_out, W.shape[1]) and W.stride(1) == 1 and b is not None
                and b.is_contiguous() and x.shape[0] >= GEMM_TN_MIN_ROWS and GEMM_TN_H2 and GEMM_H2 and n_out >= 256 and x.shape[1] >= 128
                and L.lib().uavgnn_gemm_tn_h2_supported(x.shape[0], n_out, x.shape[1])):
            # time-batched training forward: the layer's weight gradient will run on the f16x2 kernel - the bound of its X operand (this
            # x) is a by-product of this GEMM's staging
            ctx.x_rowmax = th.empty(x.shape[0], dtype=th.float32, device=x.device)
            y = gemm_x3(x, W, False, bias=b, relu=True, rowmax_out=ctx.x_rowmax)
        else:
            y = _mm_nt(x, W, b, relu=True)
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        n, C = dy.shape
        if (RELU_BWD_FUSED and ctx.needs_input_grad[2] and dy.is_cuda and dy.dtype == th.float32 and y.dtype == th.float32 and n > 0
                and C % 4 == 0 and dy.stride(1) == 1 and y.stride(1) == 1 and dy.stride(0) % 4 == 0 and y.stride(0) % 4 == 0
                and dy.stride(0) >= C and y.stride(0) >= C      # a row-broadcast gradient (stride 0) takes the torch path below
                and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0):
            # the mask and the bias gradient in ONE pass over the gradient (csrc/colsum.hip): autograd's threshold_backward + sum
            # are two (5.1 + 1.7 GB per C3 update on the time-batched encoder)
            S = _row_blocks(n)
            dym = th.empty((n, C), dtype=th.float32, device=dy.device)
            part = th.zeros((S, C), dtype=th.float32, device=dy.device)
            rowmax = None
            if C == 256 and ctx.needs_input_grad[0] and gemm_h2_supported(dym, W.shape[1], C) and W.stride(1) == 1:
                # the masked gradient's row maxima on the way: d x = dym W runs on the f16x2 kernel (csrc/gemm_h2.hip)
                rowmax = th.empty(n, dtype=th.float32, device=dy.device)
                L.check(L.lib().uavgnn_relu_bwd_colsum_rowmax(dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), dym.data_ptr(), C, n,
                                                              C, part.data_ptr(), S, rowmax.data_ptr(), L.stream()),
                        "uavgnn_relu_bwd_colsum_rowmax")
            else:
                L.check(L.lib().uavgnn_relu_bwd_colsum(dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), dym.data_ptr(), C, n, C,
                                                       part.data_ptr(), S, L.stream()), "uavgnn_relu_bwd_colsum")
            dx, dW, _ = _LinearSplitK._grads(ctx, x, W, dym, False, rowmax=rowmax, x_rowmax=ctx.x_rowmax)
            return (dx, dW, part.sum(0)) + (None,) * ctx.n_extra
        dy = th.ops.aten.threshold_backward(dy, y, 0.0)     # dy where y > 0 else 0, one pass (compare + where were two)
        return tuple(_LinearSplitK._grads(ctx, x, W, dy, True)) + (None,) * ctx.n_extra


def linear_relu(x, W, b):
    rm = getattr(x, "_uavgnn_rowmax", None)      # left by hetero_gatv2 on a time-batched launch
    if rm is not None:
        return _LinearReLU.apply(x, W, b, rm[0], rm[1], th.is_grad_enabled())
    return _LinearReLU.apply(x, W, b)


# ---------------------------------------------------------------------------------------------------------------------
# Fused recurrent step of the TarMAC agent: ONE autograd node per time step.

class WeightGradSink:
    """In-place accumulator for the weight / bias gradients of the recurrent step across the T+1 steps of a BPTT
    backward.  Every step adds its contribution INSIDE the weight-gradient GEMM (beta = 1, row chunks batched as in
    ``_LinearSplitK``), so the per-step ``sum`` over chunks, the per-step bias reduction result and autograd's
    per-parameter ``AccumulateGrad`` adds (12 parameters x 51 steps small kernels) disappear; ``flush()`` folds the
    accumulators into ``param.grad`` once, in a fixed order (deterministic)."""

    def __init__(self):
        self.seq = None      # time-batched staging of the sequence in flight (begin_sequence)
        self._seq_bufs = None
        self.slots = {}      # (key, chunk count) -> (buffer, flush_fn)
        # d h buffer the fused step's backward handed to autograd LAST (private), keyed by its address.  The entry holds a
        # strong reference: while it is alive the allocator cannot hand that address to another tensor, so an incoming
        # gradient with this data_ptr IS this storage (a bare set of addresses could match a recycled block).  At most
        # one entry: whatever the very next step does not consume is dropped.
        self.owned = {}

    @staticmethod
    def _chunks(n):
        S = 1
        while S < 64 and n % (2 * S) == 0 and n // (2 * S) >= 2048:
            S *= 2
        return S

    def weight(self, key, dy, x, flush_fn, allow_tn=True):
        """buffer[S, out, in] += chunked dy^T x"""
        if allow_tn and gemm_tn_x3_supported(dy, x):      # bf16x3 kernel: accumulates into its own [S, out, in] partials in place
            S = L.lib().uavgnn_gemm_tn_x3_chunks(x.shape[0], dy.shape[1], x.shape[1])
            key = (key, "tn", S)
            slot = self.slots.get(key)
            if slot is None:
                buf = th.zeros((S, dy.shape[1], x.shape[1]), dtype=th.float32, device=x.device)
                self.slots[key] = slot = (buf, flush_fn)
            gemm_tn_x3(dy, x, slot[0], accumulate=True)
            return
        n, S = x.shape[0], self._chunks(x.shape[0])
        key = (key, S)          # one slot per chunk count: a change of N mid-accumulation never drops a partial sum
        slot = self.slots.get(key)
        if slot is None:
            buf = th.zeros((S, dy.shape[1], x.shape[1]), dtype=th.float32, device=x.device)
            self.slots[key] = slot = (buf, flush_fn)
        if S == 1:
            slot[0][0].addmm_(dy.t(), x)
        else:
            slot[0].baddbmm_(dy.view(S, n // S, -1).transpose(1, 2), x.view(S, n // S, -1))

    def weight_h2(self, key, dy, x, bound_y, bound_x, flush_fn):
        """buffer[S, out, in] += chunked dy^T x on the f16x2 kernel (csrc/gemm_tn_h2.hip).  bound_y / bound_x: 0-dim device tensors, upper
        bounds of max |.| over ALL of dy / x - the column scales of the split (see gemm_tn_h2_supported)."""
        lib = L.lib()
        n, Mo, Ko = dy.shape[0], dy.shape[1], x.shape[1]
        S = lib.uavgnn_gemm_tn_h2_chunks(n, Mo, Ko)
        key = (key, "tnh2", S)
        slot = self.slots.get(key)
        acc = slot is not None
        if slot is None:
            self.slots[key] = slot = (th.empty((S, Mo, Ko), dtype=th.float32, device=x.device), flush_fn)
        cy, cx = bound_y.expand(Mo).contiguous(), bound_x.expand(Ko).contiguous()
        with KERNEL_TIMER.span("gemm_tn_h2", (n, Mo, Ko)):
            rc = lib.uavgnn_gemm_tn_h2(dy.data_ptr(), dy.stride(0), Mo, x.data_ptr(), x.stride(0), Ko, n, cy.data_ptr(), cx.data_ptr(),
                                       slot[0].data_ptr(), S, int(acc), L.stream())
        L.check(rc, "uavgnn_gemm_tn_h2")

    def bias(self, key, dy, flush_fn):
        """buffer[S, out] += row-blocked column sums of dy: one streaming HIP pass per matrix (csrc/colsum.hip), in
        place (torch: a reduction into a temporary plus an add per step; a transposed GEMV was 20x slower still)"""
        n, C = dy.shape
        S = _row_blocks(n)
        key = (key, S)
        slot = self.slots.get(key)
        if slot is None:
            self.slots[key] = slot = (th.zeros((S, C), dtype=th.float32, device=dy.device), flush_fn)
        if dy.dtype != th.float32 or dy.stride(1) != 1:
            dy = dy.float().contiguous()
        L.check(L.lib().uavgnn_colsum_acc(dy.data_ptr(), dy.stride(0), n, C, slot[0].data_ptr(), S, L.stream()),
                "uavgnn_colsum_acc")

    # ---- time-batched staging of ONE BPTT sequence -------------------------------------------------------------------
    # None of the weight / bias gradient reductions of the recurrent step depends on the recurrence: they are sums over
    # (time step, agent).  Inside begin_sequence() ... end_sequence() the fused step therefore does not reduce anything per
    # step: its forward writes [x || c] and h' into slots of [T1, N, .] buffers, its backward writes d_gi, d_gh, d_proj (and
    # copies dq) into slots of the same shape - the kernels that produce them write there directly, no extra traffic - and
    # end_sequence() issues ONE dy^T x per weight over T1 * N rows (1.67 M at C3: the shape class where csrc/gemm_tn_x3.hip
    # beats the vendor's split-K GEMM) and ONE column sum per bias: 5 + 4 launches per sequence instead of (5 + 4) * T1
    # (autograd of gnn_agents.py:243-246,:56 under learner.py:157).
    def begin_sequence(self, T1, N, x_all):
        """x_all: the time-major [T1 * N, H] input of the T1 recurrent steps (the time-batched encoder's output)."""
        self.seq = _SequenceStage(T1, N, x_all.detach(), getattr(self, "_seq_bufs", None))
        self._seq_bufs = self.seq.bufs          # the buffers are reused by the next sequence (chunks of one accumulate)

    def end_sequence(self):
        seq, self.seq = getattr(self, "seq", None), None
        if seq is None or not seq.bwd_steps:
            return
        T1, N = seq.T1, seq.N
        full = sorted(seq.bwd_steps) == list(range(T1)) and seq.t_fwd == T1
        spans = [(0, T1)] if full else [(t, t + 1) for t in sorted(seq.bwd_steps)]
        split, ids, H = seq.split, seq.ids, seq.H
        for (t0, t1) in spans:
            rows = lambda name, lo=0: seq.bufs[name][t0 + lo:t1 + lo].reshape((t1 - t0) * N, -1)   # noqa: E731
            d_proj, d_gi, d_gh = rows("d_proj"), rows("d_gi"), rows("d_gh")
            dq = seq.dq_rows(t0, t1)
            x = seq.x_all[t0 * N:t1 * N]
            h, h2, inp = rows("h"), rows("h", 1), rows("inp")
            # the vendor's batched split-K fp32 GEMM (64 row chunks): at these shapes - 768-, 96- and 9-row outputs over
            # 1.67 M rows - it runs at 135-141 TFLOP/s against 85-107 for csrc/gemm_tn_x3.hip (tools/gemm_tn_big_probe.py),
            # and 25-30 % above its own per-step rate (32 768 rows per call)
            tn = False
            h2_bounds = full and seq.rowmax_steps >= set(range(T1)) and seq.rm_g_steps >= set(range(T1))
            if h2_bounds and seq.rm_p_steps >= set(range(T1)) and gemm_tn_h2_supported(x, d_proj, 96) and gemm_tn_h2_supported(h, d_proj, 96):
                # dWp on the f16x2 kernel with the operands swapped - x^T d_proj = (dWp_x)^T, [H x 96]: one 256 x 128 tile three quarters full
                # (d_proj^T x would fill three eighths of two: slower than the vendor) - 0.93 -> 0.5 ms each against the vendor's 0.74;
                # column bounds: the message kernel's row maxima bound x and h, the per-step row maxima of d_proj (taken for the d x
                # product of the step) bound d_proj
                bxh, bpj = _max_two_stage(seq.bufs["rowmax"][:T1]), _max_two_stage(seq.bufs["rm_p"][:T1])
                self.weight_h2(("Wp_x", ids["Wp"]), x, d_proj, bxh, bpj, lambda g: split("Wp", g.t(), 0))
                self.weight_h2(("Wp_h", ids["Wp"]), h, d_proj, bxh, bpj, lambda g: split("Wp", g.t(), H))
            else:
                self.weight(("Wp_x", ids["Wp"]), d_proj, x, lambda g: split("Wp", g, 0), tn)
                self.weight(("Wp_h", ids["Wp"]), d_proj, h, lambda g: split("Wp", g, H), tn)
            self.bias(("bp", ids["Wp"]), d_proj, lambda g: split("bp", g, 0))
            if h2_bounds and gemm_tn_h2_supported(d_gi, inp) and gemm_tn_h2_supported(d_gh, h):
                # dW_ih / dW_hh on the f16x2 kernel (177-187 TFLOP/s against the vendor's 131-143 at C3): the column scales of the split
                # come for free - the row maxima the message kernel (max over [x || c || h] per agent) and the gate kernel (d_gi / d_gh) left
                # for the f16x2 cell / input-gradient products bound every column; a column far below the global maximum is held with fewer
                # bits (absolute error <= 2^-39 of the bound per element, averaged down over 10^6 rows: DESIGN.md section 5)
                bx, by = _max_two_stage(seq.bufs["rowmax"][:T1]), _max_two_stage(seq.bufs["rm_g"][:T1])
                self.weight_h2(("W_ih", ids["W_ih"]), d_gi, inp, by, bx, lambda g: split("W_ih", g, 0))
                self.weight_h2(("W_hh", ids["W_hh"]), d_gh, h, by, bx, lambda g: split("W_hh", g, 0))
            else:
                self.weight(("W_ih", ids["W_ih"]), d_gi, inp, lambda g: split("W_ih", g, 0), tn)
                self.weight(("W_hh", ids["W_hh"]), d_gh, h, lambda g: split("W_hh", g, 0), tn)
            if all(t in seq.gsum_steps for t in range(t0, t1)):
                # the gate kernels of these steps left per-workgroup column sums d_r | d_z | d_n (input) | d_n (hidden): [., 4H]
                gs = seq.bufs["gsum"][t0:t1]
                gs = gs.reshape(gs.shape[0] * gs.shape[1], 4 * H)
                self.bias(("b_ih", ids["W_ih"]), gs[:, :3 * H], lambda g: split("b_ih", g, 0))
                self.bias(("b_hh_n", ids["W_hh"]), gs[:, 3 * H:], lambda g: split("b_hh", g, 2 * H))
            else:
                self.bias(("b_ih", ids["W_ih"]), d_gi, lambda g: split("b_ih", g, 0))
                self.bias(("b_hh_n", ids["W_hh"]), d_gh[:, 2 * H:], lambda g: split("b_hh", g, 2 * H))
            self.weight(("W_out", ids["W_out"]), dq, h2, lambda g: split("W_out", g, 0), tn)
            self.bias(("b_out", ids["W_out"]), dq, lambda g: split("b_out", g, 0))

    def flush(self):
        self.end_sequence()
        for key, (buf, fn) in self.slots.items():
            if fn is not None:
                fn(buf.sum(0))
        self.slots = {}
        self.owned.clear()
        self._seq_bufs = None


class _SequenceStage:
    """Slots of the time-batched buffers of one BPTT sequence (WeightGradSink.begin_sequence)."""

    def __init__(self, T1, N, x_all, bufs=None):
        self.T1, self.N, self.x_all = T1, N, x_all
        self.bufs = bufs if bufs is not None else {}
        self.t_fwd = 0
        self.bwd_steps = []
        self.gsum_steps = set()          # steps whose gate kernel wrote its column-sum partials ("gsum" slots)
        self.rowmax_steps = set()        # steps whose message kernel left the row maxima of [x || c || h] ("rowmax" slots)
        self.rm_g_steps = set()          # steps whose gate kernel left the row maxima of d_gi / d_gh ("rm_g" slots)
        self.rm_p_steps = set()          # steps whose backward took the row maxima of d_proj ("rm_p" slots)
        self.dq_steps = {}               # step -> the gradient of the step's Q values as autograd handed it over
        self.split = self.ids = self.H = None

    def dq_rows(self, t0, t1):
        """[(t1 - t0) N, A] gradient of the Q values of steps t0 .. t1 - 1.  The per-step gradients autograd hands to the recurrent step are
        normally consecutive [N, A] slices of ONE buffer (the backward of the learner's stack of the per-step Q values): then the rows are
        a view of it - rounds 4-5 copied every step into a staging slot (51 launches of 4.7 us per update); anything else is gathered by
        one torch.cat."""
        parts = [self.dq_steps[t] for t in range(t0, t1)]
        if len(parts) == 1:
            return parts[0] if parts[0].is_contiguous() else parts[0].contiguous()
        first, (N, A) = parts[0], parts[0].shape
        if all(p.is_contiguous() and p.shape == (N, A) and p.dtype == first.dtype and
               p.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() and
               p.storage_offset() == first.storage_offset() + i * N * A for i, p in enumerate(parts)):
            return first.as_strided(((t1 - t0) * N, A), (A, 1), first.storage_offset())
        return th.cat(parts, 0)

    def slot(self, name, t, cols, extra=0, rows=None):
        """[N, cols] slot t of the [T1 + extra, N, cols] buffer `name` (``rows``: another row count than N per step)."""
        n = self.N if rows is None else rows
        b = self.bufs.get(name)
        if b is None or b.shape != (self.T1 + extra, n, cols) or b.device != self.x_all.device:
            b = self.bufs[name] = th.empty((self.T1 + extra, n, cols), dtype=th.float32, device=self.x_all.device)
        return b[t]


def _wgrad(dy, x):
    """dy^T x with the reduction over rows split into chunks (see _LinearSplitK)."""
    if gemm_tn_x3_supported(dy, x):
        return gemm_tn_x3(dy, x).sum(0)
    n, S = x.shape[0], WeightGradSink._chunks(x.shape[0])
    if S == 1:
        return th.mm(dy.t(), x)
    return th.bmm(dy.view(S, n // S, -1).transpose(1, 2), x.view(S, n // S, -1)).sum(0)


GRAD_SINK = None   # set by the learner around loss.backward(); None -> gradients are returned to autograd as usual
SEQ_STAGING = os.environ.get("UAVGNN_SEQ_STAGING", "1") != "0"   # weight / bias gradients of a BPTT sequence reduced once (A/B switch)


def _add_grad(p, g):
    if p.grad is None:
        p.grad = g.clone()
    else:
        p.grad.add_(g)


def _add_grad_cols(p, g, start):
    if p.grad is None:
        p.grad = th.zeros_like(p)
    p.grad[start:start + g.shape[0]].add_(g)


class _TimeSplit(th.autograd.Function):
    """x_all [T1 * N, H] -> T1 per-step views, with a gradient buffer the steps' backward passes write INTO.

    ``unbind`` would make autograd stack the T1 step gradients at the end (1.7 GB of copies for C3, T = 50); here the
    [T1, N, H] buffer is allocated up front, ``slots[t]`` (its t-th slice) is handed to the consumer of x_t (the fused
    recurrent step writes d x_t with ``out=``) and the backward of the split returns the buffer as it is when every
    incoming gradient is its own slice - otherwise it falls back to copying the strays in."""

    @staticmethod
    def forward(ctx, x_all, T1, holder):
        N = x_all.shape[0] // T1
        buf = th.empty((T1, N, x_all.shape[1]), dtype=x_all.dtype, device=x_all.device)
        holder.append(buf)
        ctx.buf = buf
        return tuple(x_all.detach().view(T1, N, -1).unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        buf = ctx.buf
        for t, g in enumerate(grads):
            if g is None:
                buf[t].zero_()
            elif g.data_ptr() != buf[t].data_ptr() or g.stride() != buf[t].stride():
                buf[t].copy_(g)
        return buf.view(-1, buf.shape[2]), None, None


def time_split(x_all, T1):
    """(xs, slots): per-step views of the time-batched encoder output and the slices of its gradient buffer."""
    if not (th.is_grad_enabled() and x_all.requires_grad):
        return x_all.view(T1, x_all.shape[0] // T1, -1).unbind(0), [None] * T1
    holder = []
    xs = _TimeSplit.apply(x_all, T1, holder)
    return xs, list(holder[0].unbind(0))


MSG_FUSED = os.environ.get("UAVGNN_MSG_FUSED", "1") != "0"   # K3a + K3b in one launch (csrc/tarmac_msg.hip); A/B switch


def _tarmac_msg_plan(x, h, Wp, bp, M, K, env, N, H):
    """(weight tiles, agents per graph) when the fused K3a + K3b launch covers this call, else None: graphs with a uniform number
    of agents (every graph <= n_max and N == B n_max), aligned row-major operands, a contiguous stacked projection weight."""
    if not MSG_FUSED or env is None or N == 0:
        return None
    _, B, n_ag = env
    if N != B * n_ag or (H + M) % 4 or not L.lib().uavgnn_tarmac_msg_supported(H, M, K, n_ag):
        return None
    if (Wp.shape != (M + 2 * K, 2 * H) or not Wp.is_contiguous() or Wp.data_ptr() % 16 or not bp.is_contiguous()
            or x.data_ptr() % 16 or h.data_ptr() % 16 or x.dtype != th.float32 or Wp.dtype != th.float32):
        return None
    lib = L.lib()
    tiles = _cached_planes(("msgw", Wp.data_ptr(), Wp._version, H, M, K), lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), x.device,
                           lambda buf: L.check(lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), Wp.stride(0), H, M, K, buf.data_ptr(),
                                                                            L.stream()), "uavgnn_tarmac_msg_prepare"),
                           keep=(Wp,))
    return tiles, n_ag


def _launch_tarmac_msg(msg, x, h, bp, M, K, talk_off, talk_src, N, H, c_ptr, ld_c, a_save, proj, ld_p, x_copy, ld_xc, planes=None,
                       rowmax=None):
    """rowmax [N] (optional): receives max(|x_row|, |c_row|, |h_row|) - the row scales of the f16x2 GRU cell behind this launch."""
    tiles, n_ag = msg
    if rowmax is not None:
        with KERNEL_TIMER.span("tarmac_msg_fwd", (N, H, M, K, int(proj is not None), int(x_copy is not None))):
            rc = L.lib().uavgnn_tarmac_msg_fwd_rowmax(x.data_ptr(), x.stride(0), h.data_ptr(), h.stride(0), N, H, n_ag, tiles.data_ptr(),
                                                      bp.data_ptr(), M, K, L.ptr(talk_off), L.ptr(talk_src), 1.0 / K, c_ptr, ld_c, a_save,
                                                      proj, ld_p, x_copy, ld_xc, rowmax.data_ptr(), L.stream())
        L.check(rc, "uavgnn_tarmac_msg_fwd_rowmax")
        return
    with KERNEL_TIMER.span("tarmac_msg_fwd", (N, H, M, K, int(proj is not None), int(x_copy is not None))):
        rc = L.lib().uavgnn_tarmac_msg_fwd(x.data_ptr(), x.stride(0), h.data_ptr(), h.stride(0), N, H, n_ag, tiles.data_ptr(),
                                           bp.data_ptr(), M, K, L.ptr(talk_off), L.ptr(talk_src), 1.0 / K, c_ptr, ld_c, a_save, proj,
                                           ld_p, x_copy, ld_xc, L.ptr(planes), L.stream())
    L.check(rc, "uavgnn_tarmac_msg_fwd")


HEAD_KERNEL = os.environ.get("UAVGNN_HEAD_KERNEL", "1") != "0"   # the Q head on csrc/head.hip (one pass over h'); 0: vendor GEMM


def head_fwd(h, W_out, b_out):
    """q = h W_out^T + b_out (gnn_agents.py:56), no autograd: csrc/head.hip for n_actions <= 16, else the vendor GEMM."""
    N, H = h.shape
    A = W_out.shape[0]
    if (HEAD_KERNEL and h.is_cuda and h.dtype == th.float32 and N > 0 and h.stride(1) == 1 and W_out.stride(1) == 1
            and h.stride(0) % 4 == 0 and W_out.stride(0) % 4 == 0 and h.data_ptr() % 16 == 0 and W_out.data_ptr() % 16 == 0
            and b_out.is_contiguous() and L.lib().uavgnn_head_supported(H, A)):
        q = th.empty((N, A), dtype=th.float32, device=h.device)
        with KERNEL_TIMER.span("head_fwd", (N, H, A)):
            rc = L.lib().uavgnn_head_fwd(h.data_ptr(), h.stride(0), N, H, W_out.data_ptr(), W_out.stride(0), b_out.data_ptr(), A,
                                         q.data_ptr(), A, L.stream())
        L.check(rc, "uavgnn_head_fwd")
        return q
    return th.addmm(b_out, h, W_out.t())


class _TarmacStep(th.autograd.Function):
    """q, h' = head(GRU([x || c], h)), c = targeted attention over `talk` of the projections of [x || stopgrad(h)]
    (gnn_agents.py:248-271 with n_rounds = 1, then :56).  Forward: 5 vendor GEMMs + K3b + K4, the projection of the two
    halves accumulated in place, c written by K3b straight into the GRU input buffer (no cat, no add kernels).
    Backward: hand-written; weight gradients go to ``GRAD_SINK`` when one is active."""

    @staticmethod
    def forward(ctx, x, h, Wp, bp, W_ih, b_ih, W_hh, b_hh, W_out, b_out, M, K, talk_off, talk_src, t_off, t_dst, t_pos,
                split, env=None, dx_out=None, train=True):
        L.require_gpu(x, h, Wp, W_ih, talk_off)
        N, H = x.shape
        x, h = L.f32c(x), L.f32c(h)
        # K3a + K3b as ONE launch (csrc/tarmac_msg.hip) when the batch is made of graphs with a uniform number of agents: the
        # projections stay on the chip on no-grad calls (two vendor GEMMs + K3b: 52 us per C3 step; traffic floor ~13 us)
        msg = _tarmac_msg_plan(x, h, Wp, bp, M, K, env, N, H)
        if msg is not None:
            proj = None
        elif gemm_x3_supported(x, Wp.shape[0], H) and gemm_x3_supported(h, Wp.shape[0], H) and bp.is_contiguous():
            proj = gemm_x3(x, Wp[:, :H], bias=bp)
            gemm_x3(h, Wp[:, H:], out=proj, accumulate=True)
        else:
            proj = th.addmm(bp, x, Wp[:, :H].t())
            proj.addmm_(h, Wp[:, H:].t())                                 # [N, M + 2K]: value | signature | query
        E = talk_src.shape[0]
        a_save = th.empty(max(E, 1), dtype=th.float32, device=x.device)
        ld = M + 2 * K
        ctx.seq = ctx.seq_t = None
        aligned = all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (W_ih, W_hh))
        # the f16x2 cell needs the row maxima of [x || c || h]: the fused message launch writes them on its way
        rowmax = None
        if (msg is not None and aligned and GRU_FUSED and gru_cell_h2_supported(H + M, H, N, max(x.stride(0), H + M))
                and H % 32 == 0 and M % 32 == 0 and L.lib().uavgnn_tarmac_msg_rowmax_supported(H, M, K, msg[1])):
            rowmax = th.empty(N, dtype=th.float32, device=x.device)
        c_only = None
        if not train and aligned and h.shape[0] >= GRU_FUSED_MIN_ROWS and GRU_FUSED:
            c_only = th.empty((N, M), dtype=th.float32, device=x.device)
            if not gru_cell_two_piece_supported(x, c_only, h):
                c_only = None
        if c_only is not None:
            # no-grad call (rollout, target network): nothing keeps [x || c] for a backward, so K3b writes c alone and the cell
            # reads its input from the two buffers - the 2 x 4 H bytes per agent of the concatenating copy disappear
            if msg is not None:
                _launch_tarmac_msg(msg, x, h, bp, M, K, talk_off, talk_src, N, H, c_only.data_ptr(), M, None, None, 0, None, 0,
                                   rowmax=rowmax)
            else:
                _launch_talk_fwd(env, proj.data_ptr() + 4 * M, ld, proj.data_ptr() + 4 * (M + K), ld, proj.data_ptr(), ld, K, M,
                                 talk_off, talk_src, N, 1.0 / K, c_only.data_ptr(), M, a_save.data_ptr(), None, 0, 0)
            h2, _ = _gru_cell_launch(x, h, W_ih, b_ih, W_hh, b_hh, save=False, inp2=c_only, rowmax=rowmax)
            inp, fused = c_only, True
            gi = gh = h2                                           # placeholders keep save_for_backward's arity
        else:
            # inside a staged BPTT sequence (WeightGradSink.begin_sequence) [x || c] and h' are written into the slots of
            # step t of the time-batched buffers the weight gradients are reduced from at the end of the sequence
            seq = GRAD_SINK.seq if (train and GRAD_SINK is not None) else None
            if seq is not None and (seq.t_fwd >= seq.T1 or N != seq.N or seq.x_all.shape[1] != H or
                                    x.data_ptr() != seq.x_all.data_ptr() + 4 * H * N * seq.t_fwd):
                seq = None           # not a step of the sequence that was announced (or more steps than announced)
            if seq is not None and seq.t_fwd > 0 and h.data_ptr() != seq.slot("h", seq.t_fwd, H, extra=1).data_ptr():
                # end_sequence() reduces W_hh / Wp_h / W_out from the h SLOTS: a caller that masks or copies h between two steps
                # hands over another tensor than the slot the previous step wrote - this step (and, through the x check above,
                # every later one) reduces per step instead
                seq = None
            if seq is not None:
                seq_t = seq.t_fwd
                seq.t_fwd += 1
                inp = seq.slot("inp", seq_t, H + M)
                if seq_t == 0:
                    seq.slot("h", 0, H, extra=1).copy_(h)
            else:
                inp = th.empty((N, H + M), dtype=th.float32, device=x.device)     # [x || c], both halves filled by K3b
            if seq is not None and rowmax is not None:       # kept for the sequence: its maximum scales the weight-gradient products
                rowmax = seq.slot("rowmax", seq_t, 1).view(N)
                seq.rowmax_steps.add(seq_t)
            if msg is not None:     # proj, the attention weights and the x half of [x || c] are the launch's training outputs
                proj = th.empty((N, ld), dtype=th.float32, device=x.device)
                _launch_tarmac_msg(msg, x, h, bp, M, K, talk_off, talk_src, N, H, inp.data_ptr() + 4 * H, H + M, a_save.data_ptr(),
                                   proj.data_ptr(), ld, inp.data_ptr(), H + M, rowmax=rowmax)
            else:
                _launch_talk_fwd(env, proj.data_ptr() + 4 * M, ld, proj.data_ptr() + 4 * (M + K), ld, proj.data_ptr(), ld, K, M,
                                 talk_off, talk_src, N, 1.0 / K, inp.data_ptr() + 4 * H, H + M, a_save.data_ptr(), x.data_ptr(),
                                 x.stride(0), H)
            fused = gru_cell_supported(inp, h) and aligned
            if fused:      # K4 in one launch: gi / gh never reach HBM; training forwards keep the [N, 4H] pre-activation sets
                h2, pre = _gru_cell_launch(inp, h, W_ih, b_ih, W_hh, b_hh, save=bool(train),
                                           h2_out=None if seq is None else seq.slot("h", seq_t + 1, H, extra=1), rowmax=rowmax)
                if seq is not None and pre is not None:
                    ctx.seq, ctx.seq_t = seq, seq_t
                gi = gh = pre if pre is not None else h2           # placeholders keep save_for_backward's arity
            else:
                gi = th.addmm(b_ih, inp, W_ih.t())
                gh = th.addmm(b_hh, h, W_hh.t())
                h2 = th.empty_like(h)
                with KERNEL_TIMER.span("gru_gates_fwd"):
                    rc = L.lib().uavgnn_gru_gates_fwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), N, H, h2.data_ptr(), L.stream())
                L.check(rc, "uavgnn_gru_gates_fwd")
            # (an accepted step that ran unfused wrote h' into a tensor of its own, not into slot t + 1: the next step's h then
            # fails the slot check above and reduces per step, like this one - ctx.seq stays None)
        q = head_fwd(h2, W_out, b_out)
        if proj is None:
            proj = h2                                              # no-grad call through the fused message launch: placeholder
        ctx.dims = (M, K)
        ctx.split, ctx.env, ctx.dx_out, ctx.fused_gru = split, env, dx_out, fused
        ctx.have_pre = bool(train) or not fused
        ctx.save_for_backward(x, h, proj, inp, gi, gh, h2, a_save, Wp, W_ih, W_hh, W_out, talk_off, talk_src, t_off,
                              t_dst, t_pos)
        return q, h2

    @staticmethod
    def backward(ctx, dq, dh2):
        (x, h, proj, inp, gi, gh, h2, a_save, Wp, W_ih, W_hh, W_out, talk_off, talk_src, t_off, t_dst,
         t_pos) = ctx.saved_tensors
        M, K = ctx.dims
        N, H = x.shape
        if not ctx.have_pre:    # gi would be the [N, H] placeholder: reading 4H floats per row from it is out of bounds
            raise L.UavGnnError("tarmac_step: backward through a forward that saved no pre-activations (train=False)")
        sink = GRAD_SINK
        rm_g = None
        dq = L.f32c(dq) if dq is not None else th.zeros((N, W_out.shape[0]), dtype=th.float32, device=x.device)
        # d h' = d_hout + dq W_out inside the gate kernel when the cell ran fused (its pre-activation sets are what that kernel reads)
        head = None
        if (HEAD_FUSED_BWD and ctx.fused_gru and H % 4 == 0 and W_out.shape[0] <= 64 and W_out.is_contiguous()
                and W_out.dtype == th.float32 and W_out.data_ptr() % 16 == 0
                and (dh2 is None or (dh2.is_contiguous() and dh2.dtype == th.float32 and dh2.shape == (N, H)))):
            head = (dq, W_out.detach())
            dh2_tot = dh2
            if sink is not None:
                sink.owned.clear()
        elif dh2 is None:
            dh2_tot = th.mm(dq, W_out)
        elif (sink is not None and dh2.is_contiguous() and dh2.dtype == th.float32 and dh2.shape == (N, H)
              and sink.owned.get(dh2.data_ptr()) is not None):
            # inside the learner's BPTT the incoming d h' is the buffer the NEXT step's backward of this very op
            # allocated and returned (registered in sink.owned and kept alive there): nobody else holds it, so accumulate
            # in place instead of copying 33 MB into a fresh output first.  Any other gradient tensor (a hook's,
            # retain_grad's, one autograd summed from several consumers) lives at another address and is left untouched.
            sink.owned.clear()
            dh2_tot = dh2.addmm_(dq, W_out)
        else:
            dh2_tot = th.addmm(dh2, dq, W_out)
        seq = ctx.seq if (sink is not None and ctx.seq is not None and sink.seq is ctx.seq) else None
        if seq is not None:      # staged sequence: the gate gradients go straight into the time-batched buffers
            t = ctx.seq_t
            G = L.lib().uavgnn_gru_gates_bwd_sum_rows(N, H) if GATE_SUMS else 0
            sums = None
            if G and (head is not None or dh2_tot is not None):
                # the step's share of db_ih / db_hh comes out of the gate kernel: end_sequence() sums [T1 G, 4H] partials instead of
                # streaming the [T1 N, 3H] gate gradients twice more
                sums = seq.slot("gsum", t, 4 * H, rows=G)
                seq.gsum_steps.add(t)
            rm_g = None
            if (sums is not None and H == 256 and W_hh.stride(1) == 1 and gemm_h2_supported(seq.slot("d_gh", t, 3 * H), H, 3 * H)):
                rm_g = seq.slot("rm_g", t, 1).view(N)      # row maxima of d_gi / d_gh for the f16x2 products below (and, over the whole
                seq.rm_g_steps.add(t)                      # sequence, for the weight gradients)
            d_gi, d_gh, dh = _gru_gates_bwd_from_pre(gi, h, dh2_tot, seq.slot("d_gi", t, 3 * H), seq.slot("d_gh", t, 3 * H),
                                                     head=head, sums=sums, rowmax=rm_g)
        elif ctx.fused_gru:
            d_gi, d_gh, dh = _gru_gates_bwd_from_pre(gi, h, dh2_tot, head=head)      # gi holds the saved pre-activation sets
        else:
            d_gi, d_gh, dh = th.empty_like(gi), th.empty_like(gh), th.empty_like(h)
            with KERNEL_TIMER.span("gru_gates_bwd"):
                rc = L.lib().uavgnn_gru_gates_bwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), dh2_tot.data_ptr(), N, H,
                                                  d_gi.data_ptr(), d_gh.data_ptr(), dh.data_ptr(), L.stream())
            L.check(rc, "uavgnn_gru_gates_bwd")
        # d_inp = d_gi W_ih is [N, H + M]: d x | d c.  When the bf16x3 GEMM has the H-column shape, the two halves are separate
        # products: d x lands where the caller wants it (the slice of the time-split gradient buffer) and the projection term is
        # accumulated into it in place - no [N, H] copy in front of the addmm, and 256 of the 320 columns leave the vendor's
        # 64 x 32-tile solution (134 us) for the matrix-core kernel (A/B: UAVGNN_DINP_SPLIT=0)
        split_dinp = (DINP_SPLIT and M > 0 and W_ih.stride(1) == 1 and gemm_x3_supported(d_gi, H, W_ih.shape[0])
                      and (ctx.dx_out is None or (ctx.dx_out.stride(1) == 1 and ctx.dx_out.dtype == th.float32)))
        ld = M + 2 * K
        d_proj = th.empty((N, ld), dtype=th.float32, device=x.device) if seq is None else seq.slot("d_proj", ctx.seq_t, ld)
        dx_cat = False
        if split_dinp:
            dx = ctx.dx_out if ctx.dx_out is not None else th.empty((N, H), dtype=th.float32, device=x.device)
            # d x = d_gi W_ih[:, :H] + d_proj Wp[:, :H]: ONE product over [d_gi || d_proj] once d_proj exists (behind the attention
            # backward) instead of a product + an accumulating vendor GEMM that re-reads and re-writes d x (31 us per step)
            dx_cat = Wp.stride(1) == 1 and gemm_x3_cat_supported(d_gi, d_proj, H)
            if not dx_cat:
                _mm_nn(d_gi, W_ih[:, :H], out=dx)
            d_c = th.mm(d_gi, W_ih[:, H:])                                 # [N, M]
            d_c_ptr, d_c_ld = d_c.data_ptr(), M
        else:
            d_inp = _mm_nn(d_gi, W_ih)
            d_c_ptr, d_c_ld = d_inp.data_ptr() + 4 * H, H + M
        _mm_nn(d_gh, W_hh, out=dh, accumulate=True, rowmax=rm_g)
        if sink is not None:
            sink.owned.clear()
            sink.owned[dh.data_ptr()] = dh
        _launch_talk_bwd(ctx.env, proj.data_ptr() + 4 * M, ld, proj.data_ptr() + 4 * (M + K), ld, proj.data_ptr(), ld,
                         K, M, talk_off, talk_src, (t_off, t_dst, t_pos), N, 1.0 / K, a_save, d_c_ptr,
                         d_c_ld, d_proj.data_ptr() + 4 * M, ld, d_proj.data_ptr() + 4 * (M + K), ld, d_proj.data_ptr(),
                         ld)
        if dx_cat and rm_g is not None and gemm_h2_supported(d_gi, H, d_gi.shape[1] + d_proj.shape[1]):
            # f16x2: the row scale of [d_gi || d_proj] is the larger of the gate kernel's bound and d_proj's own, which the launch takes
            # itself (96 columns per row, read once more by its workgroups: a microsecond against a 7.8-us pass of uavgnn_row_absmax)
            if seq is not None:      # kept for the sequence: the column bound of d_proj in the weight gradient dWp (end_sequence)
                rm_p = seq.slot("rm_p", ctx.seq_t, 1).view(N)
                seq.rm_p_steps.add(ctx.seq_t)
            else:
                rm_p = th.empty(N, dtype=th.float32, device=d_gi.device)
            if GEMM_H2_RM2:
                gemm_h2(d_gi, W_ih[:, :H], rm_g, True, out=dx, a2=d_proj, W2=Wp[:, :H], rowmax2_out=rm_p)
            else:
                gemm_h2(d_gi, W_ih[:, :H], rm_g, True, out=dx, a2=d_proj, W2=Wp[:, :H], rowmax2=row_absmax(d_proj, out=rm_p))
        elif dx_cat:
            gemm_x3_cat(d_gi, d_proj, W_ih[:, :H], Wp[:, :H], dx)
        elif split_dinp:
            dx.addmm_(d_proj, Wp[:, :H])                                   # h enters the projections stop-gradded
        else:
            dx = ctx.dx_out                                                # slice of the time-split gradient buffer
            if dx is not None:
                th.addmm(d_inp[:, :H], d_proj, Wp[:, :H], out=dx)
            else:
                dx = th.addmm(d_inp[:, :H], d_proj, Wp[:, :H])
        if seq is not None:      # reduced once per sequence (WeightGradSink.end_sequence)
            seq.dq_steps[ctx.seq_t] = dq      # (a reference, no copy: see _SequenceStage.dq_rows)
            seq.bwd_steps.append(ctx.seq_t)
            seq.split, seq.H = ctx.split, H
            seq.ids = {"Wp": id(Wp), "W_ih": id(W_ih), "W_hh": id(W_hh), "W_out": id(W_out)}
            gWp = gbp = gWih = gbih = gWhh = gbhh = gWo = gbo = None
        elif sink is not None:
            split = ctx.split
            sink.weight(("Wp_x", id(Wp)), d_proj, x, lambda g: split("Wp", g, 0))
            sink.weight(("Wp_h", id(Wp)), d_proj, h, lambda g: split("Wp", g, H))
            sink.bias(("bp", id(Wp)), d_proj, lambda g: split("bp", g, 0))
            sink.weight(("W_ih", id(W_ih)), d_gi, inp, lambda g: split("W_ih", g, 0))
            sink.bias(("b_ih", id(W_ih)), d_gi, lambda g: split("b_ih", g, 0))
            sink.weight(("W_hh", id(W_hh)), d_gh, h, lambda g: split("W_hh", g, 0))
            # d_gh == d_gi on the r and z columns: only the n block of b_hh needs its own pass over d_gh
            sink.bias(("b_hh_n", id(W_hh)), d_gh[:, 2 * H:], lambda g: split("b_hh", g, 2 * H))
            sink.weight(("W_out", id(W_out)), dq, h2, lambda g: split("W_out", g, 0))
            sink.bias(("b_out", id(W_out)), dq, lambda g: split("b_out", g, 0))
            gWp = gbp = gWih = gbih = gWhh = gbhh = gWo = gbo = None
        else:
            gWp = th.cat((_wgrad(d_proj, x), _wgrad(d_proj, h)), 1)
            gbp = _colsum(d_proj)
            gWih, gbih = _wgrad(d_gi, inp), _colsum(d_gi)
            gWhh, gbhh = _wgrad(d_gh, h), th.cat((gbih[:2 * H], _colsum(d_gh[:, 2 * H:])))
            gWo, gbo = _wgrad(dq, h2), _colsum(dq)
        return (dx, dh, gWp, gbp, gWih, gbih, gWhh, gbhh, gWo, gbo) + (None,) * 11


def tarmac_step(x, h, g, comm, f_out, stacked=None, dx_out=None):
    """comm: the TarMAC module (f_val / f_sign / f_que / f_udt), f_out: nn.Linear head.  Returns (q, h').
    ``stacked`` = (Wp, bp) built WITH autograd history when no GRAD_SINK will collect the weight gradients."""
    M, K = comm._msg_size, comm._key_size
    Wp, bp = stacked if stacked is not None else comm.fused_projection()
    off, src = g.talk_csc()
    cell = comm.f_udt
    env = _talk_env(g, M, K)
    t_off, t_dst, t_pos = ((None, None, None) if env is not None else
                           _talk_transpose_if_needed(g, x, h, Wp, cell.weight_ih, f_out.weight))
    H = cell.weight_hh.shape[1]
    params = {"W_ih": cell.weight_ih, "b_ih": cell.bias_ih, "W_hh": cell.weight_hh, "b_hh": cell.bias_hh,
              "W_out": f_out.weight, "b_out": f_out.bias}

    def split(name, grad, col0):
        """flush target of the sink: distribute an accumulated gradient to the module's parameters."""
        if name == "Wp":        # rows: f_val | f_sign | f_que ; columns col0 .. col0 + grad.shape[1]
            r = 0
            for lin in (comm.f_val, comm.f_sign, comm.f_que):
                rows = lin.weight.shape[0]
                if lin.weight.grad is None:
                    lin.weight.grad = th.zeros_like(lin.weight)
                lin.weight.grad[:, col0:col0 + grad.shape[1]].add_(grad[r:r + rows])
                r += rows
        elif name == "bp":
            r = 0
            for lin in (comm.f_val, comm.f_sign, comm.f_que):
                rows = lin.bias.shape[0]
                _add_grad(lin.bias, grad[r:r + rows])
                r += rows
        elif name == "b_ih":    # its r and z columns are b_hh's too (see _TarmacStep.backward)
            _add_grad(params["b_ih"], grad)
            _add_grad_cols(params["b_hh"], grad[:2 * H], 0)
        elif name == "b_hh":    # the n block
            _add_grad_cols(params["b_hh"], grad, col0)
        else:
            _add_grad(params[name], grad)

    # ANY differentiable input (a trainable Q head over a frozen encoder / GRU included) makes autograd run the backward,
    # which reads the saved pre-activation sets
    train = th.is_grad_enabled() and any(t.requires_grad for t in (x, h, Wp, bp, cell.weight_ih, cell.bias_ih,
                                                                   cell.weight_hh, cell.bias_hh, f_out.weight, f_out.bias))
    return _TarmacStep.apply(x, h, Wp, bp, cell.weight_ih, cell.bias_ih, cell.weight_hh, cell.bias_hh, f_out.weight,
                             f_out.bias, M, K, off, src, t_off, t_dst, t_pos, split, env, dx_out, train)


class _DiscComm(th.autograd.Function):
    """K5.  Hard Gumbel-softmax messages (straight-through) + OR (max) aggregation of DiscreteComm."""

    @staticmethod
    def forward(ctx, logits, gumbel, talk_off, talk_src, t_off, t_dst, t_pos, inv_tau, rng=None):
        L.require_gpu(logits, gumbel, talk_off, talk_src, rng)
        logits = L.f32c(logits)
        N, M2 = logits.shape
        M = M2 // 2
        E = talk_src.shape[0]
        if gumbel is not None:
            gumbel = L.f32c(gumbel)
            if gumbel.numel() != E * M2:
                raise L.UavGnnError(f"disc_comm: gumbel noise must be [E={E}, msg={M}, 2]")
        elif rng is None or rng.dtype != th.int64 or rng.numel() != 2:
            raise L.UavGnnError("disc_comm: either the Gumbel noise or a device int64 {seed, step} pair is required")
        c = th.empty((N, M2), dtype=th.float32, device=logits.device)
        y0 = th.empty((max(E, 1), M), dtype=th.float32, device=logits.device)
        sel = th.empty((N, M2), dtype=th.int32, device=logits.device)
        with KERNEL_TIMER.span("disc_comm_fwd"):
            rc = L.lib().uavgnn_disc_comm_fwd(logits.data_ptr(), M2, L.ptr(gumbel), L.ptr(rng), M, L.ptr(talk_off),
                                              L.ptr(talk_src), N, float(inv_tau), c.data_ptr(), M2, y0.data_ptr(),
                                              sel.data_ptr(), L.stream())
        L.check(rc, "uavgnn_disc_comm_fwd")
        ctx.inv_tau, ctx.M = float(inv_tau), M
        ctx.save_for_backward(y0, sel, t_off, t_dst, t_pos)
        return c

    @staticmethod
    def backward(ctx, d_c):
        y0, sel, t_off, t_dst, t_pos = ctx.saved_tensors
        d_c = L.f32c(d_c)
        N = d_c.shape[0]
        d_logits = th.empty_like(d_c)
        with KERNEL_TIMER.span("disc_comm_bwd"):
            rc = L.lib().uavgnn_disc_comm_bwd(d_c.data_ptr(), d_c.stride(0), y0.data_ptr(), sel.data_ptr(), ctx.M,
                                              L.ptr(t_off), L.ptr(t_dst), L.ptr(t_pos), N, ctx.inv_tau,
                                              d_logits.data_ptr(), d_logits.stride(0), L.stream())
        L.check(rc, "uavgnn_disc_comm_bwd")
        return d_logits, None, None, None, None, None, None, None, None


def disc_comm_aggregate(logits, gumbel, g, tau=0.5, rng=None):
    """Hard Gumbel-softmax messages + OR aggregation of DiscreteComm (gnn_agents.py:166-178).  logits [N, 2*msg] per
    source node; gumbel [E, msg, 2] in CSC order, or None with rng = device int64 {seed, step}: the noise is drawn inside
    the kernel (counter-based Philox keyed by the seed, counter = (CSC position, channel, step))."""
    off, src = g.talk_csc()
    t_off, t_dst, t_pos = _talk_transpose_if_needed(g, logits)
    return _DiscComm.apply(logits, gumbel, off, src, t_off, t_dst, t_pos, 1.0 / tau, rng)


def gumbel_noise(rng, E, msg):
    """The [E, msg, 2] noise tensor the in-kernel generator of K5 draws for rng = {seed, step} (tests)."""
    out = th.empty((E, msg, 2), dtype=th.float32, device=rng.device)
    L.check(L.lib().uavgnn_gumbel_noise(rng.data_ptr(), E, msg, out.data_ptr(), L.stream()), "uavgnn_gumbel_noise")
    return out

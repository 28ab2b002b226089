"""``HeteroBatch``: the observation graph of the MADRQN agent as flat HBM-resident arrays.

This is the input contract of the hot path.  It mirrors the part of the DGL graph API the reference touches
 so that code written against the reference keeps working:

    reference (DGL)                                     here
    ------------------------------------------------    ------------------------------------------------
    dgl.heterograph(data_dict, num_nodes_dict)          heterograph(data_dict, num_nodes_dict)
    g.ndata['feat'] = {...} / g.ndata['feat']           same (dict per node type)
    g.nodes['agent'].data['feat'] (= ...)               same
    dgl.batch([...]) (algos/common.py:45)               batch([...])
    dgl.merge([local_obs, comm_graph]) (env_wr.:137)    merge([...])
    g.num_nodes('agent'), g['talk'], g.to(device)       same

Layout (derived from the invariants of algos/madrqn/utils/env_wrappers.py:69-89,:139-154):
every `gt`/`ubs` source has out-degree 1 and the edges are grouped by destination, so `seen`/`near` are ragged
segments: ``x_gt [E_seen, 4]`` + ``seen_off [N_a + 1]`` (int32), ``x_ubs [E_near, 2]`` + ``near_off``.  Only `talk`
is a real graph; it is kept in CSC (``talk_off``, ``talk_src``: in-edges per destination) together with its transpose
(``t_off``, ``t_dst``, ``t_pos``) for the gather-only backward.  ``talk_eid`` maps a CSC position back to the edge id
the reference would have used (needed only to inject per-edge Gumbel noise reproducibly).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch as th

SEEN = ("gt", "seen", "agent")
NEAR = ("ubs", "near", "agent")
TALK = ("agent", "talk", "agent")
SEEN_BY = ("gt", "seen-by", "agent")  # DRQN twin (algos/drqn/utils/env_wrappers.py:66)


def _i32(x, device=None) -> th.Tensor:
    t = x if isinstance(x, th.Tensor) else th.as_tensor(np.asarray(x))
    t = t.to(th.int32)
    return t.to(device) if device is not None else t


def _offsets_from_dst(dst: th.Tensor, n: int) -> th.Tensor:
    deg = th.bincount(dst.long(), minlength=n)
    off = th.zeros(n + 1, dtype=th.int32, device=dst.device)
    off[1:] = th.cumsum(deg, 0).to(th.int32)
    return off


def seg_ids(off: th.Tensor) -> th.Tensor:
    n = off.numel() - 1
    deg = (off[1:] - off[:-1]).long()
    return th.repeat_interleave(th.arange(n, device=off.device), deg)


# A/B switch: UAVGNN_ORDER_DENSE=1 builds the degree order for dense relations too (relation_order below)
_ORDER_DENSE_SKIP = os.environ.get("UAVGNN_ORDER_DENSE", "0") != "1"


class _Relation:
    """One relation in destination-grouped (CSC) form.  ``src`` is None when src id == edge id."""

    __slots__ = ("off", "src", "eid")

    def __init__(self, off: th.Tensor, src: Optional[th.Tensor] = None, eid: Optional[th.Tensor] = None):
        self.off, self.src, self.eid = off, src, eid

    @property
    def num_edges(self) -> int:
        return int(self.off[-1]) if self.off.numel() else 0

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device, non_blocking=True)  # noqa: E731
        return _Relation(mv(self.off), mv(self.src), mv(self.eid))


class _NodeFrames:
    """``g.nodes[ntype].data`` accessor."""

    def __init__(self, g):
        self._g = g

    def __getitem__(self, ntype):
        g = self._g

        class _V:
            data = g._feat.setdefault(ntype, {})
        return _V


class _NData:
    """``g.ndata[key]`` <-> {ntype: tensor}."""

    def __init__(self, g):
        self._g = g

    def __getitem__(self, key):
        return {nt: fr[key] for nt, fr in self._g._feat.items() if key in fr}

    def __setitem__(self, key, val: Dict[str, th.Tensor]):
        for nt, t in val.items():
            t = th.as_tensor(t)
            if t.shape[0] != self._g._num_nodes.get(nt, 0):
                raise ValueError(f"feature rows ({t.shape[0]}) != number of '{nt}' nodes ({self._g._num_nodes.get(nt)})")
            self._g._feat.setdefault(nt, {})[key] = t
        self._g._cache.clear()


class RelationView:
    """``g['talk']`` - a relation slice sharing the parent's node data (what the comm blocks receive)."""

    def __init__(self, parent: "HeteroBatch", etype: str):
        self.parent, self.etype = parent, etype

    def number_of_edges(self) -> int:
        return self.parent.number_of_edges(self.etype)

    num_edges = number_of_edges


class HeteroBatch:
    def __init__(self, num_nodes: Dict[str, int], rels: Dict[tuple, _Relation],
                 feat: Optional[Dict[str, Dict[str, th.Tensor]]] = None, graph_off: Optional[th.Tensor] = None,
                 hints: Optional[Dict[str, int]] = None):
        self._num_nodes = dict(num_nodes)
        self._rels = dict(rels)
        self._feat = feat if feat is not None else {}
        self.graph_off = graph_off          # [B+1] agent-node offsets of the batched env graphs (int32) or None;
        #                                     talk edges never cross these boundaries
        # Host-side facts the builders know for free and the device arrays would only yield through a sync:
        #   "max_graph_agents"       - largest number of agents of one batched graph
        #   "max_deg:seen" / ":near" - upper bound of the in-degree (the padded width M / U of the observation)
        self.hints: Dict[str, int] = dict(hints or {})
        self._cache: Dict[str, object] = {}

    # ---- constructors -------------------------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, x_a, x_gt=None, seen_off=None, x_ubs=None, near_off=None, talk_off=None, talk_src=None,
                    talk_eid=None, graph_off=None, x_flat=None, device=None, hints=None) -> "HeteroBatch":
        """Direct construction from segment-layout arrays (tests, benchmarks, device-side producers).  graph_off asserts
        that talk edges stay inside the graphs it delimits; hints: see ``HeteroBatch.hints``."""
        f = lambda t: None if t is None else th.as_tensor(t, dtype=th.float32).to(device)  # noqa: E731
        x_a = f(x_a)
        n = x_a.shape[0]
        num_nodes, rels, feat = {"agent": n}, {}, {"agent": {"feat": x_a if x_flat is None else f(x_flat)}}
        if seen_off is not None:
            so = _i32(seen_off, device)
            rels[SEEN] = _Relation(so)
            feat["gt"] = {"feat": f(x_gt)}
            num_nodes["gt"] = feat["gt"]["feat"].shape[0]
        if near_off is not None:
            no = _i32(near_off, device)
            rels[NEAR] = _Relation(no)
            feat["ubs"] = {"feat": f(x_ubs)}
            num_nodes["ubs"] = feat["ubs"]["feat"].shape[0]
        if talk_off is not None:
            rels[TALK] = _Relation(_i32(talk_off, device), _i32(talk_src, device),
                                   None if talk_eid is None else _i32(talk_eid, device))
        go = None if graph_off is None else _i32(graph_off, device)
        hints = dict(hints or {})
        if graph_off is not None and "max_graph_agents" not in hints and not (
                isinstance(graph_off, th.Tensor) and graph_off.is_cuda):
            d = np.diff(np.asarray(graph_off, dtype=np.int64))
            hints["max_graph_agents"] = int(d.max()) if d.size else 0
        return cls(num_nodes, rels, feat, go, hints)

    # ---- DGL-like surface ---------------------------------------------------------------------------------------
    @property
    def ntypes(self):
        return list(self._num_nodes)

    @property
    def canonical_etypes(self):
        return list(self._rels)

    def num_nodes(self, ntype: Optional[str] = None) -> int:
        return sum(self._num_nodes.values()) if ntype is None else self._num_nodes[ntype]

    number_of_nodes = num_nodes

    def _canon(self, etype):
        if isinstance(etype, tuple):
            return etype
        for c in self._rels:
            if c[1] == etype:
                return c
        raise KeyError(etype)

    def number_of_edges(self, etype=None) -> int:
        if etype is None:
            return sum(r.num_edges for r in self._rels.values())
        return self._rels[self._canon(etype)].num_edges

    num_edges = number_of_edges

    def has_relation(self, etype: str) -> bool:
        return any(c[1] == etype for c in self._rels)

    def __getitem__(self, etype) -> RelationView:
        self._canon(etype)
        return RelationView(self, etype if isinstance(etype, str) else etype[1])

    @property
    def ndata(self) -> _NData:
        return _NData(self)

    @property
    def nodes(self) -> _NodeFrames:
        return _NodeFrames(self)

    @property
    def device(self):
        return self._feat["agent"]["feat"].device

    def to(self, device, non_blocking: bool = True) -> "HeteroBatch":
        device = th.device(device)
        cur = self.device
        if cur.type == device.type and (device.index is None or device.index == cur.index):
            return self       # 'cuda' names the current 'cuda:i': no copy, derived indexes stay cached
        feat = {nt: {k: v.to(device, non_blocking=non_blocking) for k, v in fr.items()}
                for nt, fr in self._feat.items()}
        rels = {c: r.to(device) for c, r in self._rels.items()}
        go = None if self.graph_off is None else self.graph_off.to(device)
        return HeteroBatch(self._num_nodes, rels, feat, go, self.hints)

    def pin_memory(self) -> "HeteroBatch":
        for fr in self._feat.values():
            for k in fr:
                fr[k] = fr[k].pin_memory()
        return self

    # ---- kernel-facing accessors --------------------------------------------------------------------------------
    def agent_feat(self) -> th.Tensor:
        return self._feat["agent"]["feat"]

    def relation_segments(self, etype: str):
        """(x_src in segment order [E,F] float32, seg_off [N+1] int32) of `seen` / `near` / `seen-by`."""
        key = "seg:" + etype
        if key not in self._cache:
            c = self._canon(etype)
            r = self._rels[c]
            x = self._feat.get(c[0], {}).get("feat")
            if x is None:
                raise KeyError(f"no 'feat' on node type {c[0]}")
            if r.src is not None:                      # general graphs: gather once into segment order
                x = x.index_select(0, r.src.long())
            # floating features keep their dtype (the HIP ops insist on float32 themselves; float64 containers serve
            # the CPU checkers), anything else becomes float32
            self._cache[key] = ((x if x.is_floating_point() else x.float()).contiguous(), r.off)
        return self._cache[key]

    def _num_edges_host(self, etype: str) -> int:
        """Edge count of a relation WITHOUT a device round trip: the row count of its source-id / feature tensor."""
        r = self._rels[self._canon(etype)]
        if r.src is not None:
            return int(r.src.shape[0])
        x = self._feat.get(self._canon(etype)[0], {}).get("feat")
        return int(x.shape[0]) if x is not None else 0

    def relation_order(self, etype: str) -> th.Tensor:
        """Destinations of a relation sorted by decreasing in-degree (int32 [N]): the hand-out order that balances
        ragged batches over the persistent wavefronts of K1 (scheduling hint only; built once per graph)."""
        key = "ord:" + etype
        if key not in self._cache:
            off = self._rels[self._canon(etype)].off
            n_dst = off.numel() - 1
            if self.hints.get("max_deg:" + etype, 1 << 30) <= 16 or n_dst <= 2048:
                # every destination costs one 16-edge row tile whatever its degree, or there are no more destinations
                # than persistent wavefronts (one each): nothing to balance
                self._cache[key] = None
            elif _ORDER_DENSE_SKIP and not self.hints.get("static") and self._num_edges_host(etype) >= 16 * n_dst:   # static: arrays at capacity, E unknown on the host
                # mean in-degree of 16 or more: (almost) no isolated destinations to move out of the way, and the round-robin
                # hand-out of 16+ destinations per persistent wavefront balances by itself.  Measured at C3 dense (32 768
                # destinations x 80 edges): K1 forward 82.7 us in natural order vs 86.8 us through the order's indirection,
                # plus 19 us for the four launches that build it - per act forward.  Env-realistic batches (94 % isolated,
                # mean degree 1.6) keep the order: 18.4 vs 25.7 us for K1.
                self._cache[key] = None
            elif off.is_cuda:     # HIP counting sort (csrc/build_graph.hip): 3 launches, no host sync
                from . import _lib as L
                n = off.numel() - 1
                order = th.empty(n, dtype=th.int32, device=off.device)
                nbytes = L.lib().uavgnn_degree_order_workspace_bytes(n)
                ws = th.empty(nbytes // 4, dtype=th.int32, device=off.device)
                L.check(L.lib().uavgnn_degree_order(off.data_ptr(), n, L.ptr(order), ws.data_ptr(), nbytes,
                                                    L.stream()), "uavgnn_degree_order")
                self._cache[key] = order
            else:
                deg = (off[1:] - off[:-1])
                self._cache[key] = th.sort(deg, descending=True, stable=True)[1].to(th.int32).contiguous()
        return self._cache[key]

    def talk_csc(self):
        r = self._rels[TALK]
        src = r.src if r.src is not None else th.arange(r.num_edges, dtype=th.int32, device=r.off.device)
        return r.off, src

    def talk_eid(self) -> Optional[th.Tensor]:
        return self._rels[TALK].eid

    def talk_transpose(self):
        """(t_off [N+1], t_dst [E], t_pos [E]): out-edges per source, their destination and CSC position."""
        if "talkT" not in self._cache:
            off, src = self.talk_csc()
            n = self._num_nodes["agent"]
            if off.is_cuda:     # HIP transpose (csrc/build_graph.hip): no sort, no host sync
                from . import _lib as L
                E = src.shape[0]
                t_off = th.empty(n + 1, dtype=th.int32, device=off.device)
                t_dst = th.empty(E, dtype=th.int32, device=off.device)
                t_pos = th.empty(E, dtype=th.int32, device=off.device)
                go = self.graph_off
                if go is not None and go.is_cuda and self.hints.get("max_graph_agents", 1 << 30) <= 256:
                    # batch of small graphs: one wavefront per graph, one launch
                    L.check(L.lib().uavgnn_csc_transpose_env(off.data_ptr(), L.ptr(src), go.data_ptr(),
                                                             go.numel() - 1, n, t_off.data_ptr(), L.ptr(t_dst),
                                                             L.ptr(t_pos), L.stream()), "uavgnn_csc_transpose_env")
                else:
                    nbytes = L.lib().uavgnn_csc_transpose_workspace_bytes(n)
                    ws = th.empty(nbytes // 4, dtype=th.int32, device=off.device)
                    L.check(L.lib().uavgnn_csc_transpose(off.data_ptr(), L.ptr(src), n, E, t_off.data_ptr(),
                                                         L.ptr(t_dst), L.ptr(t_pos), ws.data_ptr(), nbytes,
                                                         L.stream()), "uavgnn_csc_transpose")
                self._cache["talkT"] = (t_off, t_dst, t_pos)
            else:
                dst = seg_ids(off)
                order = th.sort(src.long(), stable=True)[1]
                self._cache["talkT"] = (_offsets_from_dst(src, n), dst[order].to(th.int32).contiguous(),
                                        order.to(th.int32).contiguous())
        return self._cache["talkT"]

    def fresh(self) -> "HeteroBatch":
        """The same graph (shared storage) without any derived index - how a newly built batch looks to the path."""
        return HeteroBatch(self._num_nodes, self._rels, {nt: dict(fr) for nt, fr in self._feat.items()},
                           self.graph_off, self.hints)

    def slice_agents(self, lo: int, hi: int) -> "HeteroBatch":
        """Observation part (agent features + `seen`/`near`) of agents [lo, hi) as a new HeteroBatch whose feature
        arrays are VIEWS of this one (segments are contiguous) and whose offsets are rebased.  Used to address a span
        of time steps inside a time-batched graph; the talk relation is per time step and is not carried over."""
        feat = {"agent": {"feat": self._feat["agent"]["feat"][lo:hi]}}
        rels, num_nodes = {}, {"agent": hi - lo}
        for c, r in self._rels.items():
            if c == TALK:
                continue
            if r.src is not None:
                raise NotImplementedError("slice_agents needs the segment layout (src id == edge id)")
            off = r.off[lo:hi + 1]
            e0, e1 = int(off[0]), int(off[-1])
            rels[c] = _Relation((off - off[0]).contiguous())
            feat[c[0]] = {"feat": self._feat[c[0]]["feat"][e0:e1]}
            num_nodes[c[0]] = e1 - e0
        return HeteroBatch(num_nodes, rels, feat, hints={k: v for k, v in self.hints.items() if k.startswith("max_deg:")})

    def __repr__(self):
        e = {c[1]: r.num_edges for c, r in self._rels.items()}
        return f"HeteroBatch(num_nodes={self._num_nodes}, num_edges={e}, device={self.device})"


DGLGraph = HeteroBatch  # so that `isinstance(x, DGLGraph)` style dispatch (algos/common.py:44) has a target


# ---------------------------------------------------------------------------------------------------------------------
def heterograph(data_dict, num_nodes_dict=None) -> HeteroBatch:
    """Counterpart of ``dgl.heterograph`` for the three relations of the reference (env_wrappers.py:75-81,:146-153)."""
    num_nodes: Dict[str, int] = {}
    coo = {}
    for c, (u, v) in data_dict.items():
        u, v = th.as_tensor(np.asarray(u), dtype=th.int64).reshape(-1), th.as_tensor(np.asarray(v), dtype=th.int64).reshape(-1)
        coo[c] = (u, v)
        for nt, ids in ((c[0], u), (c[2], v)):
            num_nodes[nt] = max(num_nodes.get(nt, 0), int(ids.max()) + 1 if ids.numel() else 0)
    if num_nodes_dict is not None:
        for nt, k in num_nodes_dict.items():
            num_nodes[nt] = int(k)
    rels = {}
    for c, (u, v) in coo.items():
        n_dst = num_nodes[c[2]]
        order = th.sort(v, stable=True)[1]
        su = u[order]
        off = _offsets_from_dst(v, n_dst)
        ident = bool(th.equal(su, th.arange(su.numel())))
        if c == TALK:
            rels[c] = _Relation(off, su.to(th.int32), order.to(th.int32))
        else:
            rels[c] = _Relation(off, None if ident else su.to(th.int32))
    return HeteroBatch(num_nodes, rels, {})


def _union_hints(graphs: Sequence[HeteroBatch]) -> Dict[str, int]:
    """Hints of a disjoint union: a bound holds for the union when every part states it."""
    out: Dict[str, int] = {}
    per_graph = [g.hints.get("max_graph_agents") if g.graph_off is not None else g._num_nodes.get("agent", 0)
                 for g in graphs]
    if all(v is not None for v in per_graph):
        out["max_graph_agents"] = max(per_graph)
    for key in ("max_deg:seen", "max_deg:near"):
        vals = [g.hints.get(key) for g in graphs]
        if all(v is not None for v in vals):
            out[key] = max(vals)
    if any(g.hints.get("static") for g in graphs):   # capacity-sized edge arrays: row counts are not edge counts (relation_order)
        out["static"] = 1
    return out


def batch(graphs: Sequence[HeteroBatch]) -> HeteroBatch:
    """Counterpart of ``dgl.batch`` (algos/common.py:45, env_wrappers.py:67): disjoint union.  Pure concatenation plus
    a running offset, because every array is already grouped by destination."""
    g0 = graphs[0]
    ntypes = list(g0._num_nodes)
    num_nodes = {nt: sum(g._num_nodes.get(nt, 0) for g in graphs) for nt in ntypes}
    feat: Dict[str, Dict[str, th.Tensor]] = {}
    for nt in ntypes:
        keys = set()
        for g in graphs:
            keys |= set(g._feat.get(nt, {}).keys())
        feat[nt] = {k: th.cat([g._feat[nt][k] for g in graphs if k in g._feat.get(nt, {})], 0) for k in keys}
    rels = {}
    for c in g0._rels:
        offs, srcs, eids = [], [], []
        e_base = 0
        s_base = 0
        need_src = any(g._rels[c].src is not None for g in graphs)
        has_eid = all(g._rels[c].eid is not None for g in graphs)
        for i, g in enumerate(graphs):
            r = g._rels[c]
            offs.append((r.off if i == 0 else r.off[1:]) + e_base)
            if need_src:
                s = r.src if r.src is not None else th.arange(r.num_edges, dtype=th.int32, device=r.off.device)
                srcs.append(s + s_base)
            if has_eid:
                eids.append(r.eid + e_base)
            e_base += r.num_edges
            s_base += g._num_nodes.get(c[0], 0)
        rels[c] = _Relation(th.cat(offs).to(th.int32), th.cat(srcs).to(th.int32) if need_src else None,
                            th.cat(eids).to(th.int32) if has_eid and eids else None)
    parts, base = [th.zeros(1, dtype=th.int32, device=g0.device)], 0
    for g in graphs:                       # env boundaries of the union, without leaving the device
        n_ag = g._num_nodes.get("agent", 0)
        if g.graph_off is not None:
            parts.append(g.graph_off[1:].to(g0.device) + base)
        else:
            parts.append(th.full((1,), base + n_ag, dtype=th.int32, device=g0.device))
        base += n_ag
    go = th.cat(parts).to(th.int32)
    return HeteroBatch(num_nodes, rels, feat, go.to(g0.device), _union_hints(graphs))


def merge(graphs: Sequence[HeteroBatch]) -> HeteroBatch:
    """Counterpart of ``dgl.merge([local_obs, comm_graph])`` (env_wrappers.py:137): union of the edge sets over a
    shared node set.  Each relation is taken from the graph that carries its edges (the reference never merges two
    non-empty copies of one relation)."""
    ntypes = list(graphs[0]._num_nodes)
    num_nodes = {nt: max(g._num_nodes.get(nt, 0) for g in graphs) for nt in ntypes}
    rels = {}
    def has_edges(g, c):
        r = g._rels[c]
        if r.off.is_cuda:   # no device->host sync for a structural question: the edge ARRAYS say it
            return (r.src.numel() if r.src is not None else g._num_nodes.get(c[0], 0)) > 0
        return r.num_edges > 0

    for c in graphs[0]._rels:
        holders = [g for g in graphs if c in g._rels and has_edges(g, c)]
        if len(holders) > 1:
            # the cheap test over-counts on the device (a capacity-sized relation without a real edge, a relation without a
            # source array): only this ambiguous case pays for the exact question - one host read of off[-1] per candidate
            holders = [g for g in holders if g._rels[c].num_edges > 0]
        if len(holders) > 1:
            raise NotImplementedError("merge of two non-empty copies of one relation")
        if holders:
            rels[c] = holders[0]._rels[c]
        else:
            n_dst = num_nodes[c[2]]
            rels[c] = _Relation(th.zeros(n_dst + 1, dtype=th.int32),
                                th.zeros(0, dtype=th.int32) if c == TALK else None,
                                th.zeros(0, dtype=th.int32) if c == TALK else None)
    feat: Dict[str, Dict[str, th.Tensor]] = {nt: {} for nt in ntypes}
    for g in graphs:
        for nt in ntypes:
            if g._num_nodes.get(nt, 0) == num_nodes[nt]:
                for k, v in g._feat.get(nt, {}).items():
                    feat[nt].setdefault(k, v)
    # graph_off delimits the TALK relation (talk edges never cross it), so it - and the max_graph_agents hint - may only
    # be inherited from the graph that carries the talk edges.  In the reference's flow (env_wrappers.py:137) the
    # other operand is dgl.batch of n ONE-agent observation graphs (graph_off = [0,1,..,n]) whose boundaries the
    # communication edges cross: the merged graph is ONE graph of n agents.
    n_ag = num_nodes.get("agent", 0)
    dev = next((fr["feat"].device for nt, fr in feat.items() if nt == "agent" and "feat" in fr),
               graphs[0]._rels[next(iter(graphs[0]._rels))].off.device if graphs[0]._rels else None)
    talk_holder = next((g for g in graphs if TALK in g._rels and has_edges(g, TALK)), None)
    hints: Dict[str, int] = {}
    for g in graphs:
        for k, v in g.hints.items():
            if k != "max_graph_agents":
                hints[k] = max(v, hints.get(k, v))
    if talk_holder is not None and talk_holder.graph_off is not None:
        go = talk_holder.graph_off
        if "max_graph_agents" in talk_holder.hints:
            hints["max_graph_agents"] = talk_holder.hints["max_graph_agents"]
    else:
        go = th.tensor([0, n_ag], dtype=th.int32, device=dev)
        hints["max_graph_agents"] = n_ag
    return HeteroBatch(num_nodes, rels, feat, go, hints)


def cat(data_list: List):
    """Counterpart of ``algos.common.cat`` (common.py:40-47)."""
    if isinstance(data_list[0], th.Tensor):
        return th.cat(data_list)
    if isinstance(data_list[0], HeteroBatch):
        return batch(data_list)
    raise TypeError("Unrecognised observation type.")


# ---------------------------------------------------------------------------------------------------------------------
def from_obs_dicts(obs: Sequence[dict], d_u2u=None, r_comm: float = np.inf, with_comm: bool = True) -> HeteroBatch:
    """One env step -> HeteroBatch, vectorised counterpart of ``GraphObservation.local_observation`` +
    ``MultiUbsCoverageWrapper.build_comm_graph`` + ``dgl.merge`` (env_wrappers.py:65-89,:122-154).

    obs: per-agent dicts {agent [2], ubs [n-1,3], gt [M,5]} with column 0 = visibility flag (mubs_cov.py:215-242).
    talk edges i->j exist iff d_u2u[i,j] <= r_comm, self loops included (env_wrappers.py:140-144)."""
    n = len(obs)
    gt = np.stack([np.asarray(o["gt"], dtype=np.float32) for o in obs])       # [n, M, 5]
    ub = np.stack([np.asarray(o["ubs"], dtype=np.float32) for o in obs])      # [n, n-1, 3]
    xa = np.stack([np.asarray(o["agent"], dtype=np.float32) for o in obs])
    mg, mu = gt[:, :, 0] == 1, ub[:, :, 0] == 1
    seen_off = np.concatenate([[0], np.cumsum(mg.sum(1))]).astype(np.int32)
    near_off = np.concatenate([[0], np.cumsum(mu.sum(1))]).astype(np.int32)
    kw = dict(x_a=xa, x_gt=gt[mg][:, 1:], seen_off=seen_off, x_ubs=ub[mu][:, 1:], near_off=near_off,
              graph_off=[0, n], hints={"max_deg:seen": gt.shape[1], "max_deg:near": ub.shape[1]})
    if with_comm:
        adj = np.asarray(d_u2u) <= r_comm                                    # adj[i, j]: edge i -> j
        eid_of = np.cumsum(adj.reshape(-1)).reshape(n, n) - 1                 # reference edge id (i-major order)
        src, dst = np.nonzero(adj.T)[1], np.nonzero(adj.T)[0]                 # grouped by destination j
        kw.update(talk_off=np.concatenate([[0], np.cumsum(adj.sum(0))]).astype(np.int32),
                  talk_src=src.astype(np.int32), talk_eid=eid_of[src, dst].astype(np.int32))
    return HeteroBatch.from_arrays(**kw)


_SMALL_MAX = None


def _small_max_agents() -> int:
    global _SMALL_MAX
    if _SMALL_MAX is None:
        from . import _lib as L
        _SMALL_MAX = int(L.lib().uavgnn_build_graph_small_max_agents())
    return _SMALL_MAX


_GRAPH_OFF_CACHE = {}
_SCAN4_MAX_ROWS = 1 << 17   # one-launch scans (uavgnn_offsets_scan4) up to here: 16 rounds of 8192 per array


def _uniform_graph_off(N, n, dev):
    """[0, n, 2n, ..., N] for a batch of equal-size graphs; read-only, shared between batches of the same shape."""
    if th.cuda.is_current_stream_capturing():     # a tensor born inside a capture belongs to the graph's pool: never cached
        return th.arange(0, N + 1, n, dtype=th.int32, device=dev)
    key = (N, n, str(dev))
    t = _GRAPH_OFF_CACHE.get(key)
    if t is None:
        if len(_GRAPH_OFF_CACHE) > 64:
            _GRAPH_OFF_CACHE.clear()
        t = _GRAPH_OFF_CACHE[key] = th.arange(0, N + 1, n, dtype=th.int32, device=dev)
    return t


def from_padded_obs(gt: th.Tensor, ubs: th.Tensor, agent: th.Tensor, d_u2u: Optional[th.Tensor] = None,
                    r_comm: float = float("inf"), static: bool = False) -> HeteroBatch:
    """Device-side builder for B environments at once: padded observation tensors
    gt [B,n,M,5], ubs [B,n,n-1,3] (column 0 = visibility flag), agent [B,n,2] and d_u2u [B,n,n], all resident on the GPU,
    become a HeteroBatch through two HIP passes (count, compact) and prefix sums; nothing is copied to the host except
    the three edge totals needed to size the outputs.  Bit-identical to batching ``from_obs_dicts`` per environment.

    static=True: no host round trip at all - the edge arrays are allocated at CAPACITY (B n M / B n (n-1) / B n n rows)
    and only their first E rows are written; offsets say which.  Shapes then depend on (B, n, M) only, which is what a
    hipGraph capture of the step needs (uav_bs_ctrl_amd/graphs.py); ``number_of_edges`` still reports the true count."""
    from . import _lib as L
    L.require_gpu(gt, ubs, agent, d_u2u)
    B, n, M, Sg = gt.shape
    U, Su = ubs.shape[2], ubs.shape[3]
    N = B * n
    dev = gt.device
    gt, ubs, agent = L.f32c(gt), L.f32c(ubs), L.f32c(agent)
    i32 = dict(dtype=th.int32, device=dev)
    with_comm = d_u2u is not None
    if static and 0 < N <= _small_max_agents() and n <= 64 and Sg == 5 and Su == 3:
        # small batch without a host round trip: counts, prefix sums and compaction in ONE launch
        f32 = dict(dtype=th.float32, device=dev)
        seen_off, near_off = th.empty(N + 1, **i32), th.empty(N + 1, **i32)
        talk_off = th.empty(N + 1, **i32) if with_comm else None
        x_gt, x_ubs = th.empty((N * M, 4), **f32), th.empty((N * U, 2), **f32)
        talk_src = th.empty(N * n, **i32) if with_comm else None
        talk_eid = th.empty(N * n, **i32) if with_comm else None
        graph_off = th.empty(B + 1, **i32)
        if with_comm:
            d_u2u = L.f32c(d_u2u)
        L.check(L.lib().uavgnn_build_graph_small(gt.data_ptr(), M, 4, ubs.data_ptr(), U, 2, L.ptr(d_u2u), n, B,
                                                 float(min(r_comm, 3.0e38)), seen_off.data_ptr(), near_off.data_ptr(),
                                                 L.ptr(talk_off), x_gt.data_ptr(), x_ubs.data_ptr(), L.ptr(talk_src),
                                                 L.ptr(talk_eid), graph_off.data_ptr(), L.stream()), "uavgnn_build_graph_small")
        kw = dict(x_a=agent.view(N, -1), x_gt=x_gt, seen_off=seen_off, x_ubs=x_ubs, near_off=near_off, graph_off=graph_off,
                  hints={"max_graph_agents": n, "max_deg:seen": M, "max_deg:near": U})
        if with_comm:
            kw.update(talk_off=talk_off, talk_src=talk_src, talk_eid=talk_eid)
        return HeteroBatch.from_arrays(device=dev, **kw)
    deg_s, deg_n = th.empty(N, **i32), th.empty(N, **i32)
    L.check(L.lib().uavgnn_obs_degrees(gt.data_ptr(), M, Sg - 1, ubs.data_ptr(), U, Su - 1, N, deg_s.data_ptr(),
                                       deg_n.data_ptr(), L.stream()), "uavgnn_obs_degrees")
    seen_off, near_off = th.empty(N + 1, **i32), th.empty(N + 1, **i32)
    deg_t = env_e = talk_off = env_base = None
    if with_comm:
        d_u2u = L.f32c(d_u2u)
        deg_t, env_e = th.empty(N, **i32), th.empty(B, **i32)
        L.check(L.lib().uavgnn_talk_degrees(d_u2u.data_ptr(), n, B, float(min(r_comm, 3.0e38)), deg_t.data_ptr(),
                                            env_e.data_ptr(), L.stream()), "uavgnn_talk_degrees")
        talk_off, env_base = th.empty(N + 1, **i32), th.empty(B + 1, **i32)
    if N <= _SCAN4_MAX_ROWS:
        # all exclusive scans in one launch (seen / near / talk offsets over the agents, edge bases over the environments)
        L.check(L.lib().uavgnn_offsets_scan4(deg_s.data_ptr(), N, seen_off.data_ptr(), deg_n.data_ptr(), N, near_off.data_ptr(),
                                             L.ptr(deg_t), N if with_comm else 0, L.ptr(talk_off), L.ptr(env_e),
                                             B if with_comm else 0, L.ptr(env_base), L.stream()), "uavgnn_offsets_scan4")
    else:   # the time-batched builds ((T + 1) B n rows): one workgroup per array would walk them serially - multi-block scans
        for off, deg in ((seen_off, deg_s), (near_off, deg_n), (talk_off, deg_t), (env_base, env_e)):
            if off is not None:
                off[0] = 0
                th.cumsum(deg, 0, out=off[1:])
    if with_comm:
        if not static:
            totals = th.stack((seen_off[-1], near_off[-1], talk_off[-1])).tolist()   # the one host sync: output sizes
    elif not static:
        totals = th.stack((seen_off[-1], near_off[-1])).tolist() + [0]
    if static:
        totals = [N * M, N * U, N * n if with_comm else 0]
    Es, En, Et = totals
    # static: arrays at CAPACITY, only the first E rows are written by the compaction - the rest is zero-filled (a memset is
    # capturable) so that no op over all rows can ever pick up uninitialised pool memory
    alloc = th.zeros if static else th.empty
    x_gt = alloc((Es, Sg - 1), dtype=th.float32, device=dev)
    x_ubs = alloc((En, Su - 1), dtype=th.float32, device=dev)
    L.check(L.lib().uavgnn_obs_compact(gt.data_ptr(), M, Sg - 1, ubs.data_ptr(), U, Su - 1, N, seen_off.data_ptr(),
                                       near_off.data_ptr(), x_gt.data_ptr(), x_ubs.data_ptr(), L.stream()),
            "uavgnn_obs_compact")
    kw = dict(x_a=agent.view(N, -1), x_gt=x_gt, seen_off=seen_off, x_ubs=x_ubs, near_off=near_off,
              graph_off=_uniform_graph_off(N, n, dev),
              hints={"max_graph_agents": n, "max_deg:seen": M, "max_deg:near": U, **({"static": 1} if static else {})})
    if with_comm:
        talk_src, talk_eid = alloc(Et, **i32), alloc(Et, **i32)
        L.check(L.lib().uavgnn_talk_compact(d_u2u.data_ptr(), n, B, float(min(r_comm, 3.0e38)), talk_off.data_ptr(),
                                            env_base.data_ptr(), talk_src.data_ptr(), talk_eid.data_ptr(), L.stream()),
                "uavgnn_talk_compact")
        kw.update(talk_off=talk_off, talk_src=talk_src, talk_eid=talk_eid)
    return HeteroBatch.from_arrays(device=dev, **kw)

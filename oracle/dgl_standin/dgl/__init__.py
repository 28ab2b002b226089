"""Pure-torch stand-in for the handful of DGL 0.9 symbols the reference touches.

TEST INFRASTRUCTURE ONLY, used in the build container by ``tests/golden/make_golden.py`` so that the reference's
``algos/madrqn/agents/gnn_agents.py``, ``algos/common.py`` and ``algos/madrqn/utils/env_wrappers.py`` import
*unchanged* from /root/reference.  DGL itself (dgl==0.9.0, requirements.txt:17) is not installed and cannot be
fetched; its semantics are restated here from its public documentation.  This is NOT DGL and pins nothing at the
DGL boundary ("parity unpinned"); it exists so that the reference's *wiring* (which tensors feed which op, which
are detached, parameter names/order) is exercised by the reference's own code rather than re-typed.
"""
from __future__ import annotations

from contextlib import contextmanager

import torch as th

from . import function  # noqa: F401  (dgl.function)

_ETYPES_DEFAULT = None


class _Frame(dict):
    pass


class _NodeView:
    def __init__(self, g):
        self._g = g

    def __getitem__(self, ntype):
        g = self._g

        class _NS:
            data = g._nframes[ntype]
        return _NS


class _HeteroNData:
    """``g.ndata`` of a graph with several node types: values are {ntype: tensor} dicts."""

    def __init__(self, g):
        self._g = g

    def __getitem__(self, key):
        return {nt: fr[key] for nt, fr in self._g._nframes.items() if key in fr}

    def __setitem__(self, key, val):
        assert isinstance(val, dict)
        for nt, t in val.items():
            assert t.shape[0] == self._g._num_nodes[nt], (nt, t.shape, self._g._num_nodes[nt])
            self._g._nframes[nt][key] = t


class DGLGraph:
    """Heterograph: node types with counts + per-type feature frames; canonical edge types with (src, dst)."""

    def __init__(self, num_nodes, edges, nframes=None):
        self._num_nodes = dict(num_nodes)                      # ntype -> int
        self._edges = dict(edges)                              # (st, et, dt) -> (src long, dst long)
        self._nframes = nframes if nframes is not None else {nt: _Frame() for nt in num_nodes}
        self._eframe = _Frame()                                # only single-relation slices use edata

    # ---- structure ----------------------------------------------------------------------------------------------
    @property
    def ntypes(self):
        return list(self._num_nodes)

    @property
    def canonical_etypes(self):
        return list(self._edges)

    def num_nodes(self, ntype=None):
        if ntype is None:
            return sum(self._num_nodes.values())
        return self._num_nodes[ntype]

    number_of_nodes = num_nodes

    def number_of_edges(self, etype=None):
        if etype is None:
            return sum(int(s.shape[0]) for s, _ in self._edges.values())
        return int(self._edges[self._canon(etype)][0].shape[0])

    num_edges = number_of_edges

    def _canon(self, etype):
        if isinstance(etype, tuple):
            return etype
        (c,) = [c for c in self._edges if c[1] == etype]
        return c

    def edges(self, etype=None):
        if etype is None:
            (c,) = list(self._edges)
        else:
            c = self._canon(etype)
        return self._edges[c]

    def __getitem__(self, etype):
        """Relation slice sharing the node frames of the parent (DGL semantics)."""
        c = self._canon(etype)
        st, _, dt = c
        nn_ = {st: self._num_nodes[st]} if st == dt else {st: self._num_nodes[st], dt: self._num_nodes[dt]}
        g = DGLGraph(nn_, {c: self._edges[c]}, nframes={nt: self._nframes[nt] for nt in nn_})
        return g

    def to(self, device):
        edges = {c: (s.to(device), d.to(device)) for c, (s, d) in self._edges.items()}
        nframes = {nt: _Frame({k: v.to(device) for k, v in fr.items()}) for nt, fr in self._nframes.items()}
        return DGLGraph(self._num_nodes, edges, nframes)

    # ---- frames -------------------------------------------------------------------------------------------------
    def _single(self):
        (c,) = list(self._edges)
        return c

    @property
    def nodes(self):
        return _NodeView(self)

    @property
    def ndata(self):
        if len(self._num_nodes) == 1:
            return self._nframes[next(iter(self._num_nodes))]
        return _HeteroNData(self)

    @property
    def srcdata(self):
        return self._nframes[self._single()[0]]

    @property
    def dstdata(self):
        return self._nframes[self._single()[2]]

    @property
    def edata(self):
        return self._eframe

    @contextmanager
    def local_scope(self):
        saved_n = {nt: dict(fr) for nt, fr in self._nframes.items()}
        saved_e = dict(self._eframe)
        try:
            yield
        finally:
            for nt, fr in self._nframes.items():
                fr.clear()
                fr.update(saved_n[nt])
            self._eframe.clear()
            self._eframe.update(saved_e)

    # ---- message passing ----------------------------------------------------------------------------------------
    def apply_edges(self, func):
        src, dst = self.edges()
        if isinstance(func, function.BuiltinMessage):
            self._eframe[func.out] = func(self.srcdata, self.dstdata, self._eframe, src, dst)
        else:
            self._eframe.update(func(_EdgeBatch(self, src, dst)))

    def update_all(self, message_func, reduce_func):
        src, dst = self.edges()
        n_dst = self._num_nodes[self._single()[2]]
        if isinstance(message_func, function.BuiltinMessage):
            msgs = {message_func.out: message_func(self.srcdata, self.dstdata, self._eframe, src, dst)}
        else:
            msgs = message_func(_EdgeBatch(self, src, dst))
        if isinstance(reduce_func, function.BuiltinReduce):
            self.dstdata[reduce_func.out] = reduce_func(msgs[reduce_func.msg], dst, n_dst)
            return
        # UDF reduce: degree bucketing.  Nodes are grouped by in-degree; each bucket sees a mailbox
        # [nodes, deg, ...] with messages in edge-id order.  Zero in-degree nodes get zeros.
        order = th.sort(dst, stable=True)[1]
        dst_s = dst[order]
        deg = th.bincount(dst, minlength=n_dst)
        start = th.cumsum(deg, 0) - deg
        out = {}
        for d in th.unique(deg).tolist():
            if d == 0:
                continue
            nodes = th.nonzero(deg == d).flatten()
            eidx = (start[nodes].unsqueeze(1) + th.arange(d, device=dst.device).unsqueeze(0))  # [nodes, d]
            mailbox = {k: v[order][eidx] for k, v in msgs.items()}
            res = reduce_func(_NodeBatch(mailbox))
            for k, v in res.items():
                if k not in out:
                    out[k] = th.zeros((n_dst,) + v.shape[1:], dtype=v.dtype, device=v.device)
                out[k] = out[k].index_copy(0, nodes, v)
        del dst_s
        self.dstdata.update(out)


class _EdgeBatch:
    def __init__(self, g, src, dst):
        self.src = {k: v.index_select(0, src) for k, v in g.srcdata.items()}
        self.dst = {k: v.index_select(0, dst) for k, v in g.dstdata.items()}
        self.data = g.edata


class _NodeBatch:
    def __init__(self, mailbox):
        self.mailbox = mailbox


DGLHeteroGraph = DGLGraph


def heterograph(data_dict, num_nodes_dict=None, idtype=None, device=None):
    edges = {}
    for c, (u, v) in data_dict.items():
        u = th.as_tensor(u, dtype=th.long) if not isinstance(u, th.Tensor) else u.long()
        v = th.as_tensor(v, dtype=th.long) if not isinstance(v, th.Tensor) else v.long()
        edges[c] = (u.reshape(-1), v.reshape(-1))
    num_nodes = {}
    for (st, _, dt), (u, v) in edges.items():
        for nt, ids in ((st, u), (dt, v)):
            cnt = int(ids.max()) + 1 if ids.numel() else 0
            num_nodes[nt] = max(num_nodes.get(nt, 0), cnt)
    if num_nodes_dict is not None:
        for nt, cnt in num_nodes_dict.items():
            num_nodes[nt] = int(cnt)
    return DGLGraph(num_nodes, edges)


def _cat_frames(graphs):
    nframes = {}
    for nt in graphs[0]._num_nodes:
        keys = set()
        for g in graphs:
            if g._num_nodes[nt] > 0:
                keys |= set(g._nframes[nt].keys())
        fr = _Frame()
        for k in keys:
            parts = [g._nframes[nt][k] for g in graphs if k in g._nframes[nt]]
            fr[k] = th.cat(parts, 0)
        nframes[nt] = fr
    return nframes


def batch(graphs):
    """Disjoint union: node ids of graph i are offset by the node counts of graphs < i; no cross-graph edges."""
    ntypes = graphs[0].ntypes
    num_nodes = {nt: sum(g._num_nodes[nt] for g in graphs) for nt in ntypes}
    edges = {}
    for c in graphs[0].canonical_etypes:
        st, _, dt = c
        so = do = 0
        us, vs = [], []
        for g in graphs:
            u, v = g._edges[c]
            us.append(u + so)
            vs.append(v + do)
            so += g._num_nodes[st]
            do += g._num_nodes[dt]
        edges[c] = (th.cat(us), th.cat(vs))
    return DGLGraph(num_nodes, edges, _cat_frames(graphs))


def merge(graphs):
    """Union of edges over a shared node set (the node count of each type is the max over inputs); node features
    are taken from the first graph that carries them."""
    ntypes = graphs[0].ntypes
    num_nodes = {nt: max(g._num_nodes[nt] for g in graphs) for nt in ntypes}
    edges = {}
    for c in graphs[0].canonical_etypes:
        edges[c] = (th.cat([g._edges[c][0] for g in graphs]), th.cat([g._edges[c][1] for g in graphs]))
    nframes = {nt: _Frame() for nt in ntypes}
    for g in graphs:
        for nt in ntypes:
            if g._num_nodes[nt] == num_nodes[nt]:
                for k, v in g._nframes[nt].items():
                    nframes[nt].setdefault(k, v)
    return DGLGraph(num_nodes, edges, nframes)

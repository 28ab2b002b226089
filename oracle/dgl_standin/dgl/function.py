"""Stand-in for ``dgl.function`` built-ins used at gnn_agents.py:261,266 and inside GATv2Conv.  Test infrastructure."""
import torch as th


class BuiltinMessage:
    def __init__(self, kind, lhs, rhs, out):
        self.kind, self.lhs, self.rhs, self.out = kind, lhs, rhs, out

    def __call__(self, srcdata, dstdata, edata, src, dst):
        k = self.kind
        if k == "u_dot_v":   # keeps a trailing dim of 1 (DGL semantics)
            return (srcdata[self.lhs].index_select(0, src) * dstdata[self.rhs].index_select(0, dst)).sum(-1, keepdim=True)
        if k == "u_add_v":
            return srcdata[self.lhs].index_select(0, src) + dstdata[self.rhs].index_select(0, dst)
        if k == "u_mul_e":
            return srcdata[self.lhs].index_select(0, src) * edata[self.rhs]
        if k == "copy_u":
            return srcdata[self.lhs].index_select(0, src)
        raise NotImplementedError(k)


class BuiltinReduce:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out

    def __call__(self, m, dst, n):
        if self.kind == "sum":
            return th.zeros((n,) + m.shape[1:], dtype=m.dtype, device=m.device).index_add(0, dst, m)
        raise NotImplementedError(self.kind)


def u_dot_v(lhs, rhs, out):
    return BuiltinMessage("u_dot_v", lhs, rhs, out)


def u_add_v(lhs, rhs, out):
    return BuiltinMessage("u_add_v", lhs, rhs, out)


def u_mul_e(lhs, rhs, out):
    return BuiltinMessage("u_mul_e", lhs, rhs, out)


def copy_u(lhs, out):
    return BuiltinMessage("copy_u", lhs, None, out)


def sum(msg, out):  # noqa: A001
    return BuiltinReduce("sum", msg, out)

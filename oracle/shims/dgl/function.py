"""dgl.function.u_mul_e / sum (built-in message and reduce functions)."""
import torch


def u_mul_e(lhs, rhs, out):
    def _msg(g):
        return {out: g.ndata[lhs][g._src.long()] * g.edata[rhs]}
    return _msg


def sum(msg, out):  # noqa: A001  (DGL's own name)
    def _reduce(g, m):
        v = m[msg]
        acc = torch.zeros((g._n,) + tuple(v.shape[1:]), dtype=v.dtype)
        acc.index_add_(0, g._dst.long(), v)
        g.ndata[out] = acc
    return _reduce

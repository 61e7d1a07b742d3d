"""dgl.ops.edge_softmax, norm_by='dst' (DGL's EdgeSoftmax forward: gspmm max,
gsddmm sub + exp, gspmm sum, gsddmm div)."""
import torch


def edge_softmax(g, logits, eids=None, norm_by='dst'):
    assert norm_by == 'dst'
    d = g._dst.long()
    shape = (g._n,) + tuple(logits.shape[1:])
    idx = d.view(-1, *([1] * (logits.dim() - 1))).expand_as(logits)
    mx = torch.full(shape, float('-inf'), dtype=logits.dtype)
    mx = mx.scatter_reduce(0, idx, logits, reduce='amax', include_self=True)
    score = torch.exp(logits - mx[d])
    ssum = torch.zeros(shape, dtype=logits.dtype)
    ssum.index_add_(0, d, score)
    return score / ssum[d]

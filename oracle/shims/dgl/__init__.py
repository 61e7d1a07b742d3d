"""Minimal torch stand-in for the DGL API surface the reference touches
(gat2.py:61,66,84; graph_generator.py:9-11,866-869; metrics_from_model.py:201-204).

Semantics restated from DGL's documented behaviour:
  * apply_edges(udf): udf sees src/dst node data gathered per edge, returns edge data;
  * update_all(u_mul_e(a,b,m), sum(m,o)): o[v] = sum over incoming edges e=(u->v) of
    a[u]*b[e], accumulated in edge order;
  * ops.edge_softmax(g, logits): softmax over the incoming edges of each destination
    (norm_by='dst'): max per dst, exp(x-max), sum per dst, divide.
"""
import torch
from . import function, ops, data  # noqa: F401


class _Frame(dict):
    pass


class _EdgeBatch:
    def __init__(self, g):
        s, d = g._src.long(), g._dst.long()
        self.src = {k: v[s] for k, v in g.ndata.items()}
        self.dst = {k: v[d] for k, v in g.ndata.items()}
        self.data = g.edata


class DGLGraph:
    def __init__(self, src, dst, num_nodes, idtype):
        self._src = torch.as_tensor(src, dtype=idtype)
        self._dst = torch.as_tensor(dst, dtype=idtype)
        self._n = int(num_nodes)
        self.ndata = _Frame()
        self.edata = _Frame()
        self.idtype = idtype

    def to(self, device):
        return self

    def edges(self):
        return self._src, self._dst

    def nodes(self):
        return torch.arange(self._n, dtype=self.idtype)

    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        return int(self._src.shape[0])

    num_edges = number_of_edges

    def apply_edges(self, func):
        self.edata.update(func(_EdgeBatch(self)))

    def update_all(self, message_func, reduce_func):
        msg = message_func(self)
        reduce_func(self, msg)


def graph(data, num_nodes=None, idtype=torch.int64):
    src, dst = data
    if num_nodes is None:
        num_nodes = int(max(max(src), max(dst))) + 1
    return DGLGraph(src, dst, num_nodes, idtype)


def batch(graphs):
    raise NotImplementedError('dgl.batch is training-only; not part of the oracle')


def save_graphs(path, graphs, labels=None):
    raise NotImplementedError


def load_graphs(path):
    raise NotImplementedError

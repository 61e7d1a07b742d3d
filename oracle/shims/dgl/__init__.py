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
    """dgl.batch: one graph holding the input graphs as disjoint components, nodes and edges relabelled in input order
    (node i of graph k becomes i + the node count of graphs 0..k-1), node / edge data concatenated.  Called by the collate
    functions of train_skeleton_matching.py:67-84 and test/sm_metrics_without_gt.py:46-64."""
    src, dst, base = [], [], 0
    for g in graphs:
        src.append(g._src.long() + base)
        dst.append(g._dst.long() + base)
        base += g._n
    out = DGLGraph(torch.cat(src), torch.cat(dst), base, graphs[0].idtype)
    for k in graphs[0].ndata:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], dim=0)
    for k in graphs[0].edata:
        out.edata[k] = torch.cat([g.edata[k] for g in graphs], dim=0)
    out._batch_num_nodes = [g._n for g in graphs]
    return out


def save_graphs(path, graphs, labels=None):
    """dgl.save_graphs: persist a list of graphs (graph_generator.py:895).  Stand-in format: torch.save of plain tensors."""
    torch.save([{'src': g._src, 'dst': g._dst, 'n': g._n, 'idtype': g.idtype, 'ndata': dict(g.ndata), 'edata': dict(g.edata)}
                for g in graphs], path)


def load_graphs(path):
    out = []
    for d in torch.load(path):
        g = DGLGraph(d['src'], d['dst'], d['n'], d['idtype'])
        g.ndata.update(d['ndata'])
        g.edata.update(d['edata'])
        out.append(g)
    return out, {}

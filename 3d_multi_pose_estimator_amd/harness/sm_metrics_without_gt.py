"""Counterpart of the reference's test/sm_metrics_without_gt.py: clustering quality of the skeleton-matching stage on
scenes COMPOSED from single-person recordings, where the grouping is known by construction, so no 3D ground truth is
needed.

Same as the reference script: the CLI (--testfiles, one file per individual; --modelsdir; --datastep, parsed and unused
there too, :26,74), `probabilities_set` (0.8 for the first file, 0.8 x length ratio for the others, :99-104), the scenes
of MergedMultipleHumansDataset(TEST_FILES, probabilities_set, limit=1000, mode='test_generated', alt='3') (:108 -- the mirror
of graph_generator.py:672-810 in this package: same `random` call sequence, so the same seed gives the reference's scenes),
per scene the proposals from the network's scores (:131) and from the labels used as scores (:143-147), one label per head =
index of the first proposal that holds it, else the number of proposals (:133-141, 149-157), and the four printed means
(adjusted Rand index, homogeneity, completeness, V-measure, :159-170).

Different on purpose: the scenes go through the engine in BATCHES of graphs (one frame of the batch per graph, explicit
edge-node lists: what dgl.batch is to the reference's training loop) instead of one forward call per graph; `--seed`
seeds Python's `random` first (the reference is unseeded: its numbers change run to run); and the dataset is always
rebuilt (force_reload) -- the reference would silently reuse `cache/` of an earlier run with OTHER test files, because
the cache name holds only mode, alternative and limit (graph_generator.py:560-563).
"""
import json
import random

import numpy as np
import torch

from ..calibration import Calibration
from ..graph_generator import MergedMultipleHumansDataset, batch
from ..parameters import parameters
from ..pipeline import Engine, explicit_m_cap
from .common import build_parser, load_models

CLASSIFICATION_THRESHOLD = 0.5


def labels_of(persons_row, n_persons, H):
    """:133-141: index of the first proposal whose values hold head h, else the number of proposals."""
    out = []
    for h in range(H):
        idx = n_persons
        for p in range(n_persons):
            if h in persons_row[p]:
                idx = p
                break
        out.append(idx)
    return out


def evaluate(dataset, eng, batch_graphs=64):
    from sklearn.metrics import adjusted_rand_score, homogeneity_completeness_v_measure
    tot = {'rand score': 0.0, 'homogeneity': 0.0, 'completeness': 0.0, 'v_measure': 0.0}
    n_data = 0
    per_graph = []
    for start in range(0, len(dataset), batch_graphs):
        items = [dataset[i] for i in range(start, min(len(dataset), start + batch_graphs))]
        g = batch([it[0] for it in items])
        db = g.device_batch(eng)
        _, persons, n_persons = eng.match(db, want_scores=False)
        # the true grouping: the same proposal function on the labels used as scores (:143-147)
        lab = torch.cat([it[1].reshape(-1) for it in items]).to(torch.float32)
        gt_persons, gt_n = eng.cluster(db, lab)
        eng.sync_status()
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        gt_persons, gt_n = gt_persons.cpu().numpy(), gt_n.cpu().numpy()
        for f, H in enumerate(g.batch_num_heads):
            est = labels_of([set(int(h) for h in row if h >= 0) for row in persons[f, :n_persons[f]]], int(n_persons[f]), H)
            gt = labels_of([set(int(h) for h in row if h >= 0) for row in gt_persons[f, :gt_n[f]]], int(gt_n[f]), H)
            n_data += 1
            tot['rand score'] += adjusted_rand_score(gt, est)
            hom, com, v = homogeneity_completeness_v_measure(gt, est)
            tot['homogeneity'] += hom
            tot['completeness'] += com
            tot['v_measure'] += v
            per_graph.append({'est': persons[f, :n_persons[f]].tolist(), 'gt': gt_persons[f, :gt_n[f]].tolist()})
    out = {k: v / max(1, n_data) for k, v in tot.items()}
    for k in ('rand score', 'homogeneity', 'completeness', 'v_measure'):
        print(k, out[k])
    out['n_data'] = n_data
    out['per_graph'] = per_graph
    return out


def run(args):
    if getattr(args, 'seed', None) is not None:
        random.seed(args.seed)
    first_length = len(json.loads(open(args.testfiles[0], 'rb').read()))
    probabilities_set = [0.8]
    for filename in args.testfiles[1:]:
        probabilities_set.append(0.8 * len(json.loads(open(filename, 'rb').read())) / first_length)
    print('LOADING GRAPHS...')
    dataset = MergedMultipleHumansDataset(args.testfiles, probabilities_set, limit=1000, mode='test_generated', alt='3', raw_dir='.',
                                          force_reload=True)
    if len(dataset) == 0:
        print('no graph could be composed from the test files')
        return {'n_data': 0}
    hmax = max(g.H for g in dataset.graphs)
    mmax = max(g.M for g in dataset.graphs)
    hpf = max(hmax, 2)
    while explicit_m_cap(hpf) < mmax:          # per-graph edge-node capacity follows the head capacity (include/mpe.h)
        hpf += 1
    B = max(1, min(int(args.batch), len(dataset)))
    eng = Engine(parameters, Calibration(parameters), max_frames=B, max_heads_per_frame=hpf, max_edge_nodes_per_frame=mmax,
                 threshold=CLASSIFICATION_THRESHOLD)
    load_models(eng, args, need_mlp=False)
    out = evaluate(dataset, eng, B)
    eng.close()
    return out


def main(argv=None):
    p = build_parser('Print metrics of the skeleton-matching model (ground truth is not required)')
    p.add_argument('--seed', type=int, default=None, help="seed Python's random before the scenes are sampled (the reference does not)")
    return run(p.parse_args(argv))


if __name__ == '__main__':
    main()

"""API mirror of the reference's utils/skeleton_matching_utils.py: the greedy clustering runs
in the HIP kernel k_cluster (csrc/cluster.hip) through mpe_cluster_batch."""
import torch

from . import runtime
from .parameters import parameters


def get_person_proposal_from_network_output(outputs, subgraph, indices, nodes_camera, jsons_for_head=None,
                                            CLASSIFICATION_THRESHOLD=0.5):
    """Same arguments and result as the reference (:12-132): a list of
    ``{camera_name: head id or None}`` over ``used_cameras_skeleton_matching``.  `outputs` may
    be a tensor or a Python list of per-node scores; only edge-node entries are read."""
    if getattr(subgraph, 'batch_size', 1) != 1:
        raise ValueError('one graph per call (the reference function walks one graph, skeleton_matching_utils.py:33-55)')
    cams = list(parameters.used_cameras_skeleton_matching)
    ahead = getattr(subgraph, '_ahead', None)
    if ahead is not None and ahead.matches(outputs, CLASSIFICATION_THRESHOLD) and ahead.engine.params is parameters:
        # the scores GAT2.forward returned for this very graph, untouched: clustering and the persons' MLP rows were queued behind them
        # (runtime.queue_proposals) -- provided the indices are the graph's own edge-node ids, as below
        idx = torch.as_tensor(indices).reshape(-1)
        own = subgraph.__dict__.get('_en_ids')
        if own is None:
            own = subgraph.__dict__['_en_ids'] = torch.arange(subgraph.H, subgraph.H + subgraph.M, dtype=torch.int64)
        if idx.device.type == 'cpu' and idx.dtype == torch.int64 and idx.numel() == own.numel() and torch.equal(idx, own):
            rows = runtime.take_proposals(ahead, jsons_for_head, cams)
            if rows is not None:
                return [{cam: (None if row[c] < 0 else row[c]) for c, cam in enumerate(cams)} for row in rows]
    eng = runtime.shared_engine()
    db = subgraph.device_batch(eng)
    if type(outputs) is list:
        outputs = torch.tensor(outputs, dtype=torch.float32)
    outputs = outputs.reshape(-1).float()
    idx = torch.as_tensor(indices).reshape(-1).long()
    H = subgraph.H
    if idx.numel() != subgraph.M or (idx.numel() and int(idx[0]) != H):
        raise ValueError('indices must be the edge-node ids of the graph')
    scores = outputs.to(eng.device)[idx.to(eng.device)]
    if eng.params is not parameters:
        raise RuntimeError('engine / parameters mismatch')
    eng.set_threshold(CLASSIFICATION_THRESHOLD)
    persons, n_persons = eng.cluster(db, scores)
    # the caller goes on to the 3D stage with these persons (metrics_from_model.py:243-277): their MLP input rows in ONE launch
    # now instead of one launch per person later (runtime.prefetch_mlp_rows), queued behind the clustering so that the frame pays
    # one synchronisation for both
    ahead = eng.mlp_input_rows(db, persons, n_persons) if jsons_for_head is not None and runtime.prefetch_enabled() else None
    eng.sync_status()
    n = int(n_persons[0])
    rows = persons[0, :n].cpu().tolist()
    if ahead is not None and n:
        runtime.prefetch_mlp_rows(eng, db, persons, n_persons, rows, jsons_for_head, cams, launched=ahead)
    return [{cam: (None if row[c] < 0 else row[c]) for c, cam in enumerate(cams)} for row in rows]

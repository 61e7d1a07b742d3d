"""Bare-name stand-in for the ONE DGL call the reference's inference-time callers make themselves:
`dgl.batch(graphs)` in the collate function of test/sm_metrics_without_gt.py:46-64 (and
skeleton_matching/train_skeleton_matching.py:67-84).  With this directory on sys.path in place of the reference's source
directories `import dgl` resolves here and `dgl.batch` returns the engine's frame batch over the graph
handles of 3d_multi_pose_estimator_amd.graph_generator -- there is no DGL on the MI355X path."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
batch = _il.import_module('3d_multi_pose_estimator_amd.graph_generator').batch

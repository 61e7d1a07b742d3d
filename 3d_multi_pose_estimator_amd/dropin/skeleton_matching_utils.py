"""Bare-name alias so that the reference's callers (`from skeleton_matching_utils import ...`, e.g.
test/metrics_from_model.py:12-24) resolve to the MI355X implementation when this directory is
put on sys.path in place of ../skeleton_matching, ../utils and ../ (see INTEGRATION.md)."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_m = _il.import_module('3d_multi_pose_estimator_amd.skeleton_matching_utils')
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})

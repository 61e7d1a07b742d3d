"""Importable alias of the package `3d_multi_pose_estimator_amd` (a Python identifier
cannot start with a digit): ``import mpe_amd`` / ``from mpe_amd import pipeline``."""
import importlib
import sys

_pkg = importlib.import_module('3d_multi_pose_estimator_amd')
for _sub in ('parameters', 'calibration', 'synthetic', 'packing', 'lib', 'pipeline'):
    setattr(_pkg, _sub, importlib.import_module('3d_multi_pose_estimator_amd.' + _sub))
sys.modules[__name__] = _pkg

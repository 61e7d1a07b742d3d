"""Process-wide engines for the per-frame API mirrors (gat2.GAT2, graph_generator.*, ...).

The reference's symbols work on one frame per call; here each call becomes a 1-frame batch
through the same C-ABI entry points the batched engine uses.  Engines without weights
(clustering, featurisation, triangulation) are shared; GAT2 / PoseEstimatorMLP instances own
theirs because weights are frozen into a context's padded device copies.
"""
import os

import numpy as np

_shared = {}


def max_persons_per_camera():
    return int(os.environ.get('MPE_MAX_PERSONS_PER_CAMERA', '10'))


def device():
    return os.environ.get('MPE_DEVICE', 'cuda:0')


def new_engine(params=None, calib=None, max_frames=1):
    from .parameters import parameters as default_params
    from .pipeline import Engine, explicit_m_cap
    p = params or default_params
    hpf = len(p.used_cameras_skeleton_matching) * max_persons_per_camera()
    # the mirrors also take the graphs of mode='test_generated' (explicit edge-node lists: up to explicit_m_cap per graph)
    return Engine(params, calib, max_frames=max_frames, max_persons_per_camera=max_persons_per_camera(),
                  device=device(), max_edge_nodes_per_frame=explicit_m_cap(hpf) or None)


def shared_engine(params=None, calib=None, max_frames=1):
    """Weight-less engine keyed by the calibration it was built from; rebuilt larger when a batch of graphs needs more
    frames than it has."""
    key = 'default' if calib is None else _calib_key(calib)
    if key in _shared and _shared[key].max_frames < max_frames:
        _shared.pop(key).close()
    if key not in _shared:
        _shared[key] = new_engine(params, calib, max_frames=max(1, int(max_frames)))
    return _shared[key]


def _calib_key(calib):
    return hash((calib.K32.tobytes(), calib.dist.tobytes(), calib.P.tobytes(), calib.T_i32.tobytes()))


def calibration_from_dicts(params, camera_matrices, distortion_coefficients, projection_matrices):
    """Calibration object from the explicit dicts `triangulate` receives
    (reference pose_estimator_utils.py:52)."""
    from .calibration import Calibration, TransformManager
    names = list(params.camera_names)
    tm = TransformManager()
    for c in names:
        T = np.eye(4)
        T[0:3, :] = np.asarray(projection_matrices[c], np.float64)
        tm.add_transform('root', c, T)
    cal = Calibration(params, tm)
    for i, c in enumerate(names):
        cal.K32[i] = np.asarray(camera_matrices[c], np.float32)
        cal.dist[i] = np.asarray(distortion_coefficients[c], np.float64)
        cal.P[i] = np.asarray(projection_matrices[c], np.float64)
    return cal

"""Camera calibration for the inference path.

Restates the "calibration globals" every hot-path module of the reference builds at
import time (graph_generator.py:32-52, pose_estimator_dataset_from_json.py:28-47,
metrics_from_triangulation.py:54-72, pose_estimator_utils.py:17-30):

  T_d[c]  = tm.get_transform("root", cam)        4x4 f64 (stored matrix)
  T_i[c]  = tm.get_transform(cam, "root")        4x4 f64, inverse of T_d
  K[c]    = [[fx,0,cx],[0,fy,cy],[0,0,1]]        f32 (torch.tensor of python floats)
  Kinv[c] = torch.inverse(K[c])                  f32
  centre  = T_i(f32) @ [0,0,0,1]                 f32
  dist[c] = [kd0, kd1, p1, p2, kd2]              f64 (OpenCV order k1,k2,p1,p2,k3)
  P[c]    = T_d[c][0:3, :]                       3x4 f64

The extrinsics come from a pickled ``pytransform3d.transform_manager.TransformManager``
(pytransform3d 1.9.1, not vendored by the reference).  That package is not in this
image; `TransformManager` below restates the two code paths the reference uses
(`get_transform` of a stored edge and of its reverse = ``numpy.linalg.inv`` of the
stored 4x4, as pytransform3d's ``invert_transform`` does) and the pickle is read with
a restricted unpickler that only materialises numpy arrays.
"""
import io
import json
import os
import pickle

import numpy as np
import torch

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


class TransformManager:
    """Minimal stand-in for pytransform3d's TransformManager (direct edges only)."""

    def __init__(self, transforms=None):
        self.transforms = dict(transforms or {})

    def __setstate__(self, state):
        self.transforms = dict(state.get('transforms', {}))

    def add_transform(self, from_frame, to_frame, A2B):
        self.transforms[(from_frame, to_frame)] = np.asarray(A2B, dtype=np.float64)
        return self

    def get_transform(self, from_frame, to_frame):
        key = (from_frame, to_frame)
        if key in self.transforms:
            return self.transforms[key]
        rev = (to_frame, from_frame)
        if rev in self.transforms:
            return np.linalg.inv(self.transforms[rev])
        raise KeyError('Cannot compute path from frame %r to frame %r' % key)


class _Opaque:
    """Placeholder for pickled classes we do not need (scipy csr_matrix, ...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self._state = state


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == 'pytransform3d.transform_manager' and name == 'TransformManager':
            return TransformManager
        if module in ('numpy.core.multiarray', 'numpy._core.multiarray') and \
                name in ('_reconstruct', 'scalar'):
            import numpy.core.multiarray as m
            return getattr(m, name)
        if module == 'numpy' and name in ('ndarray', 'dtype'):
            return getattr(np, name)
        return type(name, (_Opaque,), {})


def load_transform_manager(path):
    """Load a ``tm_*.pickle`` (reference format) or our ``tm_*.json`` (hex floats)."""
    if path.endswith('.json'):
        with open(path) as fh:
            doc = json.load(fh)
        tm = TransformManager()
        for t in doc['transforms']:
            mat = np.array([[float.fromhex(x) for x in row] for row in t['matrix']], dtype=np.float64)
            tm.add_transform(t['from'], t['to'], mat)
        return tm
    with open(path, 'rb') as fh:
        tm = _RestrictedUnpickler(io.BytesIO(fh.read())).load()
    if not isinstance(tm, TransformManager):
        raise TypeError('%s does not hold a TransformManager' % path)
    return tm


def default_transform_path(params):
    """Resolve ``parameters.transformations_path`` the way the reference does (cwd
    relative), falling back to the calibration shipped with this package."""
    p = params.transformations_path
    if p and os.path.exists(p):
        return p
    base = os.path.splitext(os.path.basename(p or 'tm_panoptic.pickle'))[0]
    return os.path.join(_DATA_DIR, base + '.json')


def camera_matrix_f32(params, cam_idx):
    """3x3 f32 intrinsics, as pose_estimator_utils.camera_matrix (reference :17-30)."""
    return torch.tensor([[params.fx[cam_idx], 0.0, params.cx[cam_idx]],
                         [0.0, params.fy[cam_idx], params.cy[cam_idx]],
                         [0.0, 0.0, 1.0]])


class Calibration:
    """All per-camera constants of the path, indexed by position in
    ``params.camera_names`` (== ``used_cameras`` for the shipped presets)."""

    def __init__(self, params, tm=None):
        if tm is None:
            tm = load_transform_manager(default_transform_path(params))
        self.params = params
        self.tm = tm
        names = list(params.camera_names)
        self.names = names
        n = len(names)
        self.T_d = np.zeros((n, 4, 4), np.float64)
        self.T_i = np.zeros((n, 4, 4), np.float64)
        self.T_i32 = np.zeros((n, 4, 4), np.float32)
        self.K32 = np.zeros((n, 3, 3), np.float32)
        self.Kinv32 = np.zeros((n, 3, 3), np.float32)
        self.centre32 = np.zeros((n, 4), np.float32)
        self.dist = np.zeros((n, 5), np.float64)
        self.P = np.zeros((n, 3, 4), np.float64)
        for i, cam in enumerate(names):
            self.T_d[i] = tm.get_transform('root', cam)
            self.T_i[i] = tm.get_transform(cam, 'root')
            ti = torch.from_numpy(self.T_i[i]).type(torch.float32)
            self.T_i32[i] = ti.numpy()
            k = camera_matrix_f32(params, params.cameras[i])
            self.K32[i] = k.numpy()
            self.Kinv32[i] = torch.inverse(k).numpy()
            self.centre32[i] = torch.matmul(ti, torch.tensor([0.0, 0.0, 0.0, 1.0])).numpy()
            ci = params.cameras[i]
            self.dist[i] = [params.kd0[ci], params.kd1[ci], params.p1[ci], params.p2[ci], params.kd2[ci]]
            self.P[i] = self.T_d[i][0:3, :]

    @property
    def n_cameras(self):
        return len(self.names)

    def index(self, cam_name):
        return self.names.index(cam_name)

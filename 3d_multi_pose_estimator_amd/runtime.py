"""Process-wide engines for the per-frame API mirrors (gat2.GAT2, graph_generator.*, ...).

The reference's symbols work on one frame per call; here each call becomes a 1-frame batch
through the same C-ABI entry points the batched engine uses.  Engines without weights
(clustering, featurisation, triangulation) are shared; GAT2 / PoseEstimatorMLP instances own
theirs because weights are frozen into a context's padded device copies.
"""
import os

import numpy as np

_shared = {}


class ParamWatch:
    """(data_ptr, version) of every parameter of a module tree, without walking the tree per call (0.13 ms per forward in the
    one-frame-per-call loop) and without trusting that the tree never changes: the walk's result is kept, and every call checks
    with plain dict lookups that it still describes the tree -- every module still sits under its parent's name, every parameter
    still IS the object registered under its owner's name, no module gained or lost a child or a parameter.  A swapped submodule,
    `layer.fc1.weight = nn.Parameter(...)`, a pruning re-parametrisation or `register_parameter` rebuilds the snapshot; an
    in-place update or a `.to()` changes the tuple as before."""

    def __init__(self, root):
        self.root = root
        self.links = None        # (parent, name, child, n_children of child, n_parameters of child)
        self.slots = None        # (owner, name, parameter)

    def _rebuild(self):
        self.links, self.slots = [], []
        stack = [self.root]
        self.root_counts = (len(self.root._modules), len(self.root._parameters))
        while stack:
            mod = stack.pop()
            for name, p in mod._parameters.items():
                if p is not None:
                    self.slots.append((mod, name, p))
            for name, child in mod._modules.items():
                if child is not None:
                    self.links.append((mod, name, child, len(child._modules), len(child._parameters)))
                    stack.append(child)

    def _valid(self):
        if (len(self.root._modules), len(self.root._parameters)) != self.root_counts:
            return False
        for parent, name, child, n_mod, n_par in self.links:
            if parent._modules.get(name) is not child or len(child._modules) != n_mod or len(child._parameters) != n_par:
                return False
        for owner, name, p in self.slots:
            if owner._parameters.get(name) is not p:
                return False
        return True

    def version(self):
        if self.links is None or not self._valid():
            self._rebuild()
        return tuple((p.data_ptr(), p._version) for _, _, p in self.slots)


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


# ---- MLP input rows of a frame's persons, computed in ONE launch --------------------------------------------------------
# The reference's caller builds one PoseEstimatorDataset per PERSON (metrics_from_model.py:243-277), each from the JSON text
# of that person's skeletons.  get_person_proposal_from_network_output (the call right before, :214) knows all persons of the
# frame and has the frame's arrays on the device, so it computes every person's row with one mpe_mlp_input_rows launch and
# leaves them here, keyed by exactly the strings the caller is about to hand to PoseEstimatorDataset
# (json.dumps([jsons_for_head[head]]) per camera, :250-252).  A row is a pure function of those strings and the
# calibration, and the key holds both BY CONTENT (the calibration as the hash of its arrays, not the object's id, which
# CPython may hand to another calibration later), so an entry cannot be stale; PoseEstimatorDataset falls back to its
# own launch on a miss.
# MPE_DROPIN_PREFETCH=0 switches the prefetch off.
_row_cache = {}
_ROW_CACHE_CAP = 512


def prefetch_enabled():
    return os.environ.get('MPE_DROPIN_PREFETCH', '1') != '0'


def prefetch_mlp_rows(eng, db, persons, n_persons, rows, jsons_for_head, cams, launched=None):
    """rows: the frame's persons as lists of head ids per camera of `cams` (-1 = none).  launched: (rows, valid) device tensors
    of an mpe_mlp_input_rows launch the caller has already queued behind the clustering (one synchronisation for both)."""
    import json
    used = [c for c in eng.params.used_cameras]
    r, valid = launched if launched is not None else eng.mlp_input_rows(db, persons, n_persons)
    n = len(rows)
    r, valid = r[0, :n].cpu(), valid[0, :n].cpu()
    if len(_row_cache) > _ROW_CACHE_CAP:
        _row_cache.clear()
    text_of = getattr(jsons_for_head, 'json_text', None)

    def text(h):
        t = text_of(h) if text_of is not None else None
        return t if t is not None else json.dumps([jsons_for_head[h]])
    for p, row in enumerate(rows):
        key = tuple((cam, text(row[cams.index(cam)])) for cam in used if cam in cams and row[cams.index(cam)] >= 0)
        _row_cache[(_engine_key(eng), key)] = (r[p].clone(), bool(valid[p]))


def cached_mlp_row(eng, key):
    return _row_cache.get((_engine_key(eng), key))


def _engine_key(eng):
    k = getattr(eng, '_calib_content_key', None)
    if k is None:
        k = eng._calib_content_key = _calib_key(eng.calib)
    return k

"""API mirror of the reference's skeleton_matching/graph_generator.py (test path only).

``MergedMultipleHumansDataset(dict, mode='test', alt='3')`` (reference :519-566, 813-876) and
``HumanGraphFromView`` (:213-282, 444-508) keep their names, arguments and attributes, but no
DGL graph and no N x 902 matrix is built: the frame is packed into the structure-of-arrays
batch of include/mpe.h and the `graph` handed to GAT2 is a light handle on it.  The dense
feature matrix (``graph.ndata['h']``) is still available on request; it is assembled from the
HIP featurisation kernel's output (mpe_head_features).

Out of scope (training-only in the reference): mode != 'test', alternatives '1'/'2', the DGL
cache save/load.
"""
import torch

from . import runtime
from .packing import pack_frames, pairs_of_frame
from .parameters import parameters

if parameters.format == 'COCO':
    JOINTS_TYPES = {'0': "nose", '1': "left_eye", '2': "right_eye", '3': "left_ear", '4': "right_ear",
                    '5': "left_shoulder", '6': "right_shoulder", '7': "left_elbow", '8': "right_elbow",
                    '9': "left_wrist", '10': "right_wrist", '11': "left_hip", '12': "right_hip",
                    '13': "left_knee", '14': "right_knee", '15': "left_ankle", '16': "right_ankle", '17': "neck"}
else:
    raise Exception('Format not supported by the MI355X path: %r' % (parameters.format,))

_PER_JOINT = ['i', 'j', 'valid', 'prob', 'line_pX', 'line_pY', 'line_pZ', 'line_vX', 'line_vY', 'line_vZ']


def _features_alt3(params=parameters):
    names = ['head', 'edge_node']
    for cam in params.used_cameras_skeleton_matching:
        for joint in JOINTS_TYPES.values():
            names += ['%s_%s_%s' % (cam, joint, f) for f in _PER_JOINT]
    return names


FEATURES = {'3': _features_alt3()}
RELATIONS = {'3': sorted(['h_h', 'link', 'link_link'])}


class HumanGraphFromView:
    """Static helpers of the reference class; instances are not needed on this path."""
    joints = JOINTS_TYPES

    @staticmethod
    def get_all_features(alt='1'):
        # the callers use the default argument only to size the network input
        # (metrics_from_model.py:45); alternative '3' is the deployed one
        return FEATURES['3']

    @staticmethod
    def get_rels(alt='1'):
        return RELATIONS['3']

    @staticmethod
    def get_cam_types():
        return parameters.used_cameras_skeleton_matching


class _NData(dict):
    def __init__(self, graph):
        super().__init__()
        self._g = graph

    def __missing__(self, key):
        if key == 'h':
            v = self._g._dense_features()
            self[key] = v
            return v
        raise KeyError(key)


class FrameGraph:
    """What the callers treat as the DGL graph of one frame (metrics_from_model.py:201-209)."""

    def __init__(self, packed, params):
        self.packed = packed                 # PackedBatch with n_frames == 1
        self.params = params
        self._dev = {}
        self.ndata = _NData(self)
        self.edata = {}
        h0, H, e0, M = packed.frame_counts(0)
        self.H, self.M = H, M

    def to(self, device):
        return self

    def number_of_nodes(self):
        return self.H + self.M

    num_nodes = number_of_nodes

    def nodes(self):
        return torch.arange(self.H + self.M, dtype=torch.int32)

    def edges(self):
        """(src, dst) int32 in the reference's edge order (graph_generator.py:474-477, 632-651)."""
        pairs = pairs_of_frame(self.packed.slot_n[0])
        H = self.H
        src = list(range(H))
        dst = list(range(H))
        for m, (h1, h2) in enumerate(pairs):
            X = H + m
            src += [int(h1), X, int(h2), X, X]
            dst += [X, int(h1), X, int(h2), X]
        return torch.tensor(src, dtype=torch.int32), torch.tensor(dst, dtype=torch.int32)

    def device_batch(self, engine):
        key = id(engine)
        if key not in self._dev:
            self._dev[key] = engine.to_device(self.packed)
        return self._dev[key]

    def _dense_features(self):
        eng = runtime.shared_engine()
        blk = eng.head_features(self.device_batch(eng)).cpu()      # [H][J][10]
        J = blk.shape[1]
        F = 2 + eng.V * J * 10
        feats = torch.zeros((self.H + self.M, F))
        feats[:self.H, 0] = 1.0
        feats[self.H:, 1] = 1.0
        for h in range(self.H):
            c = int(self.packed.head_cam[h])
            feats[h, 2 + c * J * 10: 2 + (c + 1) * J * 10] = blk[h].reshape(-1)
        return feats


class MergedMultipleHumansDataset:
    def __init__(self, paths, probabilities=[1.], limit='100000000', alt=None, mode='train', force_reload=False,
                 verbose=True, debug=False, raw_dir='.'):
        if alt is None:
            raise ValueError('Alt is None')
        if mode != 'test' or str(alt) != '3' or not isinstance(paths, dict):
            raise NotImplementedError('the MI355X path covers mode="test", alt="3" with a frame dict '
                                      '(reference graph_generator.py:813-876); training graphs are out of scope')
        self.mode, self.alt, self.limit, self.debug = mode, alt, limit, debug
        self.graphs, self.labels = [], []
        self.data = {'edge_nodes_indices': [], 'nodes_camera': []}
        packed = pack_frames([paths], parameters, keep_json=True)
        self.jsons_for_head = packed.jsons_for_head[0]
        self.skeleton_index = {i: int(v) for i, v in enumerate(packed.skeleton_index)}
        h0, H, e0, M = packed.frame_counts(0)
        if M > 0:                       # no edge-node -> no graph (reference :866)
            sm = list(parameters.used_cameras_skeleton_matching)
            self.graphs.append(FrameGraph(packed, parameters))
            self.labels.append(torch.zeros((M, 1), dtype=torch.float64))
            self.data['edge_nodes_indices'].append(torch.arange(H, H + M, dtype=torch.int64).unsqueeze(1))
            self.data['nodes_camera'].append([sm[int(c)] for c in packed.head_cam] + [''] * M)

    def __getitem__(self, idx):
        return self.graphs[idx], self.labels[idx], self.data['edge_nodes_indices'][idx], self.data['nodes_camera'][idx]

    def __len__(self):
        return len(self.graphs)

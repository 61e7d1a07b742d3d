"""API mirror of the reference's skeleton_matching/graph_generator.py (inference-time paths).

``MergedMultipleHumansDataset`` (reference :519-566) keeps its name, arguments and attributes for
  * ``(dict, mode='test', alt='3')`` -- one frame, implicit topology (process_test, :813-876), and
  * ``(list_of_files, probabilities, limit, mode='test_generated', alt='3')`` -- the scenes test/sm_metrics_without_gt.py:108
    composes from single-person recordings with the TRAINING graph synthesis (process_training, :672-810): samples drawn
    with Python's ``random`` in the reference's call sequence (same seed => same scenes), heads grouped by person, one
    edge-node per ORDERED head pair with a label, plus the DGL cache of :884-917 (``cache/`` in the working directory, our
    own file format).
``HumanGraphFromView`` (:213-282, 444-508) keeps its static helpers.  No DGL graph and no N x 902 matrix is built: a graph is a
light handle on the structure-of-arrays batch of include/mpe.h (with an explicit edge-node list for the generated scenes);
``batch(graphs)`` is what ``dgl.batch`` is to the reference (train_skeleton_matching.py:67-84): the engine's frame batch.
The dense feature matrix (``graph.ndata['h']``) is still available on request; it is assembled from the HIP featurisation
kernel's output (mpe_head_features).

Out of scope (training-only in the reference): mode 'train' / 'dev' (data augmentation, add_data_to_json), alternatives '1'/'2'.
"""
import json
import os
import pickle
import random

import numpy as np
import torch

from . import runtime
from .packing import concat_packed, generated_scene, pack_frames, pack_scenes
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
    """What the callers treat as the DGL graph: of one frame (metrics_from_model.py:201-209), of one generated scene
    (sm_metrics_without_gt.py:112-121) or, after batch(), of several.  Nodes are numbered graph by graph: heads, then
    edge-nodes (dgl.batch relabels the same way)."""

    def __init__(self, packed, params):
        self.packed = packed                 # PackedBatch, one frame per graph
        self.params = params
        self._dev = {}
        self.ndata = _NData(self)
        self.edata = {}
        self.batch_size = packed.n_frames
        counts = [packed.frame_counts(f) for f in range(packed.n_frames)]
        self.batch_num_heads = [c[1] for c in counts]
        self.batch_num_edge_nodes = [c[3] for c in counts]
        self.H, self.M = sum(self.batch_num_heads), sum(self.batch_num_edge_nodes)

    def to(self, device):
        return self

    def number_of_nodes(self):
        return self.H + self.M

    num_nodes = number_of_nodes

    def batch_num_nodes(self):
        return torch.tensor([h + m for h, m in zip(self.batch_num_heads, self.batch_num_edge_nodes)], dtype=torch.int64)

    def nodes(self):
        return torch.arange(self.H + self.M, dtype=torch.int32)

    def node_offsets(self):
        """First node id of every graph of the batch (+ the total)."""
        off = [0]
        for h, m in zip(self.batch_num_heads, self.batch_num_edge_nodes):
            off.append(off[-1] + h + m)
        return off

    def edges(self):
        """(src, dst) int32 in the reference's edge order (graph_generator.py:474-477, 632-651)."""
        src, dst = [], []
        base = 0
        for f, (H, M) in enumerate(zip(self.batch_num_heads, self.batch_num_edge_nodes)):
            src += list(range(base, base + H))
            dst += list(range(base, base + H))
            for m, (h1, h2) in enumerate(self.packed.pairs(f)):
                X = base + H + m
                h1, h2 = base + int(h1), base + int(h2)
                src += [h1, X, h2, X, X]
                dst += [X, h1, X, h2, X]
            base += H + M
        return torch.tensor(src, dtype=torch.int32), torch.tensor(dst, dtype=torch.int32)

    def device_batch(self, engine):
        key = id(engine)
        if key not in self._dev:
            self._dev[key] = engine.to_device(self.packed)
        return self._dev[key]

    def _dense_features(self):
        eng = runtime.shared_engine(max_frames=self.batch_size)
        blk = eng.head_features(self.device_batch(eng)).cpu()      # [H][J][10]
        J = blk.shape[1]
        F = 2 + eng.V * J * 10
        feats = torch.zeros((self.H + self.M, F))
        off = self.node_offsets()
        h = 0
        for f, Hf in enumerate(self.batch_num_heads):
            feats[off[f]:off[f] + Hf, 0] = 1.0
            feats[off[f] + Hf:off[f + 1], 1] = 1.0
            for i in range(Hf):
                c = int(self.packed.head_cam[h])
                feats[off[f] + i, 2 + c * J * 10: 2 + (c + 1) * J * 10] = blk[h].reshape(-1)
                h += 1
        return feats


def batch(graphs):
    """Several graphs as one (the reference's dgl.batch, train_skeleton_matching.py:82, sm_metrics_without_gt.py:60):
    one frame per graph in the engine's batch, nodes relabelled graph by graph."""
    graphs = list(graphs)
    if len(graphs) == 1:
        return graphs[0]
    return FrameGraph(concat_packed([g.packed for g in graphs]), graphs[0].params)


class MergedMultipleHumansDataset:
    path_save = 'cache/'

    def __init__(self, paths, probabilities=[1.], limit='100000000', alt=None, mode='train', force_reload=False,
                 verbose=True, debug=False, raw_dir='.'):
        if alt is None:
            raise ValueError('Alt is None')
        if str(alt) != '3':
            raise NotImplementedError('the MI355X path covers graph alternative "3" (the deployed one)')
        if mode not in ('test', 'test_generated'):
            raise NotImplementedError('the MI355X path covers mode="test" (one frame dict, reference graph_generator.py:813-876) '
                                      'and mode="test_generated" (scenes composed from single-person files, :672-810); the '
                                      'training modes add data augmentation (add_data_to_json) and are out of scope')
        self.name = 'MergedMultipleHumansDataset'
        self.mode, self.alt, self.limit, self.debug = mode, alt, limit, debug
        self.probabilities = probabilities
        self.force_reload = force_reload or mode == 'test'          # reference :553-554
        self.graphs, self.labels = [], []
        self.data = {'edge_nodes_indices': [], 'nodes_camera': []}
        self.inputs, self.inputs_indices = [], []
        if type(paths) == list:
            for path in paths:
                print('PATH', path)
                data = json.loads(open(path, 'rb').read())
                indices = list(range(len(data)))
                if mode != 'test':
                    random.shuffle(indices)                          # reference :534-535: the module-level generator
                self.inputs.append(data)
                self.inputs_indices.append(indices)
        elif type(paths) == dict:
            self.inputs.append(paths)
            self.inputs_indices.append(list(range(len(paths))))
        else:
            raise Exception('Unhandled type for MergedMultipleHumansDataset')
        # DGLDataset._load: cache unless force_reload, else process() + save()
        if not self.force_reload and self.has_cache():
            self.load()
        else:
            self.process()
            self.save()

    # ---- reference :657-665 ----
    def process(self):
        if self.mode != 'test':
            self.process_training()
        else:
            self.process_test()

    def process_test(self):
        if len(self.inputs) != 1:
            raise AssertionError('For testing, please provide __ONE__ single JSON file')
        iterate_over = self.inputs[0] if type(self.inputs[0]) == list else self.inputs      # a file of frames / one frame dict
        sm = list(parameters.used_cameras_skeleton_matching)
        idx = 0
        for json_view in iterate_over:
            if idx == self.limit:
                break
            idx += 1
            packed = pack_frames([json_view], parameters, keep_json=True)
            self.jsons_for_head = packed.jsons_for_head[0]          # of the last frame seen, as in the reference (:577-578)
            self.skeleton_index = {i: int(v) for i, v in enumerate(packed.skeleton_index)}
            h0, H, e0, M = packed.frame_counts(0)
            if M > 0:                   # no edge-node -> no graph (reference :866)
                self.graphs.append(FrameGraph(packed, parameters))
                self.labels.append(torch.zeros((M, 1), dtype=torch.float64))
                self.data['edge_nodes_indices'].append(torch.arange(H, H + M, dtype=torch.int64).unsqueeze(1))
                self.data['nodes_camera'].append([sm[int(c)] for c in packed.head_cam] + [''] * M)

    def sample_scenes(self):
        """The generator of process_training (reference :674-693): per scene `random.randint(1, n_files)` people, taken
        from the files with the highest probabilities (np.argpartition, the reference's own expression), each giving the
        last of its shuffled frame indices; sampling stops for good when a chosen file has run dry."""
        for _ in range(self.limit):
            if all(len(l) == 0 for l in self.inputs):
                break
            views = []
            num_people = random.randint(1, len(self.inputs))
            filter_prob = np.array(self.probabilities)
            max_indx = np.argpartition(filter_prob, -num_people)[-num_people:]
            for index in max_indx:
                try:
                    json_index = self.inputs_indices[index].pop()
                    views.append(self.inputs[index][json_index])
                except IndexError:
                    return
            if len(views) == 0:
                continue
            yield views

    def process_training(self):
        self.jsons_for_head, self.skeleton_index = dict(), dict()
        idx = 0
        for views in self.sample_scenes():
            if idx % 1000 == 0:
                print(idx)
            idx += 1
            scene = generated_scene(views, parameters)
            H, M = len(scene['heads']), len(scene['pairs'])
            if M == 0:                  # reference :799: no edge-node, no graph
                continue
            self.graphs.append(FrameGraph(pack_scenes([scene], parameters), parameters))
            self.labels.append(torch.tensor(scene['labels'], dtype=torch.float64).unsqueeze(1))
            self.data['edge_nodes_indices'].append(torch.arange(H, H + M, dtype=torch.int64).unsqueeze(1))
            self.data['nodes_camera'].append([h[0] for h in scene['heads']] + [''] * M)

    # ---- the cache of reference :560-563, 884-917 (file format is this package's own) ----
    def get_dataset_name(self):
        graphs_path = self.name + '_' + self.mode + '_alt_' + self.alt + '_s_' + str(self.limit) + '.bin'
        info_path = self.name + '_info_' + self.mode + '_alt_' + self.alt + '_s_' + str(self.limit) + '.pkl'
        return graphs_path, info_path

    def save(self):
        if self.debug or self.mode == 'test':
            return
        graphs_path, info_path = tuple((self.path_save + x) for x in self.get_dataset_name())
        print(graphs_path, info_path)
        os.makedirs(os.path.dirname(self.path_save), exist_ok=True)
        with open(graphs_path, 'wb') as fh:
            pickle.dump([g.packed for g in self.graphs], fh)
        with open(info_path, 'wb') as fh:
            pickle.dump({'edge_nodes_indices': self.data['edge_nodes_indices'], 'labels': self.labels,
                         'nodes_camera': self.data['nodes_camera']}, fh)

    def load(self):
        graphs_path, info_path = tuple((self.path_save + x) for x in self.get_dataset_name())
        with open(graphs_path, 'rb') as fh:
            self.graphs = [FrameGraph(p, parameters) for p in pickle.load(fh)]
        with open(info_path, 'rb') as fh:
            info = pickle.load(fh)
        self.data['edge_nodes_indices'] = info['edge_nodes_indices']
        self.data['nodes_camera'] = info['nodes_camera']
        self.labels = info['labels']
        print(f'LOADED {len(self.graphs)} graphs')

    def has_cache(self):
        graphs_path, info_path = tuple((self.path_save + x) for x in self.get_dataset_name())
        if self.debug:
            return False
        return os.path.exists(graphs_path) and os.path.exists(info_path)

    def __getitem__(self, idx):
        return self.graphs[idx], self.labels[idx], self.data['edge_nodes_indices'][idx], self.data['nodes_camera'][idx]

    def __len__(self):
        return len(self.graphs)

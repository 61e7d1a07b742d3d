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
        self._fits = set()
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
        """The graph's arrays on the engine's device: uploaded once per device (the GAT2 / clustering / dataset mirrors
        each own an engine), capacity checked per engine."""
        cap = (engine.max_frames, engine.hpf, engine.V, engine.J, engine.m_frame)
        if cap not in self._fits:                # (per capacity, not per call: three engines see every frame of the per-frame loop)
            engine.check_capacity(self.packed)
            self._fits.add(cap)
        key = engine.device
        dev = self._dev.get(key)
        if dev is None:
            dev = self._dev[key] = self.packed.to(key)
        return dev

    def _dense_features(self):
        """graph.ndata['h']: the dense N x F rows (reference :444-508, 629-631), assembled ON THE DEVICE from the featurisation
        kernel's output (mpe_head_features) -- a tensor on the engine's device, so the callers' `.to(device)` /
        `.float()` hand the same object back and GAT2.forward recognises the graph's own rows without comparing them."""
        eng = runtime.shared_engine(max_frames=self.batch_size)
        if self.batch_size == 1 and self.H + self.M:
            return eng.dense_rows(self.device_batch(eng))           # one graph: two launches of the library instead of a dozen indexing ops
        blk = eng.head_features(self.device_batch(eng))            # [H][J][10] on the device
        J = blk.shape[1]
        F = 2 + eng.V * J * 10
        dev = blk.device
        feats = torch.zeros((self.H + self.M, F), device=dev)
        if self.batch_size == 1:
            H = self.H
            feats[:H, 0] = 1.0
            feats[H:, 1] = 1.0
            head_rows = torch.arange(H, device=dev)
        else:
            off = self.node_offsets()
            hr = np.concatenate([np.arange(off[f], off[f] + h) for f, h in enumerate(self.batch_num_heads)]) if self.H else np.zeros(0, np.int64)
            head_rows = torch.from_numpy(hr.astype(np.int64)).to(dev)
            kind = torch.ones(self.H + self.M, dtype=torch.int64, device=dev)
            kind[head_rows] = 0
            feats[torch.arange(self.H + self.M, device=dev), kind] = 1.0
        if self.H:
            cam = torch.from_numpy(np.ascontiguousarray(self.packed.head_cam, np.int64)).to(dev)
            cols = (2 + cam * (J * 10)).unsqueeze(1) + torch.arange(J * 10, device=dev).unsqueeze(0)
            feats[head_rows.unsqueeze(1), cols] = blk.reshape(self.H, J * 10)
        return feats


class _LazyHeadJsons(dict):
    """jsons_for_head of one frame (reference :597-598: head id -> skeleton dict), filled camera by camera on first use:
    the native packer has already turned the frame into arrays, the Python dicts are only needed by callers that go on to
    the 3D stage (metrics_from_model.py:250-252)."""

    def __init__(self, frame, packed):
        super().__init__()
        self._frame, self._packed, self._done = frame, packed, False

    def _fill(self):
        if self._done:
            return
        self._done = True
        sm = list(parameters.used_cameras_skeleton_matching)
        pb = self._packed
        h = 0
        for s in range(pb.slot_cam.shape[1]):
            c, n = int(pb.slot_cam[0, s]), int(pb.slot_n[0, s])
            if c < 0:
                continue
            skeletons = json.loads(self._frame[sm[c]][0]) if n else []
            for _ in range(n):
                dict.__setitem__(self, h, skeletons[int(pb.skeleton_index[h])])
                h += 1

    def json_text(self, h):
        """'[<skeleton h>]' as json.dumps writes it, cut out of the camera's string instead of serialised again -- what the
        reference's caller produces with json.dumps([jsons_for_head[h]]) (metrics_from_model.py:250-252) WHEN the camera string
        was itself written by json.dumps with its default separators (:186-190 does exactly that).  None when the text cannot
        be cut safely (then the caller's string is rebuilt with json.dumps)."""
        if not hasattr(self, '_texts'):
            self._texts = {}
            sm = parameters.used_cameras_skeleton_matching
            pb = self._packed
            k = 0
            index_of = pb.skeleton_index.tolist()
            ext = getattr(pb, 'skeleton_extent', None)
            ext = ext.tolist() if ext is not None else None
            for c, n in zip(pb.slot_cam[0].tolist(), pb.slot_n[0].tolist()):
                if c < 0 or n == 0:
                    continue
                text = self._frame[sm[c]][0]
                if ext is not None:
                    # the native packer's byte extents of the skeleton objects: a slice of the caller's own text.  Whether that slice is
                    # what json.dumps([jsons_for_head[h]]) writes is for the comparison in runtime._RecentRows to find out -- a key that
                    # differs is a miss (the dataset computes the row itself), never a wrong row
                    for _ in range(n):
                        a, b = ext[k]
                        self._texts[k] = '[' + text[a:b] + ']'
                        k += 1
                    continue
                pieces = None
                # skeleton dicts hold lists of numbers only (no nested dict): '}, {' separates them; an "ID" member may hold anything
                if text.startswith('[{') and text.endswith('}]') and '"ID"' not in text and '\\' not in text:
                    pieces = text[2:-2].split('}, {')
                for _ in range(n):
                    i = index_of[k]
                    self._texts[k] = '[{' + pieces[i] + '}]' if pieces is not None and i < len(pieces) else None
                    k += 1
        return self._texts.get(h)

    def __missing__(self, key):
        self._fill()
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        self._fill()
        return dict.__contains__(self, key)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)


def _pack_one(frame):
    """One frame dict -> PackedBatch through the native packer (csrc/packer.cpp, mpe_pack_views_into: the camera texts as the
    caller holds them -- the Python loops over 20 skeletons x 18 joints cost more than the frame's kernels, and so did
    serialising the frame into a document for mpe_pack_json); frames it declines (keys outside the configured joints,
    non-string entries, more skeletons than the engines hold, ...) go through the Python packer, which defines the accepted
    language and the error messages."""
    try:
        from .packing import pack_views
        pb = pack_views(frame, parameters, len(parameters.used_cameras_skeleton_matching) * runtime.max_persons_per_camera())
        pb.jsons_for_head = [_LazyHeadJsons(frame, pb)]
        return pb
    except (ValueError, ImportError, TypeError, KeyError, AttributeError, IndexError):
        pass
    return pack_frames([frame], parameters, keep_json=True)


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
            packed = _pack_one(json_view)
            self.jsons_for_head = packed.jsons_for_head[0]          # of the last frame seen, as in the reference (:577-578)
            self.skeleton_index = {i: int(v) for i, v in enumerate(packed.skeleton_index)}
            h0, H, e0, M = packed.frame_counts(0)
            if M > 0:                   # no edge-node -> no graph (reference :866)
                self.graphs.append(FrameGraph(packed, parameters))
                self.labels.append(torch.zeros((M, 1), dtype=torch.float64))
                self.data['edge_nodes_indices'].append(torch.arange(H, H + M, dtype=torch.int64).unsqueeze(1))
                self.data['nodes_camera'].append([sm[int(c)] for c in packed.head_cam] + [''] * M)
                if iterate_over is self.inputs:           # one frame dict (the per-frame loop of metrics_from_model.py:178-209): its scores are wanted next
                    runtime.start_frame(self.graphs[-1])

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

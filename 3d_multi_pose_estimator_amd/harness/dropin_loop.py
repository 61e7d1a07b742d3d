"""One frame per call through the drop-in mirrors: the call pattern of the reference's evaluation script
(test/metrics_from_model.py:178-294), as opposed to the frame batches of harness/common.py.

This is what a maintainer who applies INTEGRATION.md section 2 to the reference gets: for every frame a graph is built
(`MergedMultipleHumansDataset`), scored (`GAT2.__call__`), clustered (`get_person_proposal_from_network_output`), every person
is turned into an MLP row (`PoseEstimatorDataset`) and all rows of the frame go through `PoseEstimatorMLP` once.  The ORDER of
those calls and the shapes they receive are pinned against the reference's own loop by a fixture: `oracle/gen_dropin_trace.py`
runs the reference script with tracing wrappers around its symbols and stores the sequence under tests/golden/dropin/;
`tests/test_gpu_dropin.py::test_frame_loop_calls_the_mirrors_like_the_reference_script` compares this module's sequence with it.

Two clocks per frame, cut where the reference cuts them (graph matching = frame in hand -> person proposals, :179/:229; 3D =
proposals -> joints of every person, :236/:295), and a third the reference does not have: the host time spent INSIDE the
package's symbols (`inside_mirrors_ms`); the rest of a frame is the caller's own JSON and tensor handling.  The reference's
README quotes 31.67 ms + 19.65 ms per frame for its own path on its authors' GPU; bench.py prints this loop beside them
(`dropin_loop`).
"""
import contextlib
import io
import json
import time
from dataclasses import dataclass, field

import numpy as np
import torch

README_MS = {'graph_matching': 31.67, 'pose_3d': 19.65}      # reference README.md ("Time" table), its authors' hardware
SCORE_THRESHOLD = 0.5                                         # CLASSIFICATION_THRESHOLD of the script (:34)
GRAPH_LIMIT = 10000


def build_models(gat_sd, prm, mlp_sd):
    """The two network mirrors with their weights, constructed the way the script constructs the reference's (:73-100)."""
    from ..gat2 import GAT2
    from ..mlp import PoseEstimatorMLP
    from ..parameters import parameters as P
    matcher = GAT2(None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'],
                   torch.nn.LeakyReLU(), torch.nn.Sigmoid(), prm['in_drop'], prm['attn_drop'], prm['alpha'], prm['residual'], bias=True)
    matcher.load_state_dict({k: torch.as_tensor(v) for k, v in gat_sd.items()})
    with contextlib.redirect_stdout(io.StringIO()):          # the reference's constructor prints its input size
        lifter = PoseEstimatorMLP(input_dimensions=len(P.cameras) * len(P.joint_list) * P.numbers_per_joint,
                                  output_dimensions=3 * len(P.joint_list))
    lifter.load_state_dict({k: torch.as_tensor(v) for k, v in mlp_sd.items()})
    return matcher, lifter


def _describe(value):
    """Shape-level description of an argument (what the call-sequence fixture stores)."""
    if isinstance(value, torch.Tensor):
        return ['tensor', list(value.shape)]
    if isinstance(value, np.ndarray):
        return ['array', list(value.shape)]
    if isinstance(value, dict):                               # camera-keyed dicts by their keys, everything else by its size
        return ['dict', sorted(value)] if 0 < len(value) <= 8 and all(isinstance(k, str) for k in value) else ['dict', len(value)]
    if isinstance(value, (list, tuple)):
        return ['list', len(value)]
    if isinstance(value, (bool, int, float, str)) or value is None:
        return value
    if hasattr(value, 'number_of_nodes'):
        return ['graph', int(value.number_of_nodes())]
    return type(value).__name__


@dataclass
class MirrorCalls:
    """Every call into the package goes through here: host seconds inside the mirrors, optionally the sequence of calls."""
    seconds: float = 0.0
    trace: list = None
    by_symbol: dict = field(default_factory=dict)

    def __call__(self, name, fn, *args, **kwargs):
        t0 = time.perf_counter()
        out = fn(*args, **kwargs)
        dt = time.perf_counter() - t0
        self.seconds += dt
        self.by_symbol[name] = self.by_symbol.get(name, 0.0) + dt
        if self.trace is not None and not name.startswith('graph.'):      # (attribute reads of the graph object are not calls of the script)
            self.trace.append([name, [_describe(a) for a in args], {k: _describe(v) for k, v in sorted(kwargs.items())}])
        return out


@dataclass
class Clocks:
    matching_s: float = 0.0
    matching_frames: int = 0
    lifting_s: float = 0.0
    lifting_frames: int = 0
    frames: int = 0
    persons: int = 0
    started: float = field(default_factory=time.time)


def cameras_with_skeletons(frame):
    """The caller's own pre-processing (:182-191): the skeleton list of every camera is decoded and encoded again, cameras whose
    list is empty are left out."""
    kept = {}
    for cam, record in frame.items():
        skeletons = list(json.loads(record[0]))
        if skeletons:
            kept[cam] = [json.dumps(skeletons), record[1]]
    return kept


def match_frame(frame, matcher, calls, device):
    """Frame -> (scenario, proposals) or None when the frame has no cross-camera pair (no graph is built, :195-196)."""
    from ..graph_generator import MergedMultipleHumansDataset
    from ..parameters import parameters as P
    from ..skeleton_matching_utils import get_person_proposal_from_network_output
    scenario = calls('MergedMultipleHumansDataset', MergedMultipleHumansDataset, cameras_with_skeletons(frame), mode='test',
                     limit=GRAPH_LIMIT, debug=True, alt=P.graph_alternative, verbose=False)
    if not scenario.graphs:
        return None
    graph = scenario.graphs[0].to(device)
    edge_node_ids = scenario.data['edge_nodes_indices'][0].to(device)
    node_features = calls('graph.ndata[h]', lambda: graph.ndata['h']).to(device)
    matcher.g = graph                                        # the script hands the graph to the model and to every layer (:206-208)
    for layer in matcher.layers:
        layer.g = graph
    scores = torch.squeeze(calls('GAT2.__call__', matcher, node_features.float(), graph))
    proposals = calls('get_person_proposal_from_network_output', get_person_proposal_from_network_output, scores, graph,
                      torch.squeeze(edge_node_ids).to('cpu'), scenario.data['nodes_camera'][0], scenario.jsons_for_head, SCORE_THRESHOLD)
    return scenario, proposals


def lift_frame(scenario, proposals, lifter, calls, device):
    """Person proposals -> one list of per-joint (3,) arrays per person (:243-294): a dataset row per person from the JSON text of
    its skeletons, one MLP call for the frame, x10 (the network works in decimetres), joints cut out of the 54-vector."""
    from ..parameters import parameters as P
    from ..pose_estimator_dataset_from_json import PoseEstimatorDataset
    rows = []
    for person in proposals:
        views = {cam: [json.dumps([scenario.jsons_for_head[person[cam]]])] for cam in P.used_cameras if person[cam] is not None}
        sample = calls('PoseEstimatorDataset', PoseEstimatorDataset, views, P.cameras, P.joint_list, save=False)
        if len(sample):
            vector = sample[0][0]
            rows.append(vector.reshape([1, vector.size()[0]]).to(device))
    if not rows:
        return []
    out = calls('PoseEstimatorMLP.__call__', lifter, torch.cat(rows, dim=0).to(device))
    n_joints = len(P.joint_list)
    people = []
    for k in range(out.shape[0]):
        metres = (torch.squeeze(out[k]) * 10.).to('cpu')
        xs, ys, zs = metres[0::3], metres[1::3], metres[2::3]
        people.append([np.array([xs[j], ys[j], zs[j]]) for j in range(n_joints)])
    return people


def run(frames, matcher, lifter, warmup=3, device=None, trace=None):
    """frames: wire-format frame dicts, one call chain per frame.  -> dict(mean ms of both stages as the reference averages them
    -- over the frames that produced proposals / poses --, frames/s, host ms inside the mirrors, the last frame's poses).
    `trace`: a list that receives one list of [symbol, args, kwargs] descriptions per frame (the call-sequence fixture's form)."""
    device = device or torch.device('cuda')
    calls = MirrorCalls()
    clk = Clocks()
    last = None
    for index, frame in enumerate(frames):
        if index == warmup:                                  # everything before is warm-up: clocks start here
            torch.cuda.synchronize()
            calls.seconds = 0.0
            calls.by_symbol = {}
            clk = Clocks()
        if trace is not None:
            calls.trace = []
            trace.append(calls.trace)
        t0 = time.time()
        matched = match_frame(frame, matcher, calls, device)
        if matched is None:
            continue
        scenario, proposals = matched
        t1 = time.time()
        if len(proposals) > 0:
            clk.matching_s += t1 - t0
            clk.matching_frames += 1
        people = lift_frame(scenario, proposals, lifter, calls, device)
        if len(people) > 0:
            clk.lifting_s += time.time() - t1
            clk.lifting_frames += 1
        clk.frames += 1
        clk.persons += len(people)
        last = people
    torch.cuda.synchronize()
    wall = time.time() - clk.started
    n = max(1, clk.frames)
    return {'frames': clk.frames, 'persons_per_frame': clk.persons / n,
            'graph_matching_ms': 1e3 * clk.matching_s / max(1, clk.matching_frames), 'pose_3d_ms': 1e3 * clk.lifting_s / max(1, clk.lifting_frames),
            'ms_per_frame': 1e3 * wall / n, 'frames_per_s': clk.frames / wall if wall > 0 else 0.0,
            # host time inside the package's symbols per frame; the remainder is the caller's own Python (decoding and encoding
            # every camera list and every person, tensor reshapes, waiting for the MLP kernels in .to('cpu'), numpy boxing)
            'inside_mirrors_ms': 1e3 * calls.seconds / n,
            'inside_mirrors_by_symbol_ms': {k: round(1e3 * v / n, 4) for k, v in calls.by_symbol.items()},
            'reference_readme_ms': README_MS, 'last': last}

"""GPU: the reference's per-frame loop (test/metrics_from_model.py:178-294 and
metrics_from_triangulation.py:187-272) written against the drop-in modules, frame by frame,
compared with the fixtures the reference itself produced."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import CASES, GOLDEN, ROOT, load_case, oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dropin(gat_weights, mlp_weights):
    sys.path.insert(0, os.path.join(ROOT, '3d_multi_pose_estimator_amd', 'dropin'))
    from gat2 import GAT2 as GAT
    from graph_generator import MergedMultipleHumansDataset, HumanGraphFromView
    from pose_estimator_dataset_from_json import PoseEstimatorDataset
    from mlp import PoseEstimatorMLP
    from skeleton_matching_utils import get_person_proposal_from_network_output
    from pose_estimator_utils import camera_matrix, triangulate
    from parameters import parameters
    sd, prm = gat_weights
    model = GAT(None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'],
                torch.nn.LeakyReLU(), torch.nn.Sigmoid(), prm['in_drop'], prm['attn_drop'], prm['alpha'],
                prm['residual'], bias=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mlp = PoseEstimatorMLP(input_dimensions=len(parameters.cameras) * len(parameters.joint_list) * parameters.numbers_per_joint,
                           output_dimensions=54)
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_weights.items()})
    return locals()


@pytest.mark.parametrize('name', CASES + ['fz_random_shapes'])      # + 40 frames of random shape through the reference (round 6)
def test_reference_loop_on_dropin(dropin, name, mlp_weights):
    d = dropin
    parameters = d['parameters']
    device = torch.device('cuda')
    arr, frames = load_case(name)
    assert len(d['HumanGraphFromView'].get_all_features()) == 902
    for n, input_element in enumerate(frames):
        p = 'f%d_' % n
        processed_input = {}
        for cam in input_element:
            data = json.loads(input_element[cam][0])
            if data:
                processed_input[cam] = [json.dumps(data), input_element[cam][1]]
        scenario = d['MergedMultipleHumansDataset'](processed_input, mode='test', limit=10000, debug=True,
                                                    alt=parameters.graph_alternative, verbose=False)
        assert len(scenario.graphs) == 1
        subgraph = scenario.graphs[0].to(device)
        indices = scenario.data['edge_nodes_indices'][0].to(device)
        nodes_camera = scenario.data['nodes_camera'][0]
        feats = subgraph.ndata['h'].to(device)
        model = d['model']
        model.g = subgraph
        for layer in model.layers:
            layer.g = subgraph
        outputs = torch.squeeze(model(feats.float(), subgraph))
        indices = torch.squeeze(indices, 1).to('cpu')
        final_output = d['get_person_proposal_from_network_output'](outputs, subgraph, indices, nodes_camera,
                                                                    scenario.jsons_for_head, 0.5)
        # graph surface
        src, dst = subgraph.edges()
        assert np.array_equal(src.numpy(), arr[p + 'src']) and np.array_equal(dst.numpy(), arr[p + 'dst'])
        assert list(arr[p + 'nodes_camera']) == nodes_camera
        dense = torch.zeros_like(feats.cpu())
        rc = arr[p + 'feat_rc']
        dense[rc[:, 0], rc[:, 1]] = torch.from_numpy(arr[p + 'feat_v'])
        np.testing.assert_allclose(feats.cpu().numpy(), dense.numpy(), rtol=0, atol=5e-7)
        np.testing.assert_allclose(outputs.cpu().numpy(), arr[p + 'scores'], rtol=0, atol=2e-5)
        want = arr[p + 'persons']
        got = np.array([[(-1 if fo[c] is None else fo[c]) for c in parameters.used_cameras_skeleton_matching]
                        for fo in final_output], np.int32).reshape(-1, 5)
        assert np.array_equal(got, want)
        # 3D stage A
        batched_input = []
        for person in final_output:
            raw_input = {}
            for camera in parameters.used_cameras:
                if person[camera] is not None:
                    raw_input[camera] = [json.dumps([scenario.jsons_for_head[person[camera]]])]
            inputs = d['PoseEstimatorDataset'](raw_input, parameters.cameras, parameters.joint_list, save=False)
            assert len(inputs) == 1
            batched_input.append(inputs[0][0].reshape([1, inputs[0][0].size()[0]]).to(device))
        if batched_input:
            input_all = torch.cat(batched_input, dim=0)
            np.testing.assert_allclose(input_all.cpu().numpy(), arr[p + 'mlp_in'], rtol=0, atol=3e-7)
            output_all = d['mlp'](input_all.to(device))
            # same rule as test_mlp_error_budget_every_golden_row, on the mirror's own rows
            onp = oracle()
            ex = onp.mlp_exact(mlp_weights, input_all.cpu()).numpy()
            y_cpu = onp.mlp_forward(mlp_weights, input_all.cpu()).numpy()
            e_gpu, e_cpu = np.abs(output_all.cpu().numpy() - ex).max(), np.abs(y_cpu - ex).max()
            assert e_gpu <= e_cpu, (e_gpu, e_cpu)
            drift = np.abs(y_cpu - arr[p + 'mlp_out']).max()       # rows differ by <= 3e-7 from the reference's
            assert np.abs(output_all.cpu().numpy() - arr[p + 'mlp_out']).max() <= e_gpu + e_cpu + drift
        # 3D stage B
        has_id = any('ID' in sk for cam in input_element for sk in json.loads(input_element[cam][0]))
        if has_id:
            continue
        from pose_estimator_dataset_from_json import parameters as _p  # noqa: F401
        calibration = __import__('importlib').import_module('3d_multi_pose_estimator_amd.calibration')
        cal = calibration.Calibration(parameters)
        cam_matrix = {c: d['camera_matrix'](i).cpu().numpy() for i, c in enumerate(parameters.camera_names)}
        dist = {c: cal.dist[i] for i, c in enumerate(parameters.camera_names)}
        proj = {c: cal.P[i] for i, c in enumerate(parameters.camera_names)}
        for k, person in enumerate(final_output):
            points_2D = {}
            for cam_idx in parameters.cameras:
                camera = parameters.camera_names[cam_idx]
                if person[camera] is not None:
                    for j, pos in scenario.jsons_for_head[person[camera]].items():
                        points_2D.setdefault(j, {})[camera] = np.array([pos[1], pos[2]])
            result3D = d['triangulate'](points_2D, cam_matrix, dist, proj, parameters.axes_3D['Y'][0])
            for j in parameters.joint_list:
                assert (str(j) in result3D) == bool(arr[p + 'tri_valid'][k, j])
                if str(j) in result3D and j in parameters.used_joints:
                    np.testing.assert_allclose(result3D[str(j)].reshape(3), arr[p + 'tri'][k, j], rtol=1e-9, atol=1e-9)


def test_gat2_honours_caller_features(dropin, calib):
    """GAT2.forward(inputs, g) with inputs that are NOT the graph's own rows (dense path)."""
    d = dropin
    arr, frames = load_case('c2_3x2')
    frame = frames[0]
    pi = {c: [frame[c][0], 0] for c in frame if json.loads(frame[c][0])}
    scenario = d['MergedMultipleHumansDataset'](pi, mode='test', limit=10000, debug=True, alt='3', verbose=False)
    g = scenario.graphs[0]
    own = g.ndata['h']
    out_own = torch.squeeze(d['model'](own.clone().cuda(), g)).cpu()
    np.testing.assert_allclose(out_own.numpy(), arr['f0_scores'], rtol=0, atol=2e-5)
    scaled = own * 0.5
    out_scaled = torch.squeeze(d['model'](scaled.cuda(), g)).cpu()
    assert (out_scaled - out_own).abs().max() > 1e-4        # different input, different output
    # dense path on the graph's own rows must agree with the de-duplicated production path
    eng = d['model']._engine
    sc, sh = eng.gat_scores(g.device_batch(eng), heads=True, feats=own.cuda())
    np.testing.assert_allclose(torch.cat([sh, sc]).cpu().numpy(), arr['f0_scores'], rtol=0, atol=2e-5)


def test_gat2_without_final_activation_returns_logits(dropin):
    """final_activation=None (gat2.py:146-148 skips the activation): raw logits from the device,
    not logit(sigmoid(x)), which turns into +/-inf where the fp32 sigmoid saturates."""
    d = dropin
    arr, frames = load_case('c2_5x4_clean')
    frame = frames[0]
    pi = {c: [frame[c][0], 0] for c in frame if json.loads(frame[c][0])}
    scenario = d['MergedMultipleHumansDataset'](pi, mode='test', limit=10000, debug=True, alt='3', verbose=False)
    g = scenario.graphs[0]
    model = d['model']
    keep = model.final_activation
    try:
        model.final_activation = None
        logits = torch.squeeze(model(g.ndata['h'].cuda(), g)).cpu().double()
    finally:
        model.final_activation = keep
    assert torch.isfinite(logits).all() and logits.min() < 0 < logits.max()
    probs = torch.squeeze(model(g.ndata['h'].cuda(), g)).cpu().numpy()
    np.testing.assert_allclose(probs, arr['f0_scores'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(torch.sigmoid(logits).numpy(), arr['f0_scores'], rtol=0, atol=2e-5)


def test_rows_prefetched_per_frame_equal_the_per_person_rows(dropin, monkeypatch):
    """The 3D stage of the reference's loop builds one PoseEstimatorDataset per PERSON (metrics_from_model.py:243-277).  The
    mirrors compute all rows of a frame in ONE mpe_mlp_input_rows launch when the persons are formed (runtime.prefetch_mlp_rows)
    and serve them by the JSON text the caller hands over; a row served from there is bit-identical to the row the dataset
    computes itself (MPE_DROPIN_PREFETCH=0), and the invalid-row rule (sum |x| > 1, :287) is the same."""
    d = dropin
    parameters = d['parameters']
    runtime = __import__('importlib').import_module('3d_multi_pose_estimator_amd.runtime')
    arr, frames = load_case('c2_5x4_messy')
    checked = 0
    for input_element in frames:
        processed_input = {}
        for cam in input_element:
            data = json.loads(input_element[cam][0])
            if data:
                processed_input[cam] = [json.dumps(data), input_element[cam][1]]
        scenario = d['MergedMultipleHumansDataset'](processed_input, mode='test', limit=10000, debug=True,
                                                    alt=parameters.graph_alternative, verbose=False)
        if not scenario.graphs:
            continue
        subgraph = scenario.graphs[0]
        indices = torch.squeeze(scenario.data['edge_nodes_indices'][0], 1)
        outputs = torch.squeeze(d['model'](None, subgraph))
        runtime._row_cache.clear()
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
        final_output = d['get_person_proposal_from_network_output'](outputs, subgraph, indices, scenario.data['nodes_camera'][0],
                                                                    scenario.jsons_for_head, 0.5)
        assert len(runtime._row_cache) == len(final_output)
        for person in final_output:
            raw_input = {}
            for camera in parameters.used_cameras:
                if person[camera] is not None:
                    raw_input[camera] = [json.dumps([scenario.jsons_for_head[person[camera]]])]
            monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
            n_before = len(runtime._row_cache)
            fast = d['PoseEstimatorDataset'](raw_input, parameters.cameras, parameters.joint_list, save=False)
            assert len(runtime._row_cache) == n_before
            monkeypatch.setenv('MPE_DROPIN_PREFETCH', '0')
            slow = d['PoseEstimatorDataset'](raw_input, parameters.cameras, parameters.joint_list, save=False)
            assert len(fast) == len(slow) == 1
            assert torch.equal(fast[0][0], slow[0][0])
            checked += 1
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
    assert checked >= 4


def test_dropin_loop_region_gives_the_engine_bits(gat_weights, mlp_weights, calib):
    """bench.py's `dropin_loop` region (harness/dropin_loop.py: the mirrors called once per frame in the order of the reference's
    loop, test/metrics_from_model.py:178-294): it reports both of the reference's timers, and the poses it ends with are, bit for
    bit, those of the batched engine on the same frame (same kernels: the per-frame row prefetch, the wave-per-tile GEMMs at
    small M)."""
    import importlib
    loop = importlib.import_module('3d_multi_pose_estimator_amd.harness.dropin_loop')
    syn = importlib.import_module('3d_multi_pose_estimator_amd.synthetic')
    pipeline = importlib.import_module('3d_multi_pose_estimator_amd.pipeline')
    par = importlib.import_module('3d_multi_pose_estimator_amd.parameters').parameters
    sd, prm = gat_weights
    model, mlp = loop.build_models(sd, prm, mlp_weights)
    frames = [syn.make_frame(calib, 40 + i, syn.FrameSpec(persons=3 + i % 2, noise_px=1.0))[0] for i in range(8)]
    out = loop.run(frames, model, mlp, warmup=2)
    assert out['frames'] == 6 and out['graph_matching_ms'] > 0 and out['pose_3d_ms'] > 0 and out['inside_mirrors_ms'] < out['ms_per_frame']
    eng = pipeline.Engine(par, calib, max_frames=2, max_persons_per_camera=5)
    eng.load_gat(sd, prm)
    eng.load_mlp(mlp_weights)
    onp = oracle()
    db = eng.to_device(eng.pack([onp.processed_input(frames[-1])]))
    _, persons, n_persons = eng.match(db, want_scores=False)
    poses, valid = eng.mlp3d(db, persons, n_persons)
    n = int(n_persons[0])
    assert n == len(out['last']) >= 3
    want = poses[0, :n].cpu().numpy()
    got = np.array([[j for j in person] for person in out['last']], np.float32)
    assert np.array_equal(got, want)
    eng.close()


def test_frame_loop_calls_the_mirrors_like_the_reference_script(calib):
    """The one-frame-per-call loop of the package (harness/dropin_loop.py, bench.py's `dropin_loop` region) is written in its
    own form; WHAT it has to share with the reference's loop is pinned here: on the committed pinning file it calls the mirrors
    in the order, and with arguments of the shapes, that test/metrics_from_model.py calls the reference's symbols with --
    recorded by oracle/gen_dropin_trace.py, which runs that script unchanged with tracing wrappers around
    MergedMultipleHumansDataset, GAT2.forward, get_person_proposal_from_network_output, PoseEstimatorDataset and
    PoseEstimatorMLP.forward (tests/golden/dropin/call_trace.json).  Frame selection (stride, frames without ground-truth bodies)
    is the script's own and is repeated here from its documented rules (:124, :137-138)."""
    import importlib
    loop = importlib.import_module('3d_multi_pose_estimator_amd.harness.dropin_loop')
    syn = importlib.import_module('3d_multi_pose_estimator_amd.synthetic')
    par = importlib.import_module('3d_multi_pose_estimator_amd.parameters').parameters
    want = json.load(open(os.path.join(GOLDEN, 'dropin', 'call_trace.json')))
    meta = json.load(open(os.path.join(GOLDEN, 'harness', 'harness_expected.json')))['inputs']
    frames = json.load(open(os.path.join(GOLDEN, 'harness', want['testfile'])))
    picked = []
    for i, frame in enumerate(frames):
        if i % want['datastep']:
            continue
        bodies = max((frame[c][3] for c in frame), key=len)
        if len(bodies):
            picked.append(frame)
    V, J = len(par.camera_names), len(par.joint_list)
    nf = 2 + V * J * 10
    gat = syn.matcher_gat_state_dict(nf, V, J, noise_seed=meta['gat']['noise_seed'], noise_bound=meta['gat']['noise_bound'])
    mlp = syn.decoder_mlp_state_dict(V, J, 14, noise_seed=meta['mlp']['noise_seed'], noise_bound=meta['mlp']['noise_bound'])
    matcher, lifter = loop.build_models(gat, syn.gat_params(nf), mlp)
    got = []
    out = loop.run(picked, matcher, lifter, warmup=0, trace=got)
    assert len(got) == len(want['frames']) == len(picked)
    for n, (g, w) in enumerate(zip(got, want['frames'])):
        assert [e[0] for e in g] == [e[0] for e in w], (n, [e[0] for e in g], [e[0] for e in w])
        for eg, ew in zip(g, w):
            assert eg == ew, (n, eg, ew)
    assert out['frames'] == sum(1 for w in want['frames'] if len(w) > 1)       # frames without a graph end after the first call


def test_gat2_forward_raises_for_a_frame_beyond_capacity_without_a_device_round_trip(dropin, calib, monkeypatch):
    """GAT2.forward skips the device status word for a single implicit-topology graph (0.2 ms per frame): the host check
    (FrameGraph.device_batch -> Engine.check_capacity) must then cover everything k_topology can flag for such a frame, i.e. more
    skeletons than max_heads_per_frame (gat.hip:topology_body raises status bit 0 for exactly H > hmax).  A frame of 15 skeletons on a
    mirror built for 10 raises from forward() itself, before anything is launched, and leaves no stale status bit behind."""
    d = dropin
    import importlib
    syn = importlib.import_module('3d_multi_pose_estimator_amd.synthetic')
    monkeypatch.setenv('MPE_MAX_PERSONS_PER_CAMERA', '2')               # a fresh mirror: capacity 5 cameras x 2 = 10 skeletons per frame
    prm = d['prm']
    model = d['GAT'](None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'], torch.nn.LeakyReLU(),
                     torch.nn.Sigmoid(), prm['in_drop'], prm['attn_drop'], prm['alpha'], prm['residual'], bias=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in d['sd'].items()})
    def graph(persons, index):
        frame = syn.make_frame(calib, index, syn.FrameSpec(persons=persons))[0]
        pi = {c: [frame[c][0], 0] for c in frame if json.loads(frame[c][0])}
        return d['MergedMultipleHumansDataset'](pi, mode='test', limit=10000, debug=True, alt='3', verbose=False).graphs[0]
    ok = graph(2, 700)
    out = model(ok.ndata['h'].cuda(), ok)
    assert out.shape[0] == ok.num_nodes()
    big = graph(3, 701)                                                   # 15 skeletons
    with pytest.raises(ValueError, match='capacity'):
        model(None, big)
    model._engine.sync_status()                                           # nothing was launched for it: the status word is clean
    again = model(ok.ndata['h'].cuda(), ok)
    assert torch.equal(again, out)
    model._engine.close()


def test_mirrors_take_f64_sums_in_the_matching_network_on_request(dropin, calib, gat_weights, monkeypatch):
    """MPE_GAT_ACC64=1 (read when a mirror builds its engine): GAT2.forward's scores are those of an Engine with
    set_precision(gat_acc64=True) -- f64 running sums in every GAT GEMM -- bit for bit, and not the default precision's."""
    d = dropin
    import importlib
    syn = importlib.import_module('3d_multi_pose_estimator_amd.synthetic')
    pipeline = importlib.import_module('3d_multi_pose_estimator_amd.pipeline')
    prm = d['prm']
    frame = syn.make_frame(calib, 905, syn.FrameSpec(persons=4))[0]
    pi = {c: [frame[c][0], 0] for c in frame if json.loads(frame[c][0])}

    def mirror_scores():
        model = d['GAT'](None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'], torch.nn.LeakyReLU(),
                         torch.nn.Sigmoid(), prm['in_drop'], prm['attn_drop'], prm['alpha'], prm['residual'], bias=True)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in d['sd'].items()})
        g = d['MergedMultipleHumansDataset'](pi, mode='test', limit=10000, debug=True, alt='3', verbose=False).graphs[0]
        out = model(g.ndata['h'].cuda(), g).reshape(-1).cpu().numpy().copy()
        model._engine.close()
        return out

    plain = mirror_scores()
    monkeypatch.setenv('MPE_GAT_ACC64', '1')
    acc = mirror_scores()
    eng = pipeline.Engine(calib.params, calib, max_frames=1, max_persons_per_camera=10)
    try:
        sd, p2 = gat_weights
        eng.load_gat(sd, p2)
        eng.set_precision(gat_acc64=True)
        from conftest import oracle
        db = eng.to_device(eng.pack([oracle().processed_input(frame)]))
        sc, sh = eng.gat_scores(db, heads=True)
        want = np.concatenate([sh.cpu().numpy(), sc.cpu().numpy()])
    finally:
        eng.close()
    assert acc.shape == want.shape and np.array_equal(acc, want)
    assert plain.shape == acc.shape and not np.array_equal(plain, acc) and np.abs(plain - acc).max() < 1e-4


@pytest.mark.parametrize('mode', ['max_accuracy', 'f64'])
def test_mlp_mirror_takes_its_precision_mode_on_request(dropin, calib, mlp_weights, monkeypatch, mode):
    """MPE_MLP_PRECISION=max_accuracy | f64 (read when the PoseEstimatorMLP mirror builds its engine): forward() gives the rows of an
    Engine with set_precision(mlp_max_accuracy=True) / (mlp_f64=True), bit for bit."""
    import importlib
    pipeline = importlib.import_module('3d_multi_pose_estimator_amd.pipeline')
    d = dropin
    in_dim = int(np.asarray(mlp_weights['layers.1.weight']).shape[1])          # the reference's module tree: Flatten, then Linear at 1, 3, ..., 17
    x = torch.from_numpy(np.random.RandomState(5).uniform(-1.0, 1.0, (7, in_dim)).astype(np.float32))

    def mirror():
        m = d['PoseEstimatorMLP'](in_dim, 54)
        m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in mlp_weights.items()})
        y = m(x.cuda()).cpu().numpy().copy()
        m._engine.close()
        return y

    plain = mirror()
    monkeypatch.setenv('MPE_MLP_PRECISION', mode)
    got = mirror()
    eng = pipeline.Engine(calib.params, calib, max_frames=4, max_persons_per_camera=10)
    try:
        eng.load_mlp(mlp_weights)
        eng.set_precision(mlp_max_accuracy=mode == 'max_accuracy', mlp_f64=mode == 'f64')
        want = eng.mlp_forward(x.cuda()).cpu().numpy()
    finally:
        eng.close()
    assert np.array_equal(got, want)
    assert plain.shape == got.shape and np.abs(plain - got).max() <= 1e-3 * max(1.0, np.abs(plain).max())
    monkeypatch.setenv('MPE_MLP_PRECISION', 'fast')
    with pytest.raises(ValueError, match='MPE_MLP_PRECISION'):
        mirror()


def _frame_inputs(d, names=CASES):
    frames = [f for name in names for f in load_case(name)[1]]
    out = []
    for input_element in frames:
        processed = {}
        for cam in input_element:
            data = json.loads(input_element[cam][0])
            if data:
                processed[cam] = [json.dumps(data), input_element[cam][1]]
        out.append(processed)
    return out


def _one_frame(d, processed, model=None, threshold=0.5, touch=None, indices=None):
    """One pass of the per-frame loop up to the persons' MLP rows -> (scores, proposals, rows, scenario)."""
    parameters = d['parameters']
    scenario = d['MergedMultipleHumansDataset'](processed, mode='test', limit=10000, debug=True, alt=parameters.graph_alternative, verbose=False)
    if not scenario.graphs:
        return None
    g = scenario.graphs[0]
    outputs = torch.squeeze((model or d['model'])(g.ndata['h'].float(), g))
    if touch is not None:
        outputs = touch(outputs)
    idx = torch.squeeze(scenario.data['edge_nodes_indices'][0], 1) if indices is None else indices(scenario)
    persons = d['get_person_proposal_from_network_output'](outputs, g, idx, scenario.data['nodes_camera'][0], scenario.jsons_for_head, threshold)
    rows = []
    for person in persons:
        views = {cam: [json.dumps([scenario.jsons_for_head[person[cam]]])] for cam in parameters.used_cameras if person[cam] is not None}
        ds = d['PoseEstimatorDataset'](views, parameters.cameras, parameters.joint_list, save=False)
        rows.append(ds[0][0].clone() if len(ds) else None)
    return outputs.detach().cpu().clone(), persons, rows, scenario


def _same(a, b):
    assert torch.equal(a[0], b[0]) and a[1] == b[1] and len(a[2]) == len(b[2])
    for x, y in zip(a[2], b[2]):
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y))


def test_queued_frame_results_are_the_step_by_step_results(dropin, monkeypatch):
    """The per-frame mirrors queue a frame's scores when its dataset is built (on the engine of the matcher that scored the previous
    frame) and its clustering + MLP rows right behind them (runtime.start_frame / queue_proposals): scores, proposals and rows must
    be the bits of the step-by-step route (MPE_DROPIN_PREFETCH=0), frame after frame, and the queued results must really be the ones
    handed out."""
    d = dropin
    runtime = __import__('importlib').import_module('3d_multi_pose_estimator_amd.runtime')
    inputs = _frame_inputs(d)
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '0')
    want = [_one_frame(d, p) for p in inputs]
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
    _one_frame(d, inputs[0])                               # the matcher's engine becomes the one frames are queued on
    used = 0
    for p, w in zip(inputs, want):
        got = _one_frame(d, p)
        assert (got is None) == (w is None)
        if got is None:
            continue
        _same(got, w)
        g = got[3].graphs[0]
        fs = g.__dict__.get('_scored')
        assert fs is not None and fs.ahead is g._ahead and fs.out.data_ptr() == d['model'](None, g).data_ptr()      # (a second call hands the same scores out)
        used += 1
    assert used >= 4


def test_queued_frame_results_are_dropped_when_the_caller_departs_from_the_script(dropin, gat_weights, monkeypatch):
    """Queued results are used only for exactly what they were computed from: another matcher (other weights, or no final activation),
    scores the caller changed, another threshold, other indices -- each must give what the step-by-step route gives for THAT call."""
    d = dropin
    parameters = d['parameters']
    inputs = [p for p in _frame_inputs(d) if p][:3]
    sd, prm = gat_weights
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
    _one_frame(d, inputs[0])

    def both(**kw):
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
        _one_frame(d, inputs[0])                           # (re-arms the hint with the fixture's matcher)
        fast = _one_frame(d, inputs[1], **kw)
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', '0')
        slow = _one_frame(d, inputs[1], **kw)
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
        _same(fast, slow)
        return fast
    base = both()
    # another matcher: same architecture, different weights, and one without the final activation
    GAT = d['GAT']
    other = GAT(None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'], torch.nn.LeakyReLU(), torch.nn.Sigmoid(),
                prm['in_drop'], prm['attn_drop'], prm['alpha'], prm['residual'], bias=True)
    other.load_state_dict({k: torch.from_numpy(v * (0.5 if k.endswith('fc2.weight') else 1.0)) for k, v in sd.items()})
    o = both(model=other)
    assert not torch.equal(o[0], base[0])
    raw = GAT(None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'], torch.nn.LeakyReLU(), None,
              prm['in_drop'], prm['attn_drop'], prm['alpha'], prm['residual'], bias=True)
    raw.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    r = both(model=raw, threshold=0.0)
    assert float(r[0].min()) < 0.0 or float(r[0].max()) > 1.0           # logits, not probabilities
    # weights changed in place between two frames: the matcher builds a new engine, the frame queued on the old one is not used
    with torch.no_grad():
        d['model'].layers[1].fc2.weight.mul_(1.0)
    both()
    # ... and proposals asked for scores of the OLD engine after the matcher has replaced it: the step-by-step route, no dead context
    sc_old = d['MergedMultipleHumansDataset'](inputs[1], mode='test', limit=10000, debug=True, alt=parameters.graph_alternative, verbose=False)
    g_old = sc_old.graphs[0]
    out_old = torch.squeeze(d['model'](None, g_old))
    with torch.no_grad():
        d['model'].layers[1].fc2.weight.mul_(1.0)
    d['model'](None, d['MergedMultipleHumansDataset'](inputs[2], mode='test', limit=10000, debug=True, alt=parameters.graph_alternative, verbose=False).graphs[0])
    assert g_old._ahead is not None and not g_old._ahead.engine.ctx
    got_old = d['get_person_proposal_from_network_output'](out_old, g_old, torch.squeeze(sc_old.data['edge_nodes_indices'][0], 1), sc_old.data['nodes_camera'][0],
                                                           sc_old.jsons_for_head, 0.5)
    assert got_old == base[1]
    # scores changed by the caller (in place, and as a new tensor), another threshold
    both(touch=lambda s: s.mul_(0.5))
    both(touch=lambda s: s * 0.5)
    both(threshold=0.9)
    both(threshold=0.999999)
    # indices that are not the graph's own edge-node ids: refused on either route
    for prefetch in ('1', '0'):
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', prefetch)
        with pytest.raises(ValueError):
            _one_frame(d, inputs[1], indices=lambda sc: torch.squeeze(sc.data['edge_nodes_indices'][0], 1)[:-1])
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
    # more frames queued than the ring holds before anybody asks for their proposals: the overwritten ones take the step-by-step route
    scen = [d['MergedMultipleHumansDataset'](p, mode='test', limit=10000, debug=True, alt=parameters.graph_alternative, verbose=False) for p in (inputs * 3)[:7]]
    for sc_, p in zip(scen, (inputs * 3)[:7]):
        g = sc_.graphs[0]
        outputs = torch.squeeze(d['model'](None, g))
        got = d['get_person_proposal_from_network_output'](outputs, g, torch.squeeze(sc_.data['edge_nodes_indices'][0], 1), sc_.data['nodes_camera'][0],
                                                           sc_.jsons_for_head, 0.5)
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', '0')
        want = _one_frame(d, p)
        monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
        assert got == want[1] and torch.equal(outputs.cpu(), want[0])


def test_queued_frame_results_on_random_frames(dropin, calib, monkeypatch):
    """The same comparison (queued against step by step) on 40 random frames: 1 ... 6 persons, dropped joints, an empty camera now and then."""
    import importlib
    syn = importlib.import_module('3d_multi_pose_estimator_amd.synthetic')
    loop = importlib.import_module('3d_multi_pose_estimator_amd.harness.dropin_loop')
    d = dropin
    cams = list(d['parameters'].used_cameras_skeleton_matching)
    inputs = []
    for i in range(40):
        spec = syn.FrameSpec(persons=1 + i % 6, empty_cameras=(cams[i % len(cams)],) if i % 5 == 2 else (), joint_drop=(0.0, 0.2)[i % 2], noise_px=1.0)
        inputs.append(loop.cameras_with_skeletons(syn.make_frame(calib, 9000 + i, spec)[0]))
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '0')
    want = [_one_frame(d, p) for p in inputs]
    monkeypatch.setenv('MPE_DROPIN_PREFETCH', '1')
    _one_frame(d, inputs[0])
    persons = 0
    for p, w in zip(inputs, want):
        got = _one_frame(d, p)
        assert (got is None) == (w is None)
        if got is not None:
            _same(got, w)
            assert got[3].graphs[0].__dict__.get('_scored') is not None
            persons += len(got[1])
    assert persons > 60

"""CPU: the oracle restatement (oracle/oracle_np.py) against the fixtures produced by the
reference's own files (oracle/gen_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ALL_CASES, ALL_CASES_FZ, CASES, GOLDEN, ROOT, env, golden_dir, load_case, oracle, pkg


@pytest.mark.parametrize('variant', ['panoptic', 'arplab', 'arprobot', 'ring23'])
def test_calibration_matches_reference_globals(variant):
    import os
    e = env(variant)
    calib = e.calib
    g = np.load(os.path.join(golden_dir(variant), 'calibration_%s.npz' % variant))
    # the graph generator's globals cover used_cameras_skeleton_matching in camera_names order
    # (graph_generator.py:38-52), the 3D stage's cover every configured camera (dataset :28-47)
    sm = [calib.index(c) for c in e.params.camera_names if c in e.params.used_cameras_skeleton_matching]
    assert np.array_equal(g['T_d'].astype(np.float64), calib.T_d[sm].astype(np.float32).astype(np.float64))
    assert np.array_equal(g['T_i32'], calib.T_i32[sm])
    assert np.array_equal(g['K32'], calib.K32[sm])
    assert np.array_equal(g['Kinv32'], calib.Kinv32[sm])
    assert np.array_equal(g['centre32'], calib.centre32[sm])
    assert np.array_equal(g['dist'], calib.dist)
    assert np.array_equal(g['P'], calib.P)
    assert len(g['features']) == 2 + len(sm) * 18 * 10 == e.meta['num_feats']


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_graph_and_gat(variant, name):
    onp = oracle()
    calib = env(variant).calib
    arr, frames = load_case(name, variant)
    sd, prm = env(variant).gat
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        g = onp.build_graph(onp.processed_input(frame), calib)
        assert g['N'] == int(arr[p + 'N'])
        assert np.array_equal(g['src'], arr[p + 'src'])
        assert np.array_equal(g['dst'], arr[p + 'dst'])
        assert np.array_equal(g['edge_nodes_indices'], arr[p + 'edge_nodes_indices'])
        assert list(arr[p + 'nodes_camera']) == g['nodes_camera']
        assert [g['skeleton_index'][i] for i in range(g['H'])] == list(arr[p + 'skeleton_index'])
        dense = torch.zeros_like(g['feats'])
        rc = arr[p + 'feat_rc']
        dense[rc[:, 0], rc[:, 1]] = torch.from_numpy(arr[p + 'feat_v'])
        # feature rows: bit-exact except the ray columns (3-term fp32 dot products through
        # MKL, whose rounding depends on operand alignment): those within 1 ulp
        ray = ((torch.arange(dense.shape[1]) - 2) % 10 >= 7) & (torch.arange(dense.shape[1]) >= 2)
        assert torch.equal(dense[:, ~ray], g['feats'][:, ~ray])
        np.testing.assert_allclose(g['feats'][:, ray].numpy(), dense[:, ray].numpy(), rtol=0, atol=1.2e-7)
        scores, inter = onp.gat_forward(sd, prm, g['feats'], g['src'], g['dst'], keep=True)
        # same torch CPU kernels on the same machine -> equal to a few ulp at most on the hand-made cases.  On the random-shape frames
        # (round 6) the two fp32 evaluations -- the reference's own modules over the DGL stand-in, and the oracle's restatement --
        # differ by up to 1.17e-5 in a score with 8 host threads (ARPLAB frame 5, at a score of 0.369) and by up to 1.65e-5 with other thread
        # counts, on PANOPTIC too (profiles/r06_reference_score_noise_by_threads.txt): MKL's blocking of the sums, amplified by
        # the fixture weights' logit gain of 25.  That is the size of the noise the 2e-5 bound of the GPU tests is about.
        # The hand-made cases keep the tight check where the host runs torch with the thread count the fixtures were written under
        # (meta.json: torch_threads; with 4 threads instead of 8, arp_6x3 frame 1 is 9.2e-6 away) and take the absolute bound elsewhere.
        same_host = torch.get_num_threads() == env(variant).meta.get('torch_threads')
        if name.startswith('fz_') or not same_host:
            np.testing.assert_allclose(scores.numpy(), arr[p + 'scores'], rtol=0, atol=2e-5)
        else:
            np.testing.assert_allclose(scores.numpy(), arr[p + 'scores'], rtol=2e-5, atol=1e-7)
        H = g['H']
        for l, a in enumerate(inter):
            np.testing.assert_allclose(a[:4].numpy(), arr[p + 'act%d_head' % l], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(a[H:H + 4].numpy(), arr[p + 'act%d_en' % l], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_cluster_on_golden_scores(variant, name):
    onp = oracle()
    calib = env(variant).calib
    arr, frames = load_case(name, variant)
    sm = list(calib.params.used_cameras_skeleton_matching)
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        g = onp.build_graph(onp.processed_input(frame), calib)
        head_cam = [sm.index(c) for c in g['nodes_camera'][:g['H']]]
        persons = onp.cluster(arr[p + 'scores'][g['H']:], g['pairs'], g['H'], head_cam, len(sm))
        assert np.array_equal(np.array(persons, np.int32).reshape(-1, len(sm)), arr[p + 'persons'])


@pytest.mark.parametrize('variant', ['panoptic', 'ring23'])
def test_cluster_known_answers(variant):
    import os
    onp = oracle()
    calib = env(variant).calib
    arr = np.load(os.path.join(golden_dir(variant), 'cluster_cases.npz'))
    sm = list(calib.params.used_cameras_skeleton_matching)
    nonempty = 0
    for i in range(int(arr['n'])):
        slot_cam, slot_n = arr['c%d_slot_cam' % i], arr['c%d_slot_n' % i]
        slots, hid = [], 0
        for c, k in zip(slot_cam, slot_n):
            slots.append((sm[c], list(range(hid, hid + k))))
            hid += k
        N, src, dst, pairs = onp.topology(slots)
        head_cam = [int(c) for c, k in zip(slot_cam, slot_n) for _ in range(k)]
        persons = onp.cluster(arr['c%d_scores' % i], pairs, hid, head_cam, len(sm))
        want = arr['c%d_persons' % i]
        assert np.array_equal(np.array(persons, np.int32).reshape(-1, len(sm)), want), i
        nonempty += len(want) > 0
    assert nonempty > (300 if variant == 'panoptic' else 100)


def test_cluster_fresh_known_answers():
    """1500 further known answers of the reference's get_person_proposal_from_network_output (skeleton_matching_utils.py:12-132; the real
    function through oracle/refenv.py, oracle/gen_cluster_fuzz.py): near-threshold scores in float32 steps, confidently wrong links
    (components with repeated cameras), up to ten skeletons per camera -- the oracle's restatement gives the reference's persons on
    every one."""
    import os
    onp = oracle()
    calib = env('panoptic').calib
    arr = np.load(os.path.join(golden_dir('panoptic'), 'cluster_cases_fuzz.npz'))
    sm = list(calib.params.used_cameras_skeleton_matching)
    nonempty = multi = 0
    for i in range(int(arr['n'])):
        slot_cam, slot_n = arr['c%d_slot_cam' % i], arr['c%d_slot_n' % i]
        slots, hid = [], 0
        for c, k in zip(slot_cam, slot_n):
            slots.append((sm[c], list(range(hid, hid + k))))
            hid += k
        N, src, dst, pairs = onp.topology(slots)
        head_cam = [int(c) for c, k in zip(slot_cam, slot_n) for _ in range(k)]
        persons = onp.cluster(arr['c%d_scores' % i], pairs, hid, head_cam, len(sm))
        want = arr['c%d_persons' % i]
        assert np.array_equal(np.array(persons, np.int32).reshape(-1, len(sm)), want), i
        nonempty += len(want) > 0
        multi += len(want) > 3
    assert int(arr['n']) == 1500 and nonempty > 1200 and multi > 300


@pytest.mark.parametrize('variant,name', ALL_CASES_FZ)
def test_stage3d(variant, name):
    onp = oracle()
    calib = env(variant).calib
    mlp_weights = env(variant).mlp
    arr, frames = load_case(name, variant)
    sm = list(calib.params.used_cameras_skeleton_matching)
    for n, frame in enumerate(frames):
        p = 'f%d_' % n
        persons = arr[p + 'persons']
        if len(persons) == 0:
            continue
        g = onp.build_graph(onp.processed_input(frame), calib)
        rows = []
        for person in persons:
            sk = onp.person_skeletons(list(person), g['jsons_for_head'], sm)
            row, kept = onp.mlp_input_row(sk, calib)
            assert kept
            rows.append(row)
            tri = onp.triangulate_person({c: {k: v for k, v in s.items() if k != 'ID'} for c, s in sk.items()}, calib)
            k = len(rows) - 1
            for j in range(18):
                assert (j in tri) == bool(arr[p + 'tri_valid'][k, j])
                if j in tri:
                    np.testing.assert_allclose(tri[j], arr[p + 'tri'][k, j], rtol=0, atol=1e-9)
        x = torch.stack(rows)
        # f64 DLT through our SVD restatement vs numpy's in the shim: same LAPACK -> tiny diff
        np.testing.assert_allclose(x.numpy(), arr[p + 'mlp_in'], rtol=0, atol=2e-7)
        out = onp.mlp_forward(mlp_weights, torch.from_numpy(arr[p + 'mlp_in']))
        np.testing.assert_allclose(out.numpy(), arr[p + 'mlp_out'], rtol=1e-5, atol=1e-6)
        poses = np.stack([onp.decode_pose(out[i], 18) for i in range(out.shape[0])])
        np.testing.assert_allclose(poses, arr[p + 'poses'], rtol=1e-5, atol=1e-5)
        # the capture-volume MLP of the fixtures (same reference module, second set of weights)
        out_room = onp.mlp_forward(env(variant).mlp_room, torch.from_numpy(arr[p + 'mlp_in']))
        np.testing.assert_allclose(out_room.numpy(), arr[p + 'mlp_out_room'], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(arr[p + 'poses_room'].reshape(len(persons), -1), arr[p + 'mlp_out_room'] * np.float32(10.0))


@pytest.mark.parametrize('mode,key', [('mlp', 'model'), ('tri', 'triangulation')])
def test_harness_bookkeeping_reproduces_reference_report(mode, key, tmp_path):
    """CPU half of the a14/a15 pin: the harness' own file handling (--datastep stride over the
    file, tm_<a>_<b>.pickle lookup, GT to world through the dataset calibration, skip rules) and
    its Metrics bookkeeping, with the ORACLE as the inference side, must print what the
    reference's scripts printed (tests/golden/harness/harness_expected.json)."""
    import argparse
    import json
    import pickle
    import torch
    from conftest import GOLDEN, harness_model_files
    onp = oracle()
    common = pkg('harness.common')
    calib = env().calib
    hd = os.path.join(GOLDEN, 'harness')
    exp = json.load(open(os.path.join(hd, 'harness_expected.json')))
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    prm = pickle.load(open(os.path.join(mdir, 'skeleton_matching.prms'), 'rb'))
    prm = dict(prm, nonlinearity=prm['nonlinearity'].negative_slope)
    gat_sd = {k: v.numpy() for k, v in torch.load(os.path.join(mdir, 'skeleton_matching.tch')).items()}
    mlp_sd = {k: v.numpy() for k, v in torch.load(os.path.join(mdir, 'pose_estimator.pytorch'))['model_state_dict'].items()}
    args = argparse.Namespace(synthetic=0, tmdir=[hd], testfiles=[os.path.join(hd, exp['inputs']['testfile'])],
                              datastep=exp['inputs']['datastep'])
    work = common.collect_work(args, calib)
    assert len(work) == 16
    J = len(calib.params.joint_list)

    def infer(frames, owners):
        out = []
        for frame in frames:
            res = onp.run_frame(frame, calib, gat_sd, prm, mlp_sd, mode=mode)
            if res is None:
                out.append(None)
            elif mode == 'mlp':
                out.append([{j: p[j] for j in range(J)} for p in res['poses']])
            else:
                out.append(res['tri'])
        return out
    metrics, n_data, _ = common.evaluate(work, infer, mode, torch.from_numpy(calib.T_i32[1]), batch=5)
    got = metrics.report()
    want = exp[key]
    assert n_data == 14                       # 16 strided frames - one without GT bodies - one without a graph
    assert abs(got['mpjpe_mm'] - want['mpjpe_mm']) < 1e-6
    for th, triple in want['ap'].items():
        assert got['ap'][th] == pytest.approx(triple, rel=1e-12, abs=1e-12), th


def _harness_inputs(tmp_path):
    import json
    import pickle
    import torch
    from conftest import GOLDEN, harness_model_files
    hd = os.path.join(GOLDEN, 'harness')
    exp = json.load(open(os.path.join(hd, 'harness_expected.json')))
    mdir = harness_model_files(str(tmp_path), exp['inputs'])
    prm = pickle.load(open(os.path.join(mdir, 'skeleton_matching.prms'), 'rb'))
    prm = dict(prm, nonlinearity=prm['nonlinearity'].negative_slope)
    gat_sd = {k: v.numpy() for k, v in torch.load(os.path.join(mdir, 'skeleton_matching.tch')).items()}
    mlp_sd = {k: v.numpy() for k, v in torch.load(os.path.join(mdir, 'pose_estimator.pytorch'))['model_state_dict'].items()}
    return hd, exp, prm, gat_sd, mlp_sd


def test_sm_metrics_bookkeeping_reproduces_reference_report(tmp_path):
    """f2: harness/sm_metrics.py (GT grouping from the 3D bodies, label lists, sklearn metrics)
    with the oracle as the inference side prints what /root/reference/test/sm_metrics.py printed."""
    import argparse
    onp = oracle()
    sm = pkg('harness.sm_metrics')
    calib = env().calib
    hd, exp, prm, gat_sd, _ = _harness_inputs(tmp_path)
    args = argparse.Namespace(synthetic=0, testfiles=[os.path.join(hd, exp['inputs']['testfile'])], datastep=exp['inputs']['datastep'])
    work = sm.collect_work(args, calib)

    def infer(frames, owners):
        out = []
        for frame in frames:
            res = onp.run_frame(frame, calib, gat_sd, prm, None, mode='none')
            out.append(None if res is None else (res['graph']['H'], [[h for h in p if h >= 0] for p in res['persons']]))
        return out
    got = sm.evaluate(work, infer, batch=6)
    for k, v in exp['sm_metrics'].items():
        assert got[k] == pytest.approx(v, rel=1e-12, abs=1e-12), k


def test_reprojection_bookkeeping_reproduces_reference_report(tmp_path):
    """f3: harness/reprojection_error.py with the oracle as the inference side against the numbers
    /root/reference/test/reprojection_error.py printed: the per-camera MEDIANS (the means are
    dominated by a few points that project through z ~ 0 and come out as 1e6..1e10 px; they are
    compared on the log scale only)."""
    import argparse
    onp = oracle()
    rp = pkg('harness.reprojection_error')
    calib = env().calib
    hd, exp, prm, gat_sd, mlp_sd = _harness_inputs(tmp_path)
    args = argparse.Namespace(synthetic=0, testfiles=[os.path.join(hd, exp['inputs']['testfile'])], datastep=exp['inputs']['datastep'])
    work = rp.collect_work(args, calib)
    sm = list(calib.params.used_cameras_skeleton_matching)

    def infer(frames, owners):
        out = []
        for frame in frames:
            res = onp.run_frame(frame, calib, gat_sd, prm, mlp_sd, mode='mlp')
            people = []
            if res is not None:
                k = 0
                for p in res['persons']:
                    skels = onp.person_skeletons(p, res['graph']['jsons_for_head'], sm)
                    row, kept = onp.mlp_input_row(skels, calib)
                    est = None
                    if kept:
                        est = res['poses'][k]
                        k += 1
                    tri = onp.triangulate_person(skels, calib, positive_ids_only=True, all_joints=True)
                    people.append((skels, est, {j: v.astype(np.float32) for j, v in tri.items()}))
            out.append(people)
        return out
    got = rp.evaluate(work, infer, calib, batch=6)
    for cam, kinds in exp['reprojection_error'].items():
        for kind, (mean, median) in kinds.items():
            g = got[(kind, cam)]
            assert g[1] == pytest.approx(median, rel=2e-4), (cam, kind, g, median)
            assert abs(np.log10(g[0]) - np.log10(mean)) < 0.5, (cam, kind, g, mean)


def test_generated_scenes_oracle_vs_reference_script():
    """the oracle's restatement of mode='test_generated' (sampling graph_generator.py:526-536, 674-693;
    graph synthesis :697-810) and of the loop of test/sm_metrics_without_gt.py:112-170 against what the REFERENCE SCRIPT built
    and printed on the committed single-person files under the same `random` seed: the same 16 scenes, identical edge lists,
    labels, node cameras and head rows, scores bit-equal (same torch-CPU ops), both proposal lists of every graph, the four
    printed means to the last digit."""
    import random

    from conftest import generated_fixture, generated_gat_weights, proposals_as_rows
    onp = oracle()
    e = env()
    exp, arr, files, probs = generated_fixture()
    data = [json.load(open(f)) for f in files]
    random.seed(exp['seed'])
    scenes = onp.generated_dataset_scenes(data, probs, 1000)
    graphs = [g for g in (onp.generated_graph(v, e.calib) for v in scenes) if g is not None]
    assert len(graphs) == exp['n_graphs'] >= 15
    sd, prm = generated_gat_weights(exp)
    hg = exp['hash_gat']
    sd2 = pkg('synthetic').gat_state_dict(hg['seed'], prm['num_feats'], logit_gain=hg['logit_gain'], logit_shift=hg['logit_shift'])
    same_host = torch.get_num_threads() == exp.get('torch_threads')
    sm = list(e.params.used_cameras_skeleton_matching)
    blk = len(e.params.joint_list) * 10
    for i, g in enumerate(graphs):
        meta = exp['graphs'][i]
        assert g['H'] == meta['H'] and g['N'] == meta['N']
        assert np.array_equal(g['src'], arr['src_%d' % i]) and np.array_equal(g['dst'], arr['dst_%d' % i])
        assert np.array_equal(g['labels'], arr['labels_%d' % i])
        assert np.array_equal(g['edge_nodes_indices'], arr['indices_%d' % i])
        assert g['nodes_camera'] == meta['nodes_camera']
        rows = np.stack([g['feats'][h, 2 + c * blk: 2 + (c + 1) * blk].numpy() for h, c in enumerate(arr['head_cam_%d' % i])])
        assert np.array_equal(rows, arr['head_blocks_%d' % i])
        assert np.count_nonzero(g['feats'].numpy()) == np.count_nonzero(rows) + g['N']      # col 0 / col 1 + the own block only
        # (bit-equal where torch runs with the thread count the fixture was written under; the reference's own fp32 scores move by
        # up to 1.65e-5 with it, profiles/r06_reference_score_noise_by_threads.txt)
        # with 1 / 2 / 4 / 6 threads the hand-built matcher network's edge-node scores sit up to 2.05e-5 from the script's, its head-node
        # scores -- consumed by nothing -- up to 1.7e-3; the hash network's 6.6e-6)
        H_ = g['H']
        same = (lambda a, b: np.array_equal(a, b)) if same_host else \
            (lambda a, b: float(np.abs(a - b)[H_:].max()) <= 3e-5 and float(np.abs(a - b)[:H_].max()) <= 5e-3)
        sc = onp.gat_forward(sd, prm, g['feats'], g['src'], g['dst']).numpy()
        assert same(sc, arr['scores_%d' % i])
        sc2 = onp.gat_forward(sd2, prm, g['feats'], g['src'], g['dst']).numpy()          # the second (hash) weight set
        assert same(sc2, arr['scores_hash_%d' % i])
        head_cam = [sm.index(c) for c in g['nodes_camera'][:g['H']]]
        assert onp.cluster(sc2[g['H']:], g['pairs'], g['H'], head_cam, len(sm), e.params.min_number_of_views) == \
            proposals_as_rows(meta['est_hash'], sm)
    res, per = onp.sm_without_gt(graphs, lambda g: onp.gat_forward(sd, prm, g['feats'], g['src'], g['dst']).numpy(), e.params)
    for i, p in enumerate(per):
        assert p['est'] == proposals_as_rows(exp['graphs'][i]['est'], sm), i
        assert p['gt'] == proposals_as_rows(exp['graphs'][i]['gt'], sm), i
    for k, v in exp['printed'].items():
        assert res[k] == v, (k, res[k], v)
    # the fixture exercises what it is meant to: spurious heads, a person missing from a camera, ordered pairs
    assert any(len(set(m['nodes_camera'][:m['H']])) < m['H'] for m in exp['graphs'])
    pairs = graphs[0]['pairs'].tolist()
    assert [pairs[0][1], pairs[0][0]] in pairs


@pytest.mark.skipif(not os.path.isdir('/root/reference/test'), reason='the reference tree exists in the build container only')
def test_dropin_call_trace_fixture_is_what_the_reference_script_does(tmp_path):
    """tests/golden/dropin/call_trace.json -- the sequence of calls (and argument shapes) the reference's per-frame loop makes on
    the committed pinning file -- regenerated by its committed script (oracle/gen_dropin_trace.py: test/metrics_from_model.py run
    unchanged under tracing wrappers) is the committed fixture.  The GPU suite holds the package's own one-frame-per-call loop to
    it (tests/test_gpu_dropin.py::test_frame_loop_calls_the_mirrors_like_the_reference_script)."""
    import subprocess
    import sys
    env = dict(os.environ, MPE_DROPIN_TRACE_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'gen_dropin_trace.py')], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    new = json.load(open(tmp_path / 'call_trace.json'))
    old = json.load(open(os.path.join(GOLDEN, 'dropin', 'call_trace.json')))
    assert new['frames'] == old['frames'] and len(old['frames']) == 15
    names = [[e[0] for e in f] for f in old['frames']]
    assert all(f[0] == 'MergedMultipleHumansDataset' for f in names)
    assert sum(len(f) for f in names) == 102

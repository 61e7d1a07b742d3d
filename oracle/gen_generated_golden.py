"""TEST INFRASTRUCTURE (build container only): pin mode='test_generated'  with the REFERENCE'S OWN
script.

Runs /root/reference/test/sm_metrics_without_gt.py UNCHANGED (as `__main__`, Python's `random` seeded just before, the
third-party stand-ins of oracle/shims first on sys.path) on four committed single-person files and stores

  * what the script PRINTS (rand score / homogeneity / completeness / v_measure, :167-170),
  * every graph its MergedMultipleHumansDataset(..., mode='test_generated') built (graph_generator.py:672-810): edge list,
    labels, edge_nodes_indices, nodes_camera, the non-zero block of every head row,
  * the reference model's scores on every graph and the two proposal lists the script derives per graph (from the scores,
    :131, and from the labels-as-scores, :147) -- recomputed after the run with the script's own objects,
  * the scores and proposals of a SECOND weight set on the same graphs (the hash weights of the frame fixtures, through a
    second instance of the reference's GAT2): the hand-built matcher network is steep (its own fp32 scores sit up to 3e-3 from
    the float64 network on head nodes), the hash network is where the 2e-5 score bound of the frame fixtures applies,

in tests/golden/generated/.  The single-person files are cut out of synthetic 4-person frames (every person keeps its
identity code in the `prob` field, which the hand-built matcher network reads; some frames carry a spurious skeleton, a
camera with an empty list or a missing camera) and have different lengths, so `probabilities_set` (:101-104) is not flat.
The GAT weights are rebuilt by both sides from 3d_multi_pose_estimator_amd/synthetic.py (not committed).

    python oracle/gen_generated_golden.py
"""
import importlib
import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
SHIMS = os.path.join(ROOT, 'oracle', 'shims')
OUT = os.path.join(ROOT, 'tests', 'golden', 'generated')
PKG = '3d_multi_pose_estimator_amd'
SEED = 20260104
LENGTHS = (16, 14, 12, 10)
GAT_NOISE_SEED, GAT_NOISE = 5, 1e-4
HASH_GAT = {'seed': 7, 'logit_gain': 25.0, 'logit_shift': 0.698}       # the weights of the frame fixtures (tests/golden/meta.json)


def build_files(calib, syn):
    """files[k] = list of single-person frames of person k."""
    P = len(LENGTHS)
    files = [[] for _ in range(P)]
    cams = list(calib.params.camera_names)
    for i in range(max(LENGTHS)):
        spec = syn.FrameSpec(persons=P, identity_prob=True, noise_px=1.0 + 0.5 * (i % 3), spurious=1 if i % 4 == 1 else 0,
                             joint_drop=0.15 if i % 3 == 0 else 0.0)
        frame, gt = syn.make_frame(calib, 9000 + i, spec)
        for k in range(P):
            if i >= LENGTHS[k]:
                continue
            view = {}
            order = cams[(i + k) % len(cams):] + cams[:(i + k) % len(cams)] if i % 2 else cams       # dict order varies
            for ci, cam in enumerate(order):
                skeletons = json.loads(frame[cam][0])
                own = gt['owner'][cam]
                keep = [sk for sk, o in zip(skeletons, own) if o == k or (o < 0 and k == i % P)]
                if i % 5 == 2 and ci == (i + k) % len(cams):
                    continue                                   # this person is not seen by this camera: key absent
                if i % 5 == 3 and ci == k:
                    keep = []                                  # ... or present with an empty list
                if i % 7 == 6 and ci == 0 and keep:
                    keep = keep + [{'ID': 7}]                  # a skeleton without any joint key is dropped (:590-591)
                view[cam] = [json.dumps(keep), float(i), 'no_image', []]
            files[k].append(view)
    return files


CHILD = r'''
import json, pickle, random, runpy, sys
sys.dont_write_bytecode = True
sys.path[:0] = %(paths)r
import numpy as np, torch
sys.argv = ['sm_metrics_without_gt.py', '--testfiles'] + %(files)r + ['--modelsdir', %(mdir)r, '--datastep', '1']
random.seed(%(seed)d)
g = runpy.run_path(%(script)r, run_name='__main__')
ds, model, fn = g['test_dataset'], g['model'], g['get_person_proposal_from_network_output']
# a second, well-conditioned weight set on the same graphs (hash weights: scores spread over (0, 1), fp32 noise ~1e-5),
# through the reference's GAT2 built the way the script builds its own (:88-92)
params = g['params']
model2 = g['GAT'](None, params['gnn_layers'], params['num_feats'], params['n_classes'], params['num_hidden'], params['heads'],
                  params['nonlinearity'], params['final_activation'], params['in_drop'], params['attn_drop'], params['alpha'],
                  params['residual'], bias=True)
model2.load_state_dict(torch.load(%(hash_tch)r))
out = []
for i in range(len(ds)):
    graph, labels, indices, nodes_camera = ds[i]
    model.g = graph
    for layer in model.layers:
        layer.g = graph
    feats = graph.ndata['h']
    scores = torch.squeeze(model(feats.float(), graph))
    idx = torch.squeeze(indices)
    model2.g = graph
    for layer in model2.layers:
        layer.g = graph
    scores_hash = torch.squeeze(model2(feats.float(), graph))
    est_hash = fn(scores_hash, graph, idx, nodes_camera, None, 0.5)
    est = fn(scores, graph, idx, nodes_camera, None, 0.5)
    lab = [0.] * graph.number_of_nodes()
    for (j, v) in zip(idx, torch.squeeze(labels).tolist()):
        lab[j] = v
    gt = fn(lab, graph, idx, nodes_camera, None, 0.5)
    src, dst = graph.edges()
    out.append({'src': src.numpy(), 'dst': dst.numpy(), 'labels': labels.numpy(), 'indices': indices.numpy(),
                'nodes_camera': nodes_camera, 'scores': scores.numpy(), 'est': est, 'gt': gt, 'feats': feats.numpy(),
                'scores_hash': scores_hash.numpy(), 'est_hash': est_hash})
pickle.dump(out, open(%(dump)r, 'wb'))
'''


def main():
    syn = importlib.import_module(PKG + '.synthetic')
    cal = importlib.import_module(PKG + '.calibration')
    par = importlib.import_module(PKG + '.parameters')
    gh = importlib.import_module('gen_harness_golden') if os.path.join(ROOT, 'oracle') in sys.path else None
    params = par.parameters
    calib = cal.Calibration(params)
    os.makedirs(OUT, exist_ok=True)
    names = []
    for k, frames in enumerate(build_files(calib, syn)):
        path = os.path.join(OUT, 'person_%d.json' % k)
        with open(path, 'w') as fh:
            json.dump(frames, fh)
        names.append(path)
    sm = list(params.used_cameras_skeleton_matching)
    V, J = len(sm), len(params.joint_list)
    with tempfile.TemporaryDirectory() as layout:
        mdir = os.path.join(layout, 'models')
        os.makedirs(mdir)
        os.makedirs(os.path.join(layout, 'test'))
        os.symlink(os.path.join(REF, 'tm_panoptic.pickle'), os.path.join(layout, 'tm_panoptic.pickle'))
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        gh = importlib.import_module('gen_harness_golden')
        gh.GAT_NOISE_SEED, gh.GAT_NOISE = GAT_NOISE_SEED, GAT_NOISE
        gh.save_models(mdir, syn, V, J)
        dump = os.path.join(layout, 'graphs.pkl')
        import torch
        hash_tch = os.path.join(layout, 'hash_gat.tch')
        torch.save({k: torch.from_numpy(v) for k, v in syn.gat_state_dict(HASH_GAT['seed'], 2 + V * J * 10, logit_gain=HASH_GAT['logit_gain'],
                                                                          logit_shift=HASH_GAT['logit_shift']).items()}, hash_tch)
        code = CHILD % {'paths': [SHIMS, os.path.join(REF, 'skeleton_matching'), os.path.join(REF, 'utils'), REF], 'files': names,
                        'mdir': mdir, 'seed': SEED, 'hash_tch': hash_tch, 'script': os.path.join(REF, 'test', 'sm_metrics_without_gt.py'), 'dump': dump}
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
        res = subprocess.run([sys.executable, '-c', code], cwd=os.path.join(layout, 'test'), env=env, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('sm_metrics_without_gt.py failed:\n' + res.stderr[-4000:])
        print(res.stdout)
        printed = gh.parse_sm(res.stdout)
        assert len(printed) == 4, res.stdout
        graphs = pickle.load(open(dump, 'rb'))
    blk = J * 10
    arrays = {}
    meta = []
    for i, g in enumerate(graphs):
        H = int(g['indices'].reshape(-1)[0])
        N = len(g['nodes_camera'])
        feats = g['feats']
        assert feats.shape == (N, 2 + V * blk)
        assert np.all(feats[H:, 1] == 1) and np.count_nonzero(feats[H:]) == N - H            # edge-node rows: one-hot at col 1
        cam_idx = np.array([sm.index(c) for c in g['nodes_camera'][:H]], np.int32)
        head_blocks = np.stack([feats[h, 2 + c * blk: 2 + (c + 1) * blk] for h, c in enumerate(cam_idx)])
        for h, c in enumerate(cam_idx):                                                       # nothing outside col 0 + own block
            row = feats[h].copy()
            assert row[0] == 1
            row[0] = 0
            row[2 + c * blk: 2 + (c + 1) * blk] = 0
            assert not row.any()
        arrays['src_%d' % i] = g['src'].astype(np.int32)
        arrays['dst_%d' % i] = g['dst'].astype(np.int32)
        arrays['labels_%d' % i] = g['labels'].reshape(-1).astype(np.float64)
        arrays['indices_%d' % i] = g['indices'].reshape(-1).astype(np.int64)
        arrays['scores_%d' % i] = g['scores'].astype(np.float32)
        arrays['scores_hash_%d' % i] = g['scores_hash'].astype(np.float32)
        arrays['head_cam_%d' % i] = cam_idx
        arrays['head_blocks_%d' % i] = head_blocks.astype(np.float32)
        meta.append({'H': H, 'N': N, 'nodes_camera': g['nodes_camera'], 'est': g['est'], 'gt': g['gt'], 'est_hash': g['est_hash']})
    np.savez_compressed(os.path.join(OUT, 'generated_graphs.npz'), **arrays)
    report = {'printed': printed, 'seed': SEED, 'n_graphs': len(graphs), 'files': [os.path.basename(n) for n in names],
              'gat': {'kind': 'matcher', 'noise_seed': GAT_NOISE_SEED, 'noise_bound': GAT_NOISE}, 'hash_gat': HASH_GAT, 'graphs': meta,
              'numpy': np.__version__, 'python': sys.version.split()[0],
              'torch_threads': int(__import__('torch').get_num_threads())}      # the fp32 scores depend on it (profiles/r06_reference_score_noise_by_threads.txt)
    with open(os.path.join(OUT, 'generated_expected.json'), 'w') as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps({k: v for k, v in report.items() if k != 'graphs'}, indent=1))
    print('graphs:', [(m['H'], m['N'] - m['H']) for m in meta])


if __name__ == '__main__':
    main()

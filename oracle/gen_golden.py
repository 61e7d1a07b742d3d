"""TEST INFRASTRUCTURE (build container only): produce tests/golden/*.npz + *.json by
running the reference's own hot-path files (via oracle/refenv.py) on deterministic
synthetic frames and weights.  The fixtures are data: inputs and the reference's
outputs at every stage boundary of SURVEY.md §8(a).  Run:

    python oracle/gen_golden.py            # rewrites tests/golden/

The reference tree never travels to the GPU box; the fixtures do.
"""
import contextlib
import importlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
OUT = os.path.join(ROOT, 'tests', 'golden')

import refenv  # noqa: E402

GAT_SEED = 7
MLP_SEED = 11
LOGIT_GAIN = 25.0
LOGIT_SHIFT = 0.698
VARIANT_SHIFT = {'arplab': 0.698 + 0.2952, 'ring23': 0.698 - 0.1833, 'arprobot': 0.698}   # centre each variant's logits
# second MLP of every case: outputs inside the capture volume (a decoder of the triangulated points with dense
# hash noise on every weight) -- the magnitude regime the 1e-3 mm item of the north star is about
ROOM_NOISE_SEED, ROOM_NOISE = 3, 0.01


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def write_variant_parameters(params, tm, workdir):
    """A `parameters.py` with the reference's schema holding the variant's values (what a
    user obtains by editing CONFIGURATION / the preset in the reference's own file), plus
    the pickled TransformManager it points to."""
    import pickle
    sys.path.insert(0, os.path.join(ROOT, 'oracle', 'shims'))
    from pytransform3d.transform_manager import TransformManager
    os.makedirs(workdir, exist_ok=True)
    tm_path = os.path.join(workdir, 'tm_variant.pickle')
    t = TransformManager()
    for k, v in tm.transforms.items():
        t.add_transform(k[0], k[1], v)
    with open(tm_path, 'wb') as fh:
        pickle.dump(t, fh)
    fields = params._replace(transformations_path=tm_path)._asdict()
    with open(os.path.join(workdir, 'parameters.py'), 'w') as fh:
        fh.write('from collections import namedtuple\n')
        fh.write('TrackerParameters = namedtuple("TrackerParameters", %r)\n' % (list(fields.keys()),))
        fh.write('parameters = TrackerParameters(**%r)\n' % (dict(fields),))
    return fields['transformations_path']


def check_arplab_against_reference(par, robot_only=False):
    """Our ARPLAB presets against the reference's file with its CONFIGURATION switch flipped and,
    for the robot-camera variant, its two commented lines (parameters.py:110-112) swapped in for
    the all-camera ones (evaluated in memory; the file itself is never modified or copied)."""
    src = open(os.path.join(refenv.REF, 'parameters.py')).read()
    src = src.replace("CONFIGURATION = 'PANOPTIC'", "CONFIGURATION = 'ARPLAB'")
    if robot_only:
        a = "        # used_cameras=['orinbot_l', 'orinbot_r'],\n        # used_cameras_skeleton_matching = ['orinbot_l', 'orinbot_r'],"
        b = ("        used_cameras=['trackera', 'trackerb', 'trackerc', 'trackerd', 'orinbot_l', 'orinbot_r'],\n"
             "        used_cameras_skeleton_matching = ['trackera', 'trackerb', 'trackerc', 'trackerd', 'orinbot_l', 'orinbot_r'],")
        assert a in src and b in src
        src = src.replace(b, '').replace(a, a.replace('# ', ''))
    ns = {}
    exec(compile(src, 'parameters_arplab', 'exec'), ns)
    theirs, ours = ns['parameters'], par.select('ARPLAB_ROBOT' if robot_only else 'ARPLAB')
    for f in theirs._fields:
        assert getattr(theirs, f) == getattr(ours, f), f


def main(variant='panoptic'):
    global OUT
    import torch
    pkg = '3d_multi_pose_estimator_amd'
    syn = importlib.import_module(pkg + '.synthetic')
    cal = importlib.import_module(pkg + '.calibration')
    par = importlib.import_module(pkg + '.parameters')
    if variant == 'panoptic':
        our_params = par.parameters
        ref = refenv.load()
        calib = cal.Calibration(our_params)
    else:
        OUT = os.path.join(OUT, variant)
        if variant in ('arplab', 'arprobot'):
            check_arplab_against_reference(par, robot_only=variant == 'arprobot')
            our_params = par.select('ARPLAB' if variant == 'arplab' else 'ARPLAB_ROBOT')
            tm = cal.load_transform_manager(os.path.join(refenv.REF, 'tm_arp.pickle'))
        else:
            our_params = par.select('RING23')
            tm = syn.ring_transform_manager(our_params)
        workdir = '/tmp/mpe_variant_' + variant
        tm_path = write_variant_parameters(our_params, tm, workdir)
        ref = refenv.load(parameters_dir=workdir)
        our_params = our_params._replace(transformations_path=tm_path)
        calib = cal.Calibration(our_params, tm)
    os.makedirs(OUT, exist_ok=True)
    rparams = ref['parameters'].parameters
    # our schema must agree with the module the reference imported, field by field
    for f in rparams._fields:
        assert getattr(rparams, f) == getattr(our_params, f), f
    gg = ref['graph_generator']
    nf = len(gg.HumanGraphFromView.get_all_features('3'))

    # calibration globals of the reference (a2)
    np.savez_compressed(
        os.path.join(OUT, 'calibration_%s.npz' % variant),
        T_d=np.stack([t.numpy() for t in gg.camera_d_transforms]),
        T_i32=np.stack([t.numpy() for t in gg.camera_i_transforms]),
        K32=np.stack([t.cpu().numpy() for t in gg.camera_matrices]),
        Kinv32=np.stack([t.numpy() for t in gg.inverse_camera_matrices]),
        centre32=np.stack([t.numpy() for t in gg.all_cameras_from_root]),
        dist=np.stack([ref['pose_estimator_dataset_from_json'].distortion_coefficients[c] for c in rparams.camera_names]),
        P=np.stack([ref['pose_estimator_dataset_from_json'].projection_matrices[c] for c in rparams.camera_names]),
        features=np.array(gg.HumanGraphFromView.get_all_features('3')),
    )

    shift = VARIANT_SHIFT.get(variant, LOGIT_SHIFT)
    gat_sd = syn.gat_state_dict(GAT_SEED, nf, logit_gain=LOGIT_GAIN, logit_shift=shift)
    model = ref['gat2'].GAT2(None, syn.GAT_LAYERS, nf, 1, syn.GAT_HIDDEN, syn.GAT_HEADS, torch.nn.LeakyReLU(),
                             torch.nn.Sigmoid(), 0., 0., syn.GAT_ALPHA, False, bias=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in gat_sd.items()})
    # metrics_from_model.py:91 sizes the MLP from len(parameters.cameras), the dataset sizes its rows from
    # len(parameters.used_cameras) (pose_estimator_dataset_from_json.py:237-285, reprojection_error.py:143): equal for the
    # all-camera presets; with the robot-camera subset only the second is consistent with the rows, so that is used
    n_used = len(rparams.used_cameras)
    in_dim = n_used * len(rparams.joint_list) * rparams.numbers_per_joint
    mlp_sd = syn.mlp_state_dict(MLP_SEED, in_dim)
    with quiet():
        mlp = ref['mlp'].PoseEstimatorMLP(input_dimensions=in_dim, output_dimensions=54)
    mlp.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_sd.items()})
    room_sd = syn.decoder_mlp_state_dict(n_used, len(rparams.joint_list), rparams.numbers_per_joint,
                                         noise_seed=ROOM_NOISE_SEED, noise_bound=ROOM_NOISE)
    with quiet():
        mlp_room = ref['mlp'].PoseEstimatorMLP(input_dimensions=in_dim, output_dimensions=54)
    mlp_room.load_state_dict({k: torch.from_numpy(v) for k, v in room_sd.items()})

    F = syn.FrameSpec
    names = list(rparams.camera_names)
    if variant == 'arplab':
        cases = [('arp_6x3', F(persons=3, add_id_key=True, noise_px=1.0, float_conf=False), [0, 1]),
                 ('arp_robot_pair', F(persons=2, cameras=['orinbot_r', 'orinbot_l']), [2]),
                 ('fz_random_shapes', random_shape_specs(names, 30, 707), list(range(700, 730)))]      # round 6, as for PANOPTIC below
    elif variant == 'arprobot':
        cases = [('arp_robot_only', F(persons=3, cameras=['orinbot_l', 'orinbot_r'], noise_px=1.0), [0, 1]),
                 ('arp_robot_only_all_streams', F(persons=2, add_id_key=True), [3])]   # the other four cameras' skeletons arrive too
    elif variant == 'ring23':
        cases = [('ring23x3', F(persons=3, noise_px=0.5), [0]),
                 ('ring23_sparse', F(persons=2, cameras=names[3::4], joint_drop=0.1), [1]),
                 # round 6, as for PANOPTIC below; at most three skeletons per camera (the capacity of the tests' 23-camera engines)
                 ('fz_random_shapes', random_shape_specs(names, 8, 808, max_persons=2, max_spurious=1), list(range(800, 808)))]
    else:
      cases = [
        ('c1_2view_1person', F(persons=1, cameras=['trackera', 'trackerb']), [0, 1]),
        ('c2_5x4_clean', F(persons=4), [0, 1, 2]),
        ('c2_5x4_messy', F(persons=4, noise_px=2.0, joint_drop=0.3, add_id_key=True, spurious=1,
                           empty_cameras=('trackerc',), float_conf=False), [3, 4]),
        ('c2_5x4_reordered', F(persons=4, cameras=['trackerd', 'trackera', 'trackere', 'trackerb', 'trackerc'],
                               joint_drop=0.15), [5]),
        ('c2_3x2', F(persons=2, cameras=['trackerb', 'trackerd', 'trackere']), [6]),
        ('c4_5x10', F(persons=10, noise_px=1.0), [7]),
      ]
      # round 6: frames of RANDOM shape through the reference (the GPU path and the oracle had met such frames only against each other,
      # tests/checkers/shape_fuzz.py): camera subsets and orders, empty cameras, spurious skeletons, dropped joints, "ID" keys, noise.
      # One spec per frame; every frame keeps at least two cameras with skeletons (the reference builds no graph otherwise).
      cases.append(('fz_random_shapes', random_shape_specs(names, 40, 606), list(range(600, 640))))
    meta = {'gat_seed': GAT_SEED, 'mlp_seed': MLP_SEED, 'logit_gain': LOGIT_GAIN, 'logit_shift': shift,
            'num_feats': nf, 'mlp_in': in_dim, 'variant': variant, 'cases': {},
            'room_mlp': {'kind': 'decoder', 'noise_seed': ROOM_NOISE_SEED, 'noise_bound': ROOM_NOISE},
            # the reference's fp32 scores depend on how MKL partitions its sums, i.e. on the host's thread count (up to 1.65e-5 in a score,
            # profiles/r06_reference_score_noise_by_threads.txt): the oracle's torch-CPU evaluation equals them to a few ulp only under the same count
            'torch_threads': int(torch.get_num_threads())}
    for name, spec, idxs in cases:
        frames_json = []
        arrays = {}
        for n, fi in enumerate(idxs):
            frame, gt = syn.make_frame(calib, fi, spec[n] if isinstance(spec, list) else spec)
            frames_json.append(frame)
            # ---- callers' pre-processing (metrics_from_model.py:182-191)
            pi = {}
            for cam in frame:
                data = json.loads(frame[cam][0])
                if data:
                    pi[cam] = [json.dumps(data), frame[cam][1]]
            with quiet():
                sc = gg.MergedMultipleHumansDataset(pi, mode='test', limit=10000, debug=True, alt='3', verbose=False)
            assert len(sc.graphs) == 1
            g = sc.graphs[0]
            feats = g.ndata['h']
            src, dst = [x.numpy() for x in g.edges()]
            idx = torch.squeeze(sc.data['edge_nodes_indices'][0], 1)
            nodes_camera = sc.data['nodes_camera'][0]
            model.g = g
            for layer in model.layers:
                layer.g = g
            # per-layer activations (what GAT2.forward computes, gat2.py:137-149)
            h = feats.float()
            inter = []
            for l in range(model.num_layers - 1):
                h = model.activation(model.layers[l](h).flatten(1))
                inter.append(h.numpy().copy())
            outputs = torch.squeeze(model(feats.float(), g))
            final_output = ref['skeleton_matching_utils'].get_person_proposal_from_network_output(
                outputs, g, idx, nodes_camera, sc.jsons_for_head, 0.5)
            nz = torch.nonzero(feats)
            p = 'f%d_' % n
            arrays[p + 'N'] = np.int64(g.number_of_nodes())
            arrays[p + 'src'] = src
            arrays[p + 'dst'] = dst
            arrays[p + 'feat_rc'] = nz.numpy().astype(np.int32)
            arrays[p + 'feat_v'] = feats[nz[:, 0], nz[:, 1]].numpy()
            arrays[p + 'edge_nodes_indices'] = idx.numpy()
            arrays[p + 'nodes_camera'] = np.array(nodes_camera)
            arrays[p + 'skeleton_index'] = np.array([sc.skeleton_index[i] for i in range(len(sc.skeleton_index))], np.int32)
            arrays[p + 'scores'] = outputs.numpy()
            for l, a in enumerate(inter):
                arrays[p + 'act%d_head' % l] = a[:4]                       # first heads
                arrays[p + 'act%d_en' % l] = a[int(idx[0]):int(idx[0]) + 4]  # first edge-nodes
            persons = np.array([[(-1 if fo[c] is None else fo[c]) for c in rparams.used_cameras_skeleton_matching]
                                for fo in final_output], np.int32).reshape(-1, len(rparams.used_cameras_skeleton_matching))
            print(name, n, 'scores>0.5: %d of %d, persons %d' % (int((outputs[idx] > 0.5).sum()), len(idx), len(persons)))
            arrays[p + 'persons'] = persons
            # ---- 3D stage A (metrics_from_model.py:243-294)
            rows = []
            for person in final_output:
                raw_input = {}
                for camera in rparams.used_cameras:
                    if person[camera] is not None:
                        raw_input[camera] = [json.dumps([sc.jsons_for_head[person[camera]]])]
                with quiet():
                    ds = ref['pose_estimator_dataset_from_json'].PoseEstimatorDataset(
                        raw_input, rparams.cameras, rparams.joint_list, save=False)
                assert len(ds) == 1
                rows.append(ds[0][0].numpy())
            if rows:
                x = torch.from_numpy(np.stack(rows))
                out = mlp(x)
                arrays[p + 'mlp_in'] = np.stack(rows)
                arrays[p + 'mlp_out'] = out.numpy()
                arrays[p + 'poses'] = np.stack([(out[i] * 10.).numpy().reshape(-1, 3) for i in range(out.shape[0])])
                out_room = mlp_room(x)
                arrays[p + 'mlp_out_room'] = out_room.numpy()
                arrays[p + 'poses_room'] = np.stack([(out_room[i] * 10.).numpy().reshape(-1, 3) for i in range(out_room.shape[0])])
            # ---- 3D stage B (metrics_from_triangulation.py:234-272)
            pe = ref['pose_estimator_dataset_from_json']
            tri = np.zeros((len(final_output), len(rparams.joint_list), 3))
            tri_valid = np.zeros((len(final_output), len(rparams.joint_list)), np.int8)
            tri_all = np.zeros((len(final_output), len(rparams.joint_list), 3))
            cam_matrix = {c: ref['pose_estimator_utils'].camera_matrix(i).cpu().numpy() for i, c in enumerate(rparams.camera_names)}
            for pi_, person in enumerate(final_output):
                points_2D = {}
                for cam_idx in rparams.cameras:
                    camera = rparams.camera_names[cam_idx]
                    # person dicts carry used_cameras_skeleton_matching only (skeleton_matching_utils.py:125): with a camera
                    # subset the caller's person[camera] (metrics_from_triangulation.py:240) raises KeyError for the unused
                    # cameras; the fixture keeps the gather over the cameras the dict has
                    if person.get(camera) is not None:
                        for j, pos in sc.jsons_for_head[person[camera]].items():
                            if j == 'ID':
                                continue      # an "ID" entry would make the reference raise here; fixtures avoid it
                            points_2D.setdefault(j, {})[camera] = np.array([pos[1], pos[2]])
                r3 = ref['pose_estimator_utils'].triangulate(points_2D, cam_matrix, pe.distortion_coefficients,
                                                              pe.projection_matrices, rparams.axes_3D['Y'][0])
                for j in rparams.joint_list:
                    if str(j) in r3:
                        tri_valid[pi_, j] = 1
                        tri_all[pi_, j] = [r3[str(j)][0][0], r3[str(j)][1][0], r3[str(j)][2][0]]
                        if j in rparams.used_joints:
                            tri[pi_, j] = [r3[str(j)][0][0], r3[str(j)][1][0], r3[str(j)][2][0]]
            arrays[p + 'tri'] = tri
            arrays[p + 'tri_valid'] = tri_valid
            arrays[p + 'tri_all'] = tri_all
            arrays[p + 'gt'] = gt['persons']
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
        with open(os.path.join(OUT, name + '.frames.json'), 'w') as fh:
            json.dump(frames_json, fh)
        meta['cases'][name] = {'frames': len(idxs)}
        print(name, 'ok', {k: v.shape for k, v in arrays.items() if k.startswith('f0_') and hasattr(v, 'shape')})

    if variant in ('panoptic', 'ring23'):
        gen_cluster_cases(ref, meta)
    if variant == 'arprobot':
        # this preset is where the two statements above differ from an UNMODIFIED reference run: say so next to the fixtures
        meta['deviations_from_an_unmodified_reference_run'] = [
            "MLP input width from len(parameters.used_cameras) (oracle/gen_golden.py), where test/metrics_from_model.py:91 uses len(parameters.cameras): with the camera-subset preset only the former matches the rows PoseEstimatorDataset builds (pose_estimator_dataset_from_json.py:237-285); numerically neutral for the all-camera presets",
            "the triangulation gather reads person.get(camera) where test/metrics_from_triangulation.py:240 reads person[camera]: with this preset the reference's own statement raises KeyError for the cameras outside used_cameras_skeleton_matching, so these fixtures pin the builder's reading of the script, not an unmodified run"]
    with open(os.path.join(OUT, 'meta.json'), 'w') as fh:
        json.dump(meta, fh, indent=1)


def random_shape_specs(names, n, seed, max_persons=6, max_spurious=2):
    """n FrameSpecs of random shape (the generator of tests/checkers/shape_fuzz.py, restricted to frames the reference builds a graph
    for: at least two cameras that hold skeletons, at least one person)."""
    from importlib import import_module
    syn = import_module('3d_multi_pose_estimator_amd.synthetic')
    rng = np.random.RandomState(seed)
    specs = []
    while len(specs) < n:
        k = rng.randint(2, len(names) + 1) if rng.rand() < 0.7 else len(names)
        cams = [str(c) for c in rng.permutation(names)[:k]]
        empty = tuple(c for c in cams if rng.rand() < 0.15)
        if len(cams) - len(empty) < 2:
            continue
        specs.append(syn.FrameSpec(persons=int(rng.randint(1, max_persons + 1)), cameras=cams, noise_px=float(rng.choice([0.0, 1.0, 3.0])),
                                   joint_drop=float(rng.choice([0.0, 0.2, 0.5])), add_id_key=bool(rng.rand() < 0.3),
                                   spurious=int(rng.randint(0, max_spurious + 1)), empty_cameras=empty, float_conf=bool(rng.rand() < 0.7)))
    return specs


class _G:
    """Duck-typed stand-in for the graph argument of get_person_proposal_from_network_output
    (it only calls .edges(), skeleton_matching_utils.py:26)."""

    def __init__(self, src, dst):
        import torch
        self._s, self._d = torch.tensor(src), torch.tensor(dst)

    def edges(self):
        return self._s, self._d


def gen_cluster_cases(ref, meta):
    """Clustering-only known answers: random and adversarial score vectors through the
    reference's get_person_proposal_from_network_output (real file, real networkx)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle_np as onp
    rparams = ref['parameters'].parameters
    cams = list(rparams.used_cameras_skeleton_matching)
    rng = np.random.default_rng(2024)
    fn = ref['skeleton_matching_utils'].get_person_proposal_from_network_output
    cases = []
    big = len(cams) > 8          # 23-camera rig: many cameras, few skeletons each (set growth, long BFS)
    for ci in range(120 if big else 400):
        if big:
            ncam = int(rng.integers(6, len(cams) + 1))
            order = list(rng.permutation(len(cams))[:ncam])
            counts = [int(rng.integers(1, 4)) for _ in order]
        else:
            ncam = int(rng.integers(2, 6))
            order = list(rng.permutation(5)[:ncam])
            counts = [int(rng.integers(1, 5 if ci % 4 else 11)) for _ in order]
        slots = []
        hid = 0
        for c, n in zip(order, counts):
            slots.append((cams[c], list(range(hid, hid + n))))
            hid += n
        N, src, dst, pairs = onp.topology(slots)
        H = hid
        M = N - H
        if M == 0:
            continue
        head_cam = [cams.index(s[0]) for s in slots for _ in s[1]]
        kind = ci % 5
        if kind == 0:      # uniform scores
            sc = rng.uniform(0, 1, M)
        elif kind == 1:    # trained-like: true pairs high, others low, with noise and some errors
            owner = [int(rng.integers(0, max(counts))) for _ in range(H)]
            sc = np.array([(0.9 if owner[a] == owner[b] else 0.1) for a, b in pairs]) + rng.normal(0, 0.25, M)
            sc = np.clip(sc, 0, 1)
        elif kind == 2:    # heavy ties (saturated sigmoid)
            sc = rng.choice([1.0, 1.0, 0.99999994, 0.75, 0.5, 0.4999], M)
        elif kind == 3:    # everything matches (merge-quirk territory)
            sc = 0.5 + 0.5 * rng.uniform(0, 1, M) ** 0.3
        else:              # coarse grid -> many ties among mid scores
            sc = np.round(rng.uniform(0.3, 1.0, M), 1)
        sc32 = sc.astype(np.float32)
        outputs = torch.zeros(N)
        outputs[H:] = torch.from_numpy(sc32)
        nodes_camera = [s[0] for s in slots for _ in s[1]] + [''] * M
        res = fn(outputs, _G(src.tolist(), dst.tolist()), torch.arange(H, N), nodes_camera, None, 0.5)
        persons = np.array([[(-1 if r[c] is None else r[c]) for c in cams] for r in res], np.int32).reshape(-1, len(cams))
        cases.append({'slot_cam': [cams.index(s[0]) for s in slots], 'slot_n': counts,
                      'scores': sc32, 'persons': persons})
    arrays = {}
    for i, c in enumerate(cases):
        arrays['c%d_slot_cam' % i] = np.array(c['slot_cam'], np.int32)
        arrays['c%d_slot_n' % i] = np.array(c['slot_n'], np.int32)
        arrays['c%d_scores' % i] = c['scores']
        arrays['c%d_persons' % i] = c['persons']
    np.savez_compressed(os.path.join(OUT, 'cluster_cases.npz'), n=np.int64(len(cases)), **arrays)
    meta['cluster_cases'] = len(cases)
    print('cluster cases', len(cases))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'panoptic')

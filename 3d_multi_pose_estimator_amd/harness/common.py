"""Evaluation harness shared by metrics_from_model.py / metrics_from_triangulation.py: the
counterpart of the reference's two callers (test/metrics_from_model.py,
test/metrics_from_triangulation.py), restated on top of the batched engine.

Kept from the reference: CLI flags (--testfiles --tmdir --modelsdir --datastep, :27-35), the
frame stride (:124), the skip rules (no GT field -> exit :129-131; no GT bodies -> skip
:137-138; no cross-camera pair -> skip :195-196), ground truth brought to world coordinates
through camera 1 of the dataset calibration (:152-161), per-person error = mean joint distance
over used_joints (:303-320), assignment by exhaustive permutation search (:322-337), MPJPE /
AP / recall bookkeeping at 25..150 mm (:339-385) and the printed quantities (:382-390), plus
frames/s.  Frames are processed in batches instead of one by one.
"""
import argparse
import itertools
import json
import os
import pickle
import sys
import time

import numpy as np
import torch

from .. import synthetic
from ..calibration import Calibration, load_transform_manager
from ..parameters import parameters
from ..pipeline import Engine

THRESHOLDS_MM = np.arange(25, 155, 25)


def build_parser(description):
    p = argparse.ArgumentParser(description=description)
    p.add_argument('--testfiles', type=str, nargs='+', required=False, default=[], help='List of json files used as input')
    p.add_argument('--tmdir', type=str, nargs=1, required=False, default=['.'],
                   help='Directory that contains the files with the transfomation matrices')
    p.add_argument('--modelsdir', type=str, nargs='?', required=False, default='../models/',
                   help="Directory that contains the models' files")
    p.add_argument('--datastep', type=int, nargs='?', required=False, default=12, help='Data step used to compute the metrics')
    p.add_argument('--batch', type=int, default=256, help='frames per device batch')
    p.add_argument('--synthetic', type=int, default=0, help='generate this many synthetic Panoptic-shaped frames instead of --testfiles')
    p.add_argument('--random-weights', action='store_true', help='deterministic hash-initialised weights (no checkpoint offline)')
    p.add_argument('--teacher-scores', action='store_true',
                   help='synthetic frames only: replace the GAT scores by the ground-truth pairing (isolates the 3D stage)')
    p.add_argument('--noise-px', type=float, default=0.0)
    p.add_argument('--persons', type=int, default=4)
    p.add_argument('--gat-acc64', action='store_true',
                   help='f64 running sums in every GEMM of the matching network (Engine.set_precision(gat_acc64=True); default: single fp32 chains for K <= 512)')
    p.add_argument('--mlp-precision', choices=['default', 'max_accuracy', 'f64'], default='default',
                   help='MLP mode 3 (default: f64 sums every second K stage), 4 (an f64 flush per stage) or 5 (the network evaluated on the f64 matrix pipe)')
    return p


def load_models(eng, args, need_mlp):
    """skeleton_matching.prms/.tch and pose_estimator.pytorch (metrics_from_model.py:89-100),
    or deterministic weights when none are available."""
    V, J = eng.V, eng.J
    nf = 2 + V * J * 10
    mdir = args.modelsdir if args.modelsdir.endswith('/') else args.modelsdir + '/'
    if args.random_weights or not os.path.exists(mdir + 'skeleton_matching.tch'):
        if not args.random_weights:
            print('no model files under %s: using deterministic random weights' % mdir)
        eng.load_gat(synthetic.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698), synthetic.gat_params(nf))
        if need_mlp:
            eng.load_mlp(synthetic.mlp_state_dict(11, V * J * parameters.numbers_per_joint))
        return
    params = pickle.load(open(mdir + 'skeleton_matching.prms', 'rb'))
    eng.load_gat(torch.load(mdir + 'skeleton_matching.tch', map_location='cpu'), params)
    if need_mlp:
        saved = torch.load(mdir + 'pose_estimator.pytorch', map_location='cpu')
        eng.load_mlp(saved['model_state_dict'])


def dataset_transform(tm_dir, file_name):
    """tm_<a>_<b>.pickle next to the test file's name (metrics_from_model.py:109-115)."""
    base = os.path.basename(file_name).split('_')
    stem = os.path.join(tm_dir, 'tm_' + base[0] + '_' + base[1])
    for ext in ('.pickle', '.json'):
        if os.path.exists(stem + ext):
            return load_transform_manager(stem + ext)
    raise FileNotFoundError(stem + '.pickle')


def ground_truth(frame, T_dataset_cam1, T_i_cam1):
    """-> (list of {joint idx: (3,) f32 world}, valid flags) or None to skip; raises SystemExit
    when the file has no GT field (reference :126-174)."""
    first_cam = list(frame.keys())[0]
    if len(frame[first_cam]) != 4:
        print('There is no ground truth in the specified file')
        raise SystemExit
    for c in frame:
        if len(frame[c][3]) > len(frame[first_cam][3]):
            first_cam = c
    bodies = frame[first_cam][3]
    if len(bodies) == 0:
        return None
    J = len(parameters.joint_list)
    gts, valid = [], []
    for body in bodies:
        g = torch.zeros(J * 3)
        for j in parameters.joint_list:
            if str(j) in body:
                g[j * 3: j * 3 + 3] = torch.tensor(np.array(body[str(j)]) / 100.)
        g = g.reshape((J, 3)).transpose(0, 1)
        hom = torch.cat((g, torch.ones(1, J)), 0)
        world = torch.matmul(T_i_cam1, torch.matmul(T_dataset_cam1, hom))[:-1].transpose(0, 1)
        gts.append({j: world[j].numpy() for j in parameters.joint_list if str(j) in body})
        valid.append('-1' in body)
    return gts, valid


class Metrics:
    """Bookkeeping of the two reference callers (metrics_from_model.py:303-390,
    metrics_from_triangulation.py:281-372), statement for statement.  The scripts differ in one
    place: the triangulation script marks a detection invalid when a used joint of the ground
    truth is missing from it (:297-298) and then books it as a false positive at every threshold
    (:333), while its MPJPE sum still counts it (:323-327)."""

    def __init__(self):
        self.acc_err = 0.0
        self.n_matching = 0
        self.n_poses = 0
        self.n_gt = 0
        self.TP = [[] for _ in THRESHOLDS_MM]
        self.FP = [[] for _ in THRESHOLDS_MM]

    def add_frame(self, gts, valid_gt, results, triangulation=False):
        """gts: list of {joint str or int: (3,)}; results: list of {joint idx: (3,)} in the order
        the 3D stage produced them."""
        G, R = len(gts), len(results)
        table = np.zeros((G, R))
        valid_detection = [True] * R
        for i, gt in enumerate(gts):
            for r, res in enumerate(results):
                tot, n = 0.0, 0
                for j, g in gt.items():
                    idx = int(j)
                    if idx in parameters.used_joints:
                        if idx in res:
                            tot += np.linalg.norm(np.asarray(res[idx]) - g)
                            n += 1
                        else:
                            valid_detection[r] = False
                if n > 0:
                    table[i, r] = tot / n
        perms = itertools.permutations(range(R), G) if G <= R else itertools.permutations(range(G), G)
        best, best_p = 10000., None
        for p in perms:
            acc = 0
            for i, r in enumerate(p):
                if r < R:
                    acc += table[i, r]
            if acc < best:
                best, best_p = acc, p
        self.n_poses += R
        self.n_gt += G
        for r in range(R):
            if r in best_p:
                i = best_p.index(r)
                if valid_gt[i]:
                    self.n_matching += 1
                    self.acc_err += table[i, r]
                else:
                    self.n_gt -= 1
            for k, th in enumerate(THRESHOLDS_MM):
                if r in best_p and (valid_detection[r] or not triangulation):
                    i = best_p.index(r)
                    if not valid_gt[i]:
                        continue
                    ok = table[i, r] * 1000. < th
                    self.TP[k].append(1 if ok else 0)
                    self.FP[k].append(0 if ok else 1)
                else:
                    self.TP[k].append(0)
                    self.FP[k].append(1)

    def report(self):
        out = {'ap': {}}
        for k, th in enumerate(THRESHOLDS_MM):
            tp, fp = np.cumsum(np.array(self.TP[k])), np.cumsum(np.array(self.FP[k]))
            recall = tp / (self.n_gt + 1e-5)
            precise = tp / (tp + fp + 1e-5)
            for n in range(len(self.TP[k]) - 2, -1, -1):
                precise[n] = max(precise[n], precise[n + 1])
            precise = np.concatenate(([0], precise, [0]))
            recall = np.concatenate(([0], recall, [1]))
            idx = np.where(recall[1:] != recall[:-1])[0]
            ap = np.sum((recall[idx + 1] - recall[idx]) * precise[idx + 1])
            print('AP, precise and recall for', th, ':', ap, precise[-2], recall[-2])
            out['ap'][str(int(th))] = [float(ap), float(precise[-2]), float(recall[-2])]
        if self.n_matching > 0:
            print('MEAN ERR (mm)', self.acc_err * 1000. / self.n_matching)
            out['mpjpe_mm'] = self.acc_err * 1000. / self.n_matching
        return out


def teacher_scores(db, owners):
    """Ground-truth pairing as scores: 1 for two views of one person, 0 otherwise."""
    from ..packing import pairs_of_frame
    sc = np.zeros(db.n_edge_nodes, np.float32)
    pb = db.host
    sm = list(parameters.used_cameras_skeleton_matching)
    for f in range(pb.n_frames):
        h0, H, e0, M = pb.frame_counts(f)
        own = []
        for s in range(pb.V):
            c = pb.slot_cam[f, s]
            if c < 0:
                continue
            cam_owner = owners[f][sm[c]]
            # heads of the slot follow list order; skeletons without joints were skipped
            own += [cam_owner[i] for i in pb.skeleton_index[h0 + len(own): h0 + len(own) + pb.slot_n[f, s]]]
        for m, (a, b) in enumerate(pairs_of_frame(pb.slot_n[f])):
            sc[e0 + m] = 1.0 if (own[a] == own[b] and own[a] >= 0) else 0.0
    return torch.from_numpy(sc)


def collect_work(args, calib):
    """[(frame, T_dataset_cam1, owners or None)] honouring --datastep across files (the counter
    runs over all files, metrics_from_model.py:102-124)."""
    work = []
    if args.synthetic:
        spec = synthetic.FrameSpec(persons=args.persons, noise_px=args.noise_px)
        for i in range(args.synthetic):
            f, gt = synthetic.make_frame(calib, i, spec)
            work.append((f, torch.from_numpy(calib.T_d[1]).type(torch.float32), gt['owner']))
        return work
    tm_dir = args.tmdir[0]
    n_input = 0
    for file in args.testfiles:
        print(file)
        T_d1 = torch.from_numpy(dataset_transform(tm_dir, file).get_transform('root', parameters.camera_names[1])).type(torch.float32)
        for frame in json.load(open(file, 'rb')):
            n_input += 1
            if (n_input - 1) % args.datastep == 0:
                work.append((frame, T_d1, None))
    return work


def evaluate(work, infer, mode, T_i1, batch=256):
    """The callers' loop around the inference path.  `infer(frames, owners)` receives the
    pre-processed frames of one batch (cameras with an empty skeleton list dropped, :182-191) and
    returns, per frame, None when the frame has no cross-camera pair (no graph, :195-196) or the
    list of 3D results {joint idx: (3,)} in production order.  Returns (Metrics, n_data, n_results)."""
    metrics = Metrics()
    n_data = n_results = 0
    for start in range(0, len(work), batch):
        keep, gts = [], []
        for frame, T_d1, owners in work[start:start + batch]:
            gt = ground_truth(frame, T_d1, T_i1)
            if gt is None:
                continue
            keep.append((frame, owners))
            gts.append(gt)
        if not keep:
            continue
        frames = [{c: [frame[c][0], frame[c][1]] for c in frame if json.loads(frame[c][0])} for frame, _ in keep]
        per_frame = infer(frames, [o for _, o in keep])
        for f, results in enumerate(per_frame):
            if results is None:
                continue
            n_data += 1
            n_results += len(results)
            metrics.add_frame(gts[f][0], gts[f][1], results, triangulation=(mode != 'mlp'))
    return metrics, n_data, n_results


def max_skeletons_per_camera(work):
    """Largest skeleton list of any camera in the selected frames: sizes the engine's per-frame
    capacity (the reference has no such limit; its graphs simply grow)."""
    m = 1
    for frame, _, _ in work:
        for cam in frame:
            m = max(m, frame[cam][0].count('{'))       # one '{' per skeleton dict in the JSON text
    return m


def run(args, mode):
    calib = Calibration(parameters)
    work = collect_work(args, calib)
    eng = Engine(parameters, calib, max_frames=args.batch,
                 max_persons_per_camera=max(4, args.persons + 1, max_skeletons_per_camera(work)))
    if getattr(args, 'gat_acc64', False) or getattr(args, 'mlp_precision', 'default') != 'default':
        eng.set_precision(gat_acc64=bool(getattr(args, 'gat_acc64', False)), mlp_max_accuracy=getattr(args, 'mlp_precision', '') == 'max_accuracy',
                          mlp_f64=getattr(args, 'mlp_precision', '') == 'f64')
    load_models(eng, args, need_mlp=(mode == 'mlp'))
    T_i1 = torch.from_numpy(calib.T_i32[1])
    J = eng.J
    t = {'match': 0.0, '3d': 0.0}

    def infer(frames, owners):
        db = eng.to_device(eng.pack(frames))
        torch.cuda.synchronize()
        t0 = time.time()
        if args.teacher_scores and owners[0] is not None:
            persons, n_persons = eng.cluster(db, teacher_scores(db, owners))
        else:
            _, persons, n_persons = eng.match(db, want_scores=False)
        torch.cuda.synchronize()
        t1 = time.time()
        if mode == 'mlp':
            poses, valid = eng.mlp3d(db, persons, n_persons)
        else:
            poses, jvalid = eng.triangulate(db, persons, n_persons)
        torch.cuda.synchronize()
        t['match'] += t1 - t0
        t['3d'] += time.time() - t1
        eng.sync_status()
        n_np, poses = n_persons.cpu().numpy(), poses.cpu().numpy()
        flags = (valid if mode == 'mlp' else jvalid).cpu().numpy()
        out = []
        for f in range(len(frames)):
            h0, H, e0, M = db.host.frame_counts(f)
            if M == 0:
                out.append(None)
                continue
            results = []
            for p in range(int(n_np[f])):
                if mode == 'mlp':
                    if flags[f, p]:
                        results.append({j: poses[f, p, j] for j in range(J)})
                else:
                    results.append({j: poses[f, p, j] for j in range(J) if flags[f, p, j]})
            out.append(results)
        return out

    metrics, n_data, n_results = evaluate(work, infer, mode, T_i1, args.batch)
    out = metrics.report()
    if n_data > 0:
        print('Mean time for graph matching', t['match'] / n_data)
        print('Mean time for graph matching (per person)', t['match'] / max(1, n_results))
        print('Mean time for 3D', t['3d'] / n_data)
        print('Mean time for 3D (per person)', t['3d'] / max(1, n_results))
        print('Frames per second', n_data / max(1e-9, t['match'] + t['3d']))
    out['n_data'] = n_data
    eng.close()
    return out

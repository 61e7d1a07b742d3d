"""Counterpart of the reference's test/sm_metrics.py on the MI355X path: clustering quality of
the skeleton-matching stage (adjusted Rand index, homogeneity, completeness, V-measure) against
a ground-truth grouping built from the per-skeleton 3D bodies.

Kept from the reference (sm_metrics.py:107-229): frame stride; GT persons built by greedy 3D
proximity over the cameras' bodies_3D lists (a skeleton joins the nearest GT person when the
mean joint distance is <= 1 unit, else founds a new one, :128-160); frames with a body lacking
the '-1' key or without any GT are skipped (:131-132, :166-167); one label per head node:
index of the proposal containing it, or len(proposals) for unassigned heads (:208-216);
metrics averaged over frames (:218-229).  Matching itself runs batched on the device.
"""
import copy
import json

import numpy as np
import torch
from sklearn.metrics import adjusted_rand_score, homogeneity_completeness_v_measure

from .. import synthetic
from ..calibration import Calibration
from ..parameters import parameters
from ..pipeline import Engine
from .common import build_parser, load_models, teacher_scores


def gt_labels(frame):
    """-> list of GT person ids, one per skeleton in (camera, list) order, or None to skip."""
    gt_people, labels, valid = [], [], True
    for cam in frame:
        if cam not in parameters.used_cameras:
            continue
        for id_skeleton, joints_3D in enumerate(frame[cam][3]):
            if '-1' not in joints_3D:
                valid = False
            best, matched, n_joints = 1000000000., -1, 0
            for pid, person in enumerate(gt_people):
                dist, n = 0.0, 0
                for idx, p3D in person.items():
                    if idx in joints_3D:
                        dist += np.linalg.norm(np.array(joints_3D[idx]) - np.array(p3D))
                        n += 1
                if dist < best:
                    best, matched, n_joints = dist, pid, n
            if n_joints == 0 or best / n_joints > 1.:
                matched = -1
            if matched < 0:
                matched = len(gt_people)
                gt_people.append(copy.deepcopy(joints_3D))
            labels.append(matched)
    if not gt_people or not valid:
        return None
    return labels


def collect_work(args, calib):
    work = []
    if args.synthetic:
        spec = synthetic.FrameSpec(persons=args.persons, noise_px=args.noise_px)
        for i in range(args.synthetic):
            f, gt = synthetic.make_frame(calib, i, spec)
            work.append((f, gt['owner']))
        return work
    n_input = 0
    for file in args.testfiles:
        print(file)
        for frame in json.load(open(file, 'rb')):
            n_input += 1
            if (n_input - 1) % args.datastep == 0:
                if len(frame[list(frame.keys())[0]]) != 4:
                    print('There is no ground truth in the specified file')
                    raise SystemExit
                work.append((frame, None))
    return work


def evaluate(work, infer, batch=256):
    """`infer(frames, owners)` -> per frame None (no graph, reference :186-187) or
    (H, proposals) with proposals = list of lists of head ids.  Returns the four averages."""
    tot = {'rand score': 0.0, 'homogeneity': 0.0, 'completeness': 0.0, 'v_measure': 0.0}
    n_data = 0
    for start in range(0, len(work), batch):
        chunk = [(f, o, gt_labels(f)) for f, o in work[start:start + batch]]
        chunk = [c for c in chunk if c[2] is not None]
        if not chunk:
            continue
        frames = [{c: [f[c][0], f[c][1]] for c in f if json.loads(f[c][0])} for f, _, _ in chunk]
        results = infer(frames, [o for _, o, _ in chunk])
        for (_, _, labels), res in zip(chunk, results):
            if res is None:
                continue
            H, proposals = res
            if len(labels) != H:
                continue          # skeletons without joints: GT list and head list no longer align
            n_data += 1
            est = []
            for h in range(H):
                idx = len(proposals)
                for p, members in enumerate(proposals):
                    if h in members:
                        idx = p
                        break
                est.append(idx)
            tot['rand score'] += adjusted_rand_score(labels, est)
            hom, com, v = homogeneity_completeness_v_measure(labels, est)
            tot['homogeneity'] += hom
            tot['completeness'] += com
            tot['v_measure'] += v
    out = {k: v / max(1, n_data) for k, v in tot.items()}
    for k in ('rand score', 'homogeneity', 'completeness', 'v_measure'):
        print(k, out[k])
    out['n_data'] = n_data
    return out


def run(args):
    from .common import max_skeletons_per_camera
    calib = Calibration(parameters)
    work = collect_work(args, calib)
    eng = Engine(parameters, calib, max_frames=args.batch,
                 max_persons_per_camera=max(4, args.persons + 1, max_skeletons_per_camera([(f, None, None) for f, _ in work])))
    load_models(eng, args, need_mlp=False)

    def infer(frames, owners):
        db = eng.to_device(eng.pack(frames))
        if args.teacher_scores and owners[0] is not None:
            persons, n_persons = eng.cluster(db, teacher_scores(db, owners))
        else:
            _, persons, n_persons = eng.match(db, want_scores=False)
        eng.sync_status()
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        out = []
        for f in range(len(frames)):
            h0, H, e0, M = db.host.frame_counts(f)
            if M == 0:
                out.append(None)
            else:
                out.append((H, [[int(h) for h in persons[f, p] if h >= 0] for p in range(int(n_persons[f]))]))
        return out

    out = evaluate(work, infer, args.batch)
    eng.close()
    return out


def main(argv=None):
    return run(build_parser('Print clustering metrics of the skeleton-matching model (CMU Panoptic only)').parse_args(argv))


if __name__ == '__main__':
    main()

"""Counterpart of the reference's test/sm_metrics_without_gt.py: clustering quality of the
skeleton-matching stage on scenes COMPOSED from single-person recordings, where the grouping is
known by construction (a skeleton belongs to the file it came from), so no 3D ground truth is
needed.

Kept from the reference (sm_metrics_without_gt.py): the CLI (--testfiles, one file per individual;
--modelsdir; --datastep), one label per head node = index of the proposal that contains it (or the
number of proposals for unassigned heads, :133-141), the labels of the true grouping obtained by
running the same proposal function on the labels-as-scores (:144-157; here: the file of origin
directly, which is what that call reconstructs), adjusted Rand index / homogeneity / completeness /
V-measure averaged over the scenes (:159-168).

NOT reproduced, and why: the reference composes its scenes with
MergedMultipleHumansDataset(mode='test_generated'), i.e. the TRAINING graph synthesis
(graph_generator.py:682-811: scenes sampled with Python's unseeded `random`, one edge-node per
ORDERED head pair, true / false / spurious blocks in that order).  That generator is outside the
inference path this package implements; its output is also not repeatable run to
run.  Here scene i merges frame i of every file into one frame of the inference-time topology
(graph_generator.py:813-876), deterministically; the numbers are therefore comparable in meaning
with the reference's, not run-for-run identical.
"""
import json

from ..calibration import Calibration
from ..parameters import parameters
from ..pipeline import Engine
from . import sm_metrics
from .common import build_parser, load_models, max_skeletons_per_camera


def compose_scenes(files, datastep):
    """[(frame, labels)]: scene i = frame i*datastep of every file merged camera by camera; labels =
    file index of every skeleton in (camera, list) order."""
    data = [json.load(open(f, 'rb')) for f in files]
    n = min(len(d) for d in data)
    scenes = []
    for i in range(0, n, datastep):
        frame, labels = {}, []
        for cam in parameters.used_cameras_skeleton_matching:
            skeletons = []
            for k, d in enumerate(data):
                if cam in d[i]:
                    for sk in json.loads(d[i][cam][0]):
                        if any(key != 'ID' for key in sk):
                            skeletons.append(sk)
                            labels.append(k)
            if skeletons:
                frame[cam] = [json.dumps(skeletons), 0]
        scenes.append((frame, labels))
    return scenes


def evaluate(scenes, infer, batch=256):
    """Same bookkeeping as sm_metrics.evaluate with the labels given by construction."""
    from sklearn.metrics import adjusted_rand_score, homogeneity_completeness_v_measure
    tot = {'rand score': 0.0, 'homogeneity': 0.0, 'completeness': 0.0, 'v_measure': 0.0}
    n_data = 0
    for start in range(0, len(scenes), batch):
        chunk = scenes[start:start + batch]
        for (frame, labels), res in zip(chunk, infer([f for f, _ in chunk], [None] * len(chunk))):
            if res is None or res[0] != len(labels):
                continue
            H, proposals = res
            est = []
            for h in range(H):
                idx = len(proposals)
                for p, members in enumerate(proposals):
                    if h in members:
                        idx = p
                        break
                est.append(idx)
            n_data += 1
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
    calib = Calibration(parameters)
    scenes = compose_scenes(args.testfiles, args.datastep)
    eng = Engine(parameters, calib, max_frames=args.batch,
                 max_persons_per_camera=max(4, max_skeletons_per_camera([(f, None, None) for f, _ in scenes])))
    load_models(eng, args, need_mlp=False)

    def infer(frames, owners):
        db = eng.to_device(eng.pack(frames))
        _, persons, n_persons = eng.match(db, want_scores=False)
        eng.sync_status()
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        out = []
        for f in range(len(frames)):
            h0, H, e0, M = db.host.frame_counts(f)
            out.append(None if M == 0 else (H, [[int(h) for h in persons[f, p] if h >= 0] for p in range(int(n_persons[f]))]))
        return out

    out = evaluate(scenes, infer, args.batch)
    eng.close()
    return out


def main(argv=None):
    return run(build_parser('Print metrics of the skeleton-matching model (ground truth is not required)').parse_args(argv))


if __name__ == '__main__':
    main()

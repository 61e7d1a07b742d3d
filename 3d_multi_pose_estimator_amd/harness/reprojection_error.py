"""Counterpart of the reference's test/reprojection_error.py: a ground-truth-free accuracy
check.  Every 3D result (MLP regression and triangulation) is projected back into each camera
that saw the person with the reference's radial-only lens model (`get_projected_coordinates`,
reprojection_error.py:89-107; `apply_distortion`, pose_estimator_utils.py:44-50) and compared
with the detected 2D joints whose confidence exceeds 0.5 (:333-343, 377-390); the mean and the
median pixel error are printed per camera (:422-430).  Matching and both 3D stages run batched
on the device; the projection is host-side numpy (it is not on the hot path).
"""
import json

import numpy as np
import torch

from .. import synthetic
from ..calibration import Calibration
from ..parameters import parameters
from ..pipeline import Engine
from .common import build_parser, load_models, teacher_scores


def project(calib, cam_idx, p3d):
    """world point (3,) -> pixel (2,), radial distortion k1,k2,k3 only, fp32 like the reference."""
    T = calib.T_d[cam_idx].astype(np.float32)
    pc = (T @ np.append(p3d.astype(np.float32), np.float32(1.0)))[:3]
    h = pc / pc[2]
    r = np.float32(h[0] * h[0] + h[1] * h[1])
    ci = calib.params.cameras[cam_idx]
    kd = np.array([calib.params.kd0[ci], calib.params.kd1[ci], calib.params.kd2[ci]], np.float32)
    f = np.float32(1) + kd[0] * r + kd[1] * r * r + kd[2] * r * r * r
    d = np.array([h[0] * f, h[1] * f, np.float32(1.0)], np.float32)
    px = calib.K32[cam_idx] @ d
    return (px / px[2])[:2]


def collect_work(args, calib):
    work = []
    if args.synthetic:
        spec = synthetic.FrameSpec(persons=args.persons, noise_px=args.noise_px, float_conf=False)
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
                work.append((frame, None))
    return work


def evaluate(work, infer, calib, batch=256):
    """`infer(frames, owners)` -> per frame a list of persons, each
    ({camera: skeleton dict}, est pose [J,3] or None, {joint idx: (3,)} triangulated).
    Bookkeeping of reprojection_error.py:350-420: per camera that saw the person, every used joint
    of the estimate / every triangulated joint whose detection has valid > 0.5."""
    names = list(parameters.camera_names)
    err = {'est': {c: [] for c in names}, 'triang': {c: [] for c in names}}
    for start in range(0, len(work), batch):
        chunk = work[start:start + batch]
        frames = [{c: [f[c][0], f[c][1]] for c in f if json.loads(f[c][0])} for f, _ in chunk]
        for persons in infer(frames, [o for _, o in chunk]):
            for skels, est, tri in persons or []:
                for cam, coords in skels.items():
                    c = names.index(cam)
                    for j in parameters.joint_list:
                        key = str(j)
                        if key not in coords or not coords[key][3] > 0.5:
                            continue
                        obs = np.array([coords[key][1], coords[key][2]])
                        if est is not None and j in parameters.used_joints:
                            err['est'][cam].append(float(np.linalg.norm(project(calib, c, est[j]) - obs)))
                        if j in tri:
                            err['triang'][cam].append(float(np.linalg.norm(project(calib, c, tri[j]) - obs)))
    print('**********************  REPROJECTION ERRORS (mean and median) **********************')
    out = {}
    for cam in names:
        print('------------------', 'CAMERA', cam, '------------------')
        for kind in ('est', 'triang'):
            if err[kind][cam]:
                a = np.array(err[kind][cam])
                print(kind, a.mean(), np.median(a))
                out[(kind, cam)] = (float(a.mean()), float(np.median(a)), len(a))
    return out


def run(args):
    from .common import max_skeletons_per_camera
    calib = Calibration(parameters)
    work = collect_work(args, calib)
    eng = Engine(parameters, calib, max_frames=args.batch,
                 max_persons_per_camera=max(4, args.persons + 1, max_skeletons_per_camera([(f, None, None) for f, _ in work])))
    load_models(eng, args, need_mlp=True)
    names = list(parameters.camera_names)

    def infer(frames, owners):
        db = eng.to_device(eng.pack(frames, keep_json=True))
        if args.teacher_scores and owners[0] is not None:
            persons, n_persons = eng.cluster(db, teacher_scores(db, owners))
        else:
            _, persons, n_persons = eng.match(db, want_scores=False)
        poses, valid = eng.mlp3d(db, persons, n_persons)
        # the script's own gather keeps joints whose id is > 0 (reprojection_error.py:296-300)
        tri, jv = eng.triangulate(db, persons, n_persons, all_joints=True, positive_ids_only=True)
        eng.sync_status()
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        poses, valid, tri, jv = poses.cpu().numpy(), valid.cpu().numpy(), tri.cpu().numpy(), jv.cpu().numpy()
        out = []
        for f in range(len(frames)):
            heads = db.host.jsons_for_head[f]
            people = []
            for p in range(int(n_persons[f])):
                skels = {cam: heads[int(persons[f, p, c])] for c, cam in enumerate(names) if persons[f, p, c] >= 0}
                people.append((skels, poses[f, p] if valid[f, p] else None,
                               {j: tri[f, p, j].astype(np.float32) for j in parameters.joint_list if jv[f, p, j]}))
            out.append(people)
        return out

    out = evaluate(work, infer, calib, args.batch)
    eng.close()
    return out


def main(argv=None):
    return run(build_parser('Print the reprojection error of the pose estimation model and of the triangulation').parse_args(argv))


if __name__ == '__main__':
    main()

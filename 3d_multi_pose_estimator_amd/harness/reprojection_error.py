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


def run(args):
    calib = Calibration(parameters)
    eng = Engine(parameters, calib, max_frames=args.batch, max_persons_per_camera=max(4, args.persons + 1))
    load_models(eng, args, need_mlp=True)
    names = list(parameters.camera_names)
    work = []
    if args.synthetic:
        spec = synthetic.FrameSpec(persons=args.persons, noise_px=args.noise_px, float_conf=False)
        for i in range(args.synthetic):
            f, gt = synthetic.make_frame(calib, i, spec)
            work.append((f, gt['owner']))
    else:
        n_input = 0
        for file in args.testfiles:
            print(file)
            for frame in json.load(open(file, 'rb')):
                n_input += 1
                if (n_input - 1) % args.datastep == 0:
                    work.append((frame, None))
    err = {'est': {c: [] for c in names}, 'triang': {c: [] for c in names}}
    for start in range(0, len(work), args.batch):
        chunk = work[start:start + args.batch]
        frames = [{c: [f[c][0], f[c][1]] for c in f if json.loads(f[c][0])} for f, _ in chunk]
        db = eng.to_device(eng.pack(frames, keep_json=True))
        if args.teacher_scores and chunk[0][1] is not None:
            persons, n_persons = eng.cluster(db, teacher_scores(db, [o for _, o in chunk]))
        else:
            _, persons, n_persons = eng.match(db, want_scores=False)
        poses, valid = eng.mlp3d(db, persons, n_persons)
        tri, jv = eng.triangulate(db, persons, n_persons, all_joints=True)
        persons, n_persons = persons.cpu().numpy(), n_persons.cpu().numpy()
        poses, valid, tri, jv = poses.cpu().numpy(), valid.cpu().numpy(), tri.cpu().numpy(), jv.cpu().numpy()
        for f in range(len(chunk)):
            heads = db.host.jsons_for_head[f]
            for p in range(int(n_persons[f])):
                for c, cam in enumerate(names):
                    h = persons[f, p, c]
                    if h < 0:
                        continue
                    coords = heads[int(h)]
                    for j in parameters.joint_list:
                        key = str(j)
                        if key not in coords or not coords[key][3] > 0.5:
                            continue
                        obs = np.array([coords[key][1], coords[key][2]])
                        if valid[f, p] and j in parameters.used_joints:
                            err['est'][cam].append(float(np.linalg.norm(project(calib, c, poses[f, p, j]) - obs)))
                        if jv[f, p, j]:
                            err['triang'][cam].append(float(np.linalg.norm(project(calib, c, tri[f, p, j]) - obs)))
    print('**********************  REPROJECTION ERRORS (mean and median) **********************')
    out = {}
    for cam in names:
        print('------------------', 'CAMERA', cam, '------------------')
        for kind in ('est', 'triang'):
            if err[kind][cam]:
                a = np.array(err[kind][cam])
                print(kind, a.mean(), np.median(a))
                out[(kind, cam)] = (float(a.mean()), float(np.median(a)))
    eng.close()
    return out


def main(argv=None):
    return run(build_parser('Print the reprojection error of the pose estimation model and of the triangulation').parse_args(argv))


if __name__ == '__main__':
    main()

"""API mirror of the reference's utils/pose_estimator_utils.py (``camera_matrix`` :17-30,
``from_homogeneous`` / ``from_homogeneous2`` :32-36, ``get_distortion_coefficients`` :39-41,
``apply_distortion`` :44-50, ``triangulate`` :52-75).  The undistortion, the pairwise DLT solves
and the 5 cm median filter run in the HIP kernel k_triangulate (mpe_triangulate_batch) on a
one-person batch; the projection helpers are the host-side tensor functions the reference's
reprojection check uses (test/reprojection_error.py:89-107) and are not on the hot path."""
import numpy as np
import torch

from . import runtime
from .packing import PackedBatch
from .parameters import parameters


def camera_matrix(cam_idx, use_cuda=True):
    dev = torch.device('cuda') if (torch.cuda.is_available() and use_cuda) else torch.device('cpu')
    return torch.tensor([[parameters.fx[cam_idx], 0.0, parameters.cx[cam_idx]],
                         [0.0, parameters.fy[cam_idx], parameters.cy[cam_idx]],
                         [0.0, 0.0, 1.0]], device=dev)


def from_homogeneous(v):
    return (v / v[-1])[:-1]


def from_homogeneous2(v):
    return v / v[-1]


def get_distortion_coefficients(cam_idx):
    """Radial coefficients k1, k2, k3 of one camera (the tangential ones are ignored here,
    as in the reference)."""
    dev = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
    return torch.tensor([parameters.kd0[cam_idx], parameters.kd1[cam_idx], parameters.kd2[cam_idx]], device=dev)


def apply_distortion(kd, v):
    """Radial-only lens model on normalised homogeneous points v [3, n]: x,y scaled by
    1 + k1 r^2 + k2 r^4 + k3 r^6 with r = |(x, y)|."""
    v2 = v.clone()
    r = torch.norm(v[:-1][:], dim=0)
    r = r * r
    f = 1 + kd[0] * r + kd[1] * r * r + kd[2] * r * r * r
    v2[0][:] = v[0][:] * f
    v2[1][:] = v[1][:] * f
    return v2


def _one_person_batch(points_2D, params):
    """points_2D[joint str][cam] = (x, y)  ->  1-frame batch with one head per camera."""
    names = list(params.camera_names)
    J = len(params.joint_list)
    cams = [c for c in names if any(c in v for v in points_2D.values())]
    pb = PackedBatch(len(names), J)
    pb.n_frames = 1
    n = len(cams)
    pb.slot_cam = np.full((1, len(names)), -1, np.int32)
    pb.slot_n = np.zeros((1, len(names)), np.int32)
    pb.head_cam = np.zeros(n, np.int32)
    pb.joint_mask = np.zeros(n, np.uint32)
    pb.xy = np.zeros((n, J, 2), np.float64)
    pb.vp = np.ones((n, J, 2), np.float32)
    for h, c in enumerate(cams):
        pb.slot_cam[0, h] = names.index(c)
        pb.slot_n[0, h] = 1
        pb.head_cam[h] = names.index(c)
        for j in params.joint_list:
            pt = points_2D.get(str(j), {}).get(c)
            if pt is not None:
                pb.joint_mask[h] |= np.uint32(1 << j)
                pb.xy[h, j] = np.asarray(pt, np.float64).reshape(2)
    pb.tri_mask = pb.joint_mask.copy()
    pb.frame_head_off = np.array([0, n], np.int32)
    pb.frame_en_off = np.array([0, n * (n - 1) // 2], np.int32)
    pb.skeleton_index = np.zeros(n, np.int32)
    person = np.full((1, len(names)), -1, np.int32)
    for h, c in enumerate(cams):
        person[0, names.index(c)] = h
    return pb, person


def triangulate(points_2D, camera_matrices, distortion_coefficients, projection_matrices, median_chek_axis):
    """Same arguments and result as the reference: ``{joint str: ndarray (3,1)}`` for joints
    seen by at least two cameras.  Pairs are formed in ``parameters.camera_names`` order (the
    order the callers insert them, metrics_from_triangulation.py:237-247)."""
    calib = runtime.calibration_from_dicts(parameters, camera_matrices, distortion_coefficients,
                                           projection_matrices)
    if median_chek_axis != parameters.axes_3D['Y'][0]:
        raise NotImplementedError('median axis is fixed by parameters.axes_3D (reference callers pass that)')
    eng = runtime.shared_engine(parameters, calib)
    pb, person = _one_person_batch(points_2D, parameters)
    db = eng.to_device(pb)
    persons = torch.full((1, eng.pcap, eng.V), -1, dtype=torch.int32)
    persons[0, 0] = torch.from_numpy(person[0])
    n_persons = torch.ones(1, dtype=torch.int32)
    # all joints, including those outside used_joints, are returned by the reference function
    # itself (the used_joints filter lives in the caller), so ask for every joint
    poses, jv = eng.triangulate(db, persons.to(eng.device), n_persons.to(eng.device), all_joints=True)
    poses, jv = poses[0, 0].cpu().numpy(), jv[0, 0].cpu().numpy()
    return {str(j): poses[j].reshape(3, 1) for j in parameters.joint_list if jv[j]}

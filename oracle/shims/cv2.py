"""Stand-in for the two OpenCV calls on the path (pose_estimator_utils.py:63-67,
pose_estimator_dataset_from_json.py:92-96,261), restating OpenCV's published
algorithms in float64:

  undistortPoints(src, K, dist): x=(u-cx)/fx, y=(v-cy)/fy, then 5 fixed-point
    iterations of the k1,k2,p1,p2,k3 model (cvUndistortPointsInternal, default
    TermCriteria(COUNT, 5)); no R/P, so normalised coordinates come back, shape (N,1,2).
  triangulatePoints(P1, P2, x1, x2): per point the 4x4 DLT matrix with rows
    x*P[2]-P[0], y*P[2]-P[1] for both views; the solution is the right singular vector
    of the smallest singular value (homogeneous, 4xN).
"""
import numpy as np


def undistortPoints(src, cameraMatrix, distCoeffs, R=None, P=None):
    assert R is None and P is None
    pts = np.asarray(src, dtype=np.float64).reshape(-1, 2)
    K = np.asarray(cameraMatrix, dtype=np.float64)
    k = np.zeros(14, np.float64)
    dc = np.asarray(distCoeffs, dtype=np.float64).reshape(-1)
    k[:dc.shape[0]] = dc
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    out = np.zeros((pts.shape[0], 1, 2), np.float64)
    for i in range(pts.shape[0]):
        u, v = float(pts[i, 0]), float(pts[i, 1])
        x = (u - cx) * ifx
        y = (v - cy) * ify
        x0, y0 = x, y
        for _ in range(5):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            if icdist < 0:
                x = (u - cx) * ifx
                y = (v - cy) * ify
                break
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
        out[i, 0, 0] = x
        out[i, 0, 1] = y
    return out


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2):
    P = [np.asarray(projMatr1, np.float64), np.asarray(projMatr2, np.float64)]
    p1 = np.asarray(projPoints1, np.float64)
    p2 = np.asarray(projPoints2, np.float64)

    def as2xn(p):
        if p.ndim == 3:                      # (N,1,2) / (1,N,2): 2-channel vector of points
            return p.reshape(-1, 2).T
        if p.shape[0] == 2:
            return p
        return p.reshape(-1, 2).T
    x = [as2xn(p1), as2xn(p2)]
    n = x[0].shape[1]
    out = np.zeros((4, n), np.float64)
    for i in range(n):
        A = np.zeros((4, 4), np.float64)
        for j in range(2):
            A[2 * j + 0] = x[j][0, i] * P[j][2] - P[j][0]
            A[2 * j + 1] = x[j][1, i] * P[j][2] - P[j][1]
        _, _, vt = np.linalg.svd(A)
        out[:, i] = vt[3]
    return out

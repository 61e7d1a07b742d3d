// The two OpenCV calls of the 3D stages restated in f64 (pose3d.hip has the citations): cv2.undistortPoints (five fixed-point iterations
// of the k1, k2, p1, p2, k3 model) and cv2.triangulatePoints (4 x 4 DLT system, right singular vector of the smallest singular value by
// a one-sided Jacobi SVD in registers).  One definition for every kernel that solves pairs (pose3d.hip: the row and triangulation
// kernels; cluster.hip: the small-batch tail launch that solves every cross-camera pair of a frame beside the clustering), so that a
// pair gives the same bits wherever it was solved.
#pragma once
#include "mpe_internal.h"

namespace mpe {
namespace dltc {

__device__ inline void undistort_point(const DevCfg *cfg, int cam, double u, double v, double *ox, double *oy) {
#pragma clang fp contract(off)
    const float *K = cfg->K[cam];
    const double fx = (double)K[0], fy = (double)K[4], cx = (double)K[2], cy = (double)K[5];
    const double *d = cfg->dist[cam];
    const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int it = 0; it < 5; ++it) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
        if (icdist < 0) {
            x = x0;
            y = y0;
            break;
        }
        const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    *ox = x;
    *oy = y;
}

// Right singular vector of the smallest singular value of the 4x4 DLT matrix, dehomogenised.
__device__ inline void dlt_solve(const double *P1, const double *P2, double x1, double y1, double x2, double y2,
                                 double *out) {
#pragma clang fp contract(fast)
    double A[4][4], Vm[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0][k] = x1 * P1[8 + k] - P1[k];
        A[1][k] = y1 * P1[8 + k] - P1[4 + k];
        A[2][k] = x2 * P2[8 + k] - P2[k];
        A[3][k] = y2 * P2[8 + k] - P2[4 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) Vm[k][j] = (k == j) ? 1.0 : 0.0;
    }
    // converged when every column pair is orthogonal to a few ulp: |<a_p,a_q>| <= 4e-16 |a_p||a_q|
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                if (ga * ga <= 1.6e-31 * (al * be) || ga == 0.0) continue;
                rotated = true;
                // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with zeta = (be - al) / (2 ga), written with one
                // division and one square root; c = 1 / sqrt(1 + t^2)
                const double d = be - al, g2 = 2.0 * fabs(ga);
                const double sg = (d == 0.0 || (d > 0.0) == (ga > 0.0)) ? 1.0 : -1.0;
                const double t = sg * g2 / (fabs(d) + sqrt(d * d + g2 * g2));
                const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = Vm[i][p], vq = Vm[i][q];
                    Vm[i][p] = c * vp - s * vq;
                    Vm[i][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int jm = 0;
    double best = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) nn += A[i][j] * A[i][j];
        if (j == 0 || nn < best) {
            best = nn;
            jm = j;
        }
    }
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j == jm) {
            v0 = Vm[0][j];
            v1 = Vm[1][j];
            v2 = Vm[2][j];
            v3 = Vm[3][j];
        }
    out[0] = v0 / v3;
    out[1] = v1 / v3;
    out[2] = v2 / v3;
}

__device__ inline int pair_index(int c1, int c2, int V) {   // lexicographic index of (c1<c2)
    return c1 * V - c1 * (c1 + 1) / 2 + (c2 - c1 - 1);
}


}  // namespace dltc
}  // namespace mpe

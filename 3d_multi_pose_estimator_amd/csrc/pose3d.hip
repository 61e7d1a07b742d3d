// 3D stage: per-person featurisation for the MLP, pairwise DLT triangulation, median filter.
//
// Replaces (reference, relative to /root/reference):
//   utils/pose_estimator_dataset_from_json.py:63-101   get_3D_from_triangulation
//   utils/pose_estimator_dataset_from_json.py:237-298  PoseEstimatorDataset, dict branch
//   utils/pose_estimator_utils.py:52-75                triangulate (+ caller gather,
//                                                      test/metrics_from_triangulation.py:234-272)
// and the two OpenCV calls they make, restated from OpenCV's published algorithms in f64:
//   cv2.undistortPoints  : normalise with K, 5 fixed-point iterations of the k1,k2,p1,p2,k3 model
//   cv2.triangulatePoints: 4x4 DLT system, right singular vector of the smallest singular
//                          value (here: one-sided Jacobi SVD in registers).
// One workgroup per (frame, person); pair solves are spread over the lanes, sums run in the
// reference's pair order.  f64 arithmetic is kept un-contracted (no fma fusion) to stay as
// close as possible to the CPU evaluation order, except inside the Jacobi SVD (an iteration to
// convergence whose result does not depend on the rounding of single steps beyond a few ulp).
#include "mpe_internal.h"
#include "dlt_common.h"

namespace mpe {

using namespace dltc;

// ---------------------------------------------------------------------------------------
// shared front end: heads of the person, undistorted points, all pair solves into LDS
// ---------------------------------------------------------------------------------------
struct PersonCtx {
    int f, p, V, J, npairs;
    int h0;
};

// s_und [V][J][2], s_pts [J][npairs][3], s_head[V] (global head index or -1)
// pair_pts (small batches, cluster.hip: k_lat_tail): every cross-camera pair of the frame has ALREADY been solved, beside the
// clustering -- [global head of the lower camera][frame-local head of the other][J][3] -- and a pair is fetched instead of solved
// (the same function on the same arguments: the same bits).
__device__ inline void person_front(const DevCfg *cfg, const PersonCtx &pc, const int32_t *s_head,
                                    const uint32_t *s_mask, const double *__restrict__ xy, double *s_und,
                                    double *s_pts, const double *__restrict__ pair_pts = nullptr, int pair_hmax = 0) {
    const int V = pc.V, J = pc.J;
    for (int i = threadIdx.x; i < V * J; i += blockDim.x) {
        const int c = i / J, j = i - c * J;
        const int gh = s_head[c];
        if (gh >= 0 && (s_mask[c] >> j & 1u)) {
            double ux, uy;
            undistort_point(cfg, c, xy[((size_t)gh * J + j) * 2], xy[((size_t)gh * J + j) * 2 + 1], &ux, &uy);
            s_und[i * 2] = ux;
            s_und[i * 2 + 1] = uy;
        }
    }
    __syncthreads();
    const int np = pc.npairs;
    for (int i = threadIdx.x; i < J * np; i += blockDim.x) {
        const int j = i / np, pi = i - j * np;
        // decode pair index -> (c1, c2)
        int c1 = 0, rem = pi;
        while (rem >= V - 1 - c1) { rem -= V - 1 - c1; ++c1; }
        const int c2 = c1 + 1 + rem;
        if (s_head[c1] >= 0 && s_head[c2] >= 0 && (s_mask[c1] >> j & 1u) && (s_mask[c2] >> j & 1u)) {
            if (pair_pts) {
                const double *src = pair_pts + (((size_t)s_head[c1] * pair_hmax + (s_head[c2] - pc.h0)) * J + j) * 3;
                s_pts[(size_t)i * 3] = src[0];
                s_pts[(size_t)i * 3 + 1] = src[1];
                s_pts[(size_t)i * 3 + 2] = src[2];
            } else {
                dlt_solve(cfg->P[c1], cfg->P[c2], s_und[(c1 * J + j) * 2], s_und[(c1 * J + j) * 2 + 1],
                          s_und[(c2 * J + j) * 2], s_und[(c2 * J + j) * 2 + 1], s_pts + (size_t)i * 3);
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// MLP input rows
// ---------------------------------------------------------------------------------------
// NT = 256 threads: the 180 pair solves of a 5-camera person (18 joints x 10 pairs, f64 Jacobi SVD
// each) run in one round; NT = 1024 for rigs whose pair tables leave room for one workgroup per CU
// only (23 cameras: 4554 pair solves per person, 116 KB of LDS)
template <int NT>
__global__ __launch_bounds__(NT) void k_mlp_rows(const DevCfg *__restrict__ cfg, int pcap,
                                                  const int32_t *__restrict__ head_off,
                                                  const uint32_t *__restrict__ joint_mask,
                                                  const uint32_t *__restrict__ tri_mask,
                                                  const double *__restrict__ xy, const float *__restrict__ vp,
                                                  const int32_t *__restrict__ persons,
                                                  const int32_t *__restrict__ n_persons,
                                                  const int32_t *__restrict__ person_off, float *__restrict__ rows,
                                                  int ld_rows, uint8_t *__restrict__ valid,
                                                  int32_t *__restrict__ scan_out, int32_t *__restrict__ total_out, int n_frames,
                                                  float *__restrict__ zero_poses, int n_out,
                                                  const double *__restrict__ pair_pts, int pair_hmax) {
#pragma clang fp contract(off)
    extern __shared__ double s_dyn64[];
    const int f = blockIdx.x / pcap, p = blockIdx.x - f * pcap;
    const int V = cfg->V, J = cfg->J, npj = cfg->npj;
    const int npairs = V * (V - 1) / 2;
    double *s_und = s_dyn64;                       // [V][J][2]
    double *s_pts = s_und + (size_t)V * J * 2;      // [J][npairs][3]
    float *s_row = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(s_pts + (size_t)J * npairs * 3) + 15) & ~(uintptr_t)15);   // [ld_rows], 16-byte aligned: the row is assembled here
    __shared__ int32_t s_head[MPE_MAX_CAMERAS];
    __shared__ uint32_t s_mask[MPE_MAX_CAMERAS], s_tmask[MPE_MAX_CAMERAS];
    __shared__ float s_red[NT];
    const int np_f = n_persons[f];
    const size_t slot = (size_t)f * pcap + p;
    // scan_out (small batches): the exclusive prefix of the persons per frame is computed HERE (k_person_scan is a launch of its own
    // for a handful of numbers); the first workgroup of a frame writes the frame's entry for the kernels behind this one
    __shared__ int s_off;
    if (scan_out) {
        if (threadIdx.x == 0) {
            int acc = 0;
            for (int g = 0; g < f; ++g) acc += min(max(n_persons[g], 0), pcap);
            s_off = acc;
            if (p == 0) {
                scan_out[f] = acc;
                if (f == n_frames - 1) {
                    const int tot = acc + min(max(np_f, 0), pcap);
                    scan_out[n_frames] = tot;
                    if (total_out) *total_out = tot;
                }
            }
        }
        __syncthreads();
    }
    if (p >= np_f) {
        if (threadIdx.x == 0 && valid) valid[slot] = 0;
        if (zero_poses)                      // the pose slots nobody fills (k_decode's job when it runs)
            for (int k = threadIdx.x; k < n_out; k += blockDim.x) zero_poses[slot * n_out + k] = 0.f;
        return;
    }
    const size_t r = scan_out ? (size_t)s_off + p : person_off ? (size_t)person_off[f] + p : slot;
    const int h0 = head_off[f];
    if (threadIdx.x < V) {
        const int lh = persons[slot * V + threadIdx.x];
        s_head[threadIdx.x] = lh >= 0 ? h0 + lh : -1;
        s_mask[threadIdx.x] = lh >= 0 ? joint_mask[h0 + lh] : 0u;
        // get_3D_from_triangulation only uses joints whose values[0] > 0 (reference :75)
        s_tmask[threadIdx.x] = lh >= 0 ? tri_mask[h0 + lh] : 0u;
    }
    float *row = s_row;                              // written once, coalesced, at the end
    for (int c = threadIdx.x; c < ld_rows; c += blockDim.x) row[c] = 0.f;
    __syncthreads();
    PersonCtx pc{f, p, V, J, npairs, h0};
    person_front(cfg, pc, s_head, s_tmask, xy, s_und, s_pts, pair_pts, pair_hmax);   // pair solves on the tri mask (or the pairs solved beside the clustering)
    // undistorted points of joints outside the tri mask (joint 0) are still needed for the rays
    for (int i = threadIdx.x; i < V * J; i += blockDim.x) {
        const int c = i / J, j = i - c * J;
        const int gh = s_head[c];
        if (gh >= 0 && (s_mask[c] >> j & 1u) && !(s_tmask[c] >> j & 1u)) {
            double ux, uy;
            undistort_point(cfg, c, xy[((size_t)gh * J + j) * 2], xy[((size_t)gh * J + j) * 2 + 1], &ux, &uy);
            s_und[i * 2] = ux;
            s_und[i * 2 + 1] = uy;
        }
    }
    __syncthreads();
    // per (camera, joint) numbers 0..9
    for (int i = threadIdx.x; i < V * J; i += blockDim.x) {
        const int c = i / J, j = i - c * J;
        const int gh = s_head[c];
        if (gh < 0 || !(s_mask[c] >> j & 1u)) continue;
        float *o = row + (size_t)c * J * npj + j * npj;
        const double x = xy[((size_t)gh * J + j) * 2], y = xy[((size_t)gh * J + j) * 2 + 1];
        const double hw = cfg->W / 2.0, hh = cfg->H / 2.0;
        o[0] = vp[((size_t)gh * J + j) * 2];
        o[1] = (float)((x - hw) / hw);
        o[2] = (float)((y - hh) / hh);
        o[3] = vp[((size_t)gh * J + j) * 2 + 1];
        const float *T = cfg->T_i[c];
        const float uf = (float)s_und[i * 2], vf = (float)s_und[i * 2 + 1];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            o[4 + rr] = T[rr * 4 + 3] / 10.f;
            float a = T[rr * 4 + 0] * uf;
            a = __builtin_fmaf(T[rr * 4 + 1], vf, a);
            a = __builtin_fmaf(T[rr * 4 + 2], 1.0f, a);
            o[7 + rr] = a / 10.f;
        }
    }
    // triangulated joints: mean over camera pairs in pair order, written into every camera block
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        double ax = 0, ay = 0, az = 0;
        int n = 0;
        for (int c1 = 0; c1 < V; ++c1) {
            if (s_head[c1] < 0 || !(s_tmask[c1] >> j & 1u)) continue;
            for (int c2 = c1 + 1; c2 < V; ++c2) {
                if (s_head[c2] < 0 || !(s_tmask[c2] >> j & 1u)) continue;
                const double *pt = s_pts + ((size_t)j * npairs + pair_index(c1, c2, V)) * 3;
                ax += pt[0];
                ay += pt[1];
                az += pt[2];
                ++n;
            }
        }
        if (n > 0) {
            const float tx = (float)((ax / n) / 10.), ty = (float)((ay / n) / 10.), tz = (float)((az / n) / 10.);
            for (int c = 0; c < V; ++c) {
                float *o = row + (size_t)c * J * npj + j * npj;
                o[10] = 1.f;
                o[11] = tx;
                o[12] = ty;
                o[13] = tz;
            }
        }
    }
    __syncthreads();
    // sample kept only if sum(|row|) > 1 (reference :287)
    float acc = 0.f;
    const int width = V * J * npj;
    for (int c = threadIdx.x; c < width; c += blockDim.x) acc += fabsf(row[c]);
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_red[threadIdx.x] += s_red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && valid) valid[slot] = s_red[0] > 1.f ? 1 : 0;
    float *dst = rows + r * ld_rows;
    if ((ld_rows & 3) == 0) {
        for (int c = threadIdx.x * 4; c < ld_rows; c += blockDim.x * 4)
            *reinterpret_cast<float4 *>(dst + c) = *reinterpret_cast<const float4 *>(row + c);
    } else {
        for (int c = threadIdx.x; c < ld_rows; c += blockDim.x) dst[c] = row[c];
    }
}

// exclusive prefix of n_persons (single workgroup) + total
__global__ __launch_bounds__(1024) void k_person_scan(int n_frames, int pcap, const int32_t *__restrict__ n_persons,
                                                      int32_t *__restrict__ person_off,
                                                      int32_t *__restrict__ total) {
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int chunk = (n_frames + 1023) / 1024;
    const int lo = t * chunk, hi = min(n_frames, lo + chunk);
    int acc = 0;
    for (int i = lo; i < hi; ++i) acc += min(max(n_persons[i], 0), pcap);
    s_part[t] = acc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = t >= off ? s_part[t - off] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int base = t ? s_part[t - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        person_off[i] = base;
        base += min(max(n_persons[i], 0), pcap);
    }
    if (t == 1023) {
        person_off[n_frames] = s_part[1023];
        *total = s_part[1023];
    }
}

hipError_t launch_person_scan(hipStream_t s, int n_frames, int pcap, const int32_t *n_persons, int32_t *person_off,
                              int32_t *total) {
    hipLaunchKernelGGL(k_person_scan, dim3(1), dim3(1024), 0, s, n_frames, pcap, n_persons, person_off, total);
    return hipGetLastError();
}

hipError_t launch_mlp_rows(hipStream_t s, const DevCfg *cfg, int V, int J, const mpe_batch &b,
                           const int32_t *persons, const int32_t *n_persons, const int32_t *person_off, int pcap,
                           float *rows, int ld_rows, uint8_t *valid, int32_t *scan_out, int32_t *total_out, float *zero_poses, int n_out,
                           const double *pair_pts, int pair_hmax) {
    if (b.n_frames <= 0) return hipSuccess;
    const size_t shm = ((size_t)V * J * 2 + (size_t)J * (V * (V - 1) / 2) * 3) * sizeof(double) + (size_t)ld_rows * sizeof(float) + 16;
    if (shm > 40 * 1024) {
        // fewer than four workgroups per CU by LDS: widen the workgroup instead
        if (shm > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_rows<1024>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_mlp_rows<1024>, dim3(b.n_frames * pcap), dim3(1024), shm, s, cfg, pcap, b.d_frame_head_off,
                           b.d_joint_mask, b.d_tri_mask, b.d_xy, b.d_vp, persons, n_persons, person_off, rows, ld_rows,
                           valid, scan_out, total_out, b.n_frames, zero_poses, n_out, pair_pts, pair_hmax);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_mlp_rows<256>, dim3(b.n_frames * pcap), dim3(256), shm, s, cfg, pcap, b.d_frame_head_off,
                       b.d_joint_mask, b.d_tri_mask, b.d_xy, b.d_vp, persons, n_persons, person_off, rows, ld_rows,
                       valid, scan_out, total_out, b.n_frames, zero_poses, n_out, pair_pts, pair_hmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// triangulation with the 5 cm median filter
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_triangulate(const DevCfg *__restrict__ cfg, int pcap,
                                                     const int32_t *__restrict__ head_off,
                                                     const uint32_t *__restrict__ joint_mask,
                                                     const double *__restrict__ xy,
                                                     const int32_t *__restrict__ persons,
                                                     const int32_t *__restrict__ n_persons,
                                                     double *__restrict__ poses, uint8_t *__restrict__ joint_valid,
                                                     uint32_t out_mask) {
#pragma clang fp contract(off)
    extern __shared__ double s_dyn64[];
    const int f = blockIdx.x / pcap, p = blockIdx.x - f * pcap;
    const int V = cfg->V, J = cfg->J;
    const int npairs = V * (V - 1) / 2;
    double *s_und = s_dyn64;
    double *s_pts = s_und + (size_t)V * J * 2;
    __shared__ int32_t s_head[MPE_MAX_CAMERAS];
    __shared__ uint32_t s_mask[MPE_MAX_CAMERAS];
    const size_t slot = (size_t)f * pcap + p;
    double *po = poses + slot * J * 3;
    uint8_t *jv = joint_valid + slot * J;
    if (p >= n_persons[f]) {
        for (int j = threadIdx.x; j < J; j += blockDim.x) {
            jv[j] = 0;
            po[j * 3] = po[j * 3 + 1] = po[j * 3 + 2] = 0.0;
        }
        return;
    }
    const int h0 = head_off[f];
    if (threadIdx.x < V) {
        const int lh = persons[slot * V + threadIdx.x];
        s_head[threadIdx.x] = lh >= 0 ? h0 + lh : -1;
        s_mask[threadIdx.x] = lh >= 0 ? joint_mask[h0 + lh] : 0u;
    }
    __syncthreads();
    PersonCtx pc{f, p, V, J, npairs, h0};
    person_front(cfg, pc, s_head, s_mask, xy, s_und, s_pts);
    const int axis = cfg->median_axis;
    const double win = (double)cfg->median_window;
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        // pairs in combination order of the cameras that see joint j
        int n = 0;
        for (int c1 = 0; c1 < V; ++c1) {
            if (s_head[c1] < 0 || !(s_mask[c1] >> j & 1u)) continue;
            for (int c2 = c1 + 1; c2 < V; ++c2)
                if (s_head[c2] >= 0 && (s_mask[c2] >> j & 1u)) ++n;
        }
        double ox = 0, oy = 0, oz = 0;
        uint8_t ok = 0;
        if (n > 0) {
            ok = 1;
            // upper median of the `axis` coordinate = element n//2 of the sorted values
            const double *base = s_pts + (size_t)j * npairs * 3;
            double med = 0;
            bool found = false;
            for (int c1 = 0; c1 < V && !found; ++c1) {
                if (s_head[c1] < 0 || !(s_mask[c1] >> j & 1u)) continue;
                for (int c2 = c1 + 1; c2 < V && !found; ++c2) {
                    if (s_head[c2] < 0 || !(s_mask[c2] >> j & 1u)) continue;
                    const int pi = pair_index(c1, c2, V);
                    const double d = base[pi * 3 + axis];
                    int rank = 0;
                    for (int e1 = 0; e1 < V; ++e1) {
                        if (s_head[e1] < 0 || !(s_mask[e1] >> j & 1u)) continue;
                        for (int e2 = e1 + 1; e2 < V; ++e2) {
                            if (s_head[e2] < 0 || !(s_mask[e2] >> j & 1u)) continue;
                            const int qi = pair_index(e1, e2, V);
                            const double dd = base[qi * 3 + axis];
                            if (dd < d || (dd == d && qi < pi)) ++rank;
                        }
                    }
                    if (rank == n / 2) {
                        med = d;
                        found = true;
                    }
                }
            }
            double sx = 0, sy = 0, sz = 0;
            int kept = 0;
            for (int c1 = 0; c1 < V; ++c1) {
                if (s_head[c1] < 0 || !(s_mask[c1] >> j & 1u)) continue;
                for (int c2 = c1 + 1; c2 < V; ++c2) {
                    if (s_head[c2] < 0 || !(s_mask[c2] >> j & 1u)) continue;
                    const double *pt = base + pair_index(c1, c2, V) * 3;
                    if (fabs(pt[axis] - med) < win) {
                        sx += pt[0];
                        sy += pt[1];
                        sz += pt[2];
                        ++kept;
                    }
                }
            }
            if (out_mask >> j & 1u) {
                ox = sx / kept;
                oy = sy / kept;
                oz = sz / kept;
            }
        }
        jv[j] = ok;
        po[j * 3] = ox;
        po[j * 3 + 1] = oy;
        po[j * 3 + 2] = oz;
    }
}

hipError_t launch_triangulate(hipStream_t s, const DevCfg *cfg, int V, int J, const mpe_batch &b,
                              const int32_t *persons, const int32_t *n_persons, int pcap, double *poses,
                              uint8_t *joint_valid, uint32_t out_mask, bool positive_ids_only) {
    if (b.n_frames <= 0) return hipSuccess;
    const size_t shm = ((size_t)V * J * 2 + (size_t)J * (V * (V - 1) / 2) * 3) * sizeof(double);
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_triangulate),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_triangulate, dim3(b.n_frames * pcap), dim3(128), shm, s, cfg, pcap, b.d_frame_head_off,
                       positive_ids_only ? b.d_tri_mask : b.d_joint_mask, b.d_xy, persons, n_persons, poses, joint_valid,
                       out_mask);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// explicit pairs (parity tests of the OpenCV restatement)
// ---------------------------------------------------------------------------------------
__global__ void k_dlt_pairs(const DevCfg *__restrict__ cfg, const double *__restrict__ pts,
                            const int32_t *__restrict__ cams, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c1 = cams[2 * i], c2 = cams[2 * i + 1];
    double u1, v1, u2, v2;
    undistort_point(cfg, c1, pts[4 * i], pts[4 * i + 1], &u1, &v1);
    undistort_point(cfg, c2, pts[4 * i + 2], pts[4 * i + 3], &u2, &v2);
    dlt_solve(cfg->P[c1], cfg->P[c2], u1, v1, u2, v2, out + 3 * (size_t)i);
}

hipError_t launch_dlt_pairs(hipStream_t s, const DevCfg *cfg, const double *pts, const int32_t *cams, int n,
                            double *out) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dlt_pairs, dim3((n + 63) / 64), dim3(64), 0, s, cfg, pts, cams, n, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// decode: poses[f][p][k] = y[row][k] * 10 (metrics_from_model.py:281-294), scattered from the
// compacted MLP rows back to fixed (frame, person) slots
// ---------------------------------------------------------------------------------------
__global__ void k_decode(int n_frames, int pcap, int n_out, float scale, const int32_t *__restrict__ n_persons,
                         const int32_t *__restrict__ person_off, const float *__restrict__ y, int ld_y,
                         float *__restrict__ poses) {
    const size_t slot = blockIdx.x;
    const int f = (int)(slot / pcap), p = (int)(slot - (size_t)f * pcap);
    float *o = poses + slot * n_out;
    if (p >= n_persons[f]) {
        for (int k = threadIdx.x; k < n_out; k += blockDim.x) o[k] = 0.f;
        return;
    }
    const float *src = y + ((size_t)person_off[f] + p) * ld_y;
    for (int k = threadIdx.x; k < n_out; k += blockDim.x) o[k] = src[k] * scale;
}

hipError_t launch_decode(hipStream_t s, int n_frames, int pcap, int n_out, float scale, const int32_t *n_persons,
                         const int32_t *person_off, const float *y, int ld_y, float *poses) {
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode, dim3(n_frames * pcap), dim3(64), 0, s, n_frames, pcap, n_out, scale, n_persons,
                       person_off, y, ld_y, poses);
    return hipGetLastError();
}

}  // namespace mpe

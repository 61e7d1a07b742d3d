// Small-batch ("latency") forms of the hot path: the reference's own call pattern is one frame per call
// (test/metrics_from_model.py:120-300), and a 5 x 4 frame is a chain of small dependent launches whose cost is each launch's
// own serial depth, not its arithmetic.  The kernels here compute EXACTLY what the batch kernels compute (same products, same
// summation orders, same bits: tests/test_gpu_latency.py) with the shortest serial depth the arithmetic allows:
//
//   k_lat_gemm         fc1 / fc2 of the graph-attention layers >= 1 (gat2.py:53-55) fed with bf16 PLANES written by the producer's epilogue
//                      (no wave splits anything), all weight fragments of a wave requested at once, coefficient epilogue
//   (gat.hip)          k_lat_l0a: topology tables + layer-0 fc1 with in-place featurisation; k_lat_attention: both halves of the
//                      attention stage in one launch;  (cluster.hip) k_lat_tail: last layer's scores + clustering + every pair solve
//
// Two experiments of round 6 lived here and are gone again (git show 102284c:3d_multi_pose_estimator_amd/csrc/lat.hip): the MLP launches from
// the fp32 weights split in registers (k_linear_lat_f64: slower than the plane kernels, profiles/r06_mlp_fp32_weights_experiment.txt) and
// the whole MLP as one launch with hand-offs between the XCDs (k_mlp_chain: 2.3x slower, profiles/r06_mlp_chain_experiment.txt).
//
// What bounds them: one memory round trip per launch plus the launch's own floor.  DESIGN.md 7.4 has the measurements and why these
// are launches and not one persistent kernel.
#include <cstdlib>

#include "mpe_internal.h"
#include "sb_common.h"

namespace mpe {
namespace lat {

using namespace sb;

// ---------------------------------------------------------------------------------------------------------------------------
// k_lat_gemm: fc1 / fc2 of a graph-attention layer >= 1 (gat2.py:53-55) for a few frames: the split-bf16 arithmetic WITHOUT f64
// sums (even / odd stages in two fp32 chains, added at the end: k_linear_sb<.., F64 = false>), one workgroup per (16-row tile,
// NT x 16 columns), one wave per 16 x 16 tile over the whole K.  No wave splits anything: the weights come as the context's planes,
// the activations as planes too (written by the producer's epilogue: one value per lane there, against 44 vector instructions per
// stage and wave here -- the wave-per-tile kernel k_linear_sb_skinny spends 2/3 of its time on that split).  Every weight fragment
// of the wave (NK stages x 3 planes) is requested before anything else, the activation tile goes through LDS once per workgroup.
// Epilogues: OPL -> LeakyReLU + the three planes of the result (fc1); COEF -> fp32 rows + the attention coefficients a1 | a2 of the
// workgroup's heads (fc2; canonical orders of gat.hip: coef40 for 40-wide heads, one chain otherwise).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void coef40_lat(const float *fv, const float *__restrict__ al, const float *__restrict__ ar, int parity,
                                           float &o1, float &o2) {
    // gat.hip: coef40 (the order of the tile kernels' epilogue): four chains over the features lane group q owns in an 80-wide tile
    float s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x1 = 0.f, x2 = 0.f;
        auto step = [&](int d) {
            const float v = fv[d];
            x1 = __builtin_fmaf(v, al[d], x1);
            x2 = __builtin_fmaf(v, ar[d], x2);
        };
        if (parity == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) step(16 * t + 4 * q + i);
            if (q < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) step(32 + 4 * q + i);
            }
        } else {
            if (q >= 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) step(4 * q - 8 + i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) step(8 + 4 * q + i);
#pragma unroll
            for (int i = 0; i < 4; ++i) step(24 + 4 * q + i);
        }
        s1[q] = x1;
        s2[q] = x2;
    }
    o1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    o2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
}

// FUSE2 (the last layer: fc1 150 -> 150 in ONE workgroup per row tile, then fc2 150 -> 1): the workgroup keeps its fc1 rows as planes
// in LDS and its first wave multiplies them with fc2's one weight row right away -- fp32 column 0 and a1 | a2 of the single
// one-wide head out, the arithmetic of this kernel's own <NK, 1, false, false, true> form; one launch and one round trip fewer.
struct LatFc2 {
    const unsigned short *W3;      // fc2's planes [3][weight_rows(1)][ldw]
    size_t w_plane;
    int ldw;
    const float *bias;
    float *C;                      // [rows][ldc] fp32: column 0
    int ldc;
    const float *attn_l, *attn_r;  // [1]
    float *a12;
};

template <int NK, int NT, bool LEAKY, bool OPL, bool COEF, bool FUSE2 = false, bool LOOP = false>
__global__ __launch_bounds__(64 * NT) void k_lat_gemm(const unsigned short *__restrict__ Apl, int lda, size_t a_plane,
                                                       const unsigned short *__restrict__ W3, size_t w_plane, int ldw,
                                                       const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                       unsigned short *__restrict__ Cpl, int ldcp, size_t c_plane, int M, int n,
                                                       float slope, const float *__restrict__ attn_l, const float *__restrict__ attn_r,
                                                       float *__restrict__ a12, int out_dim, LatFc2 f2 = LatFc2()) {
    static_assert(!FUSE2 || (OPL && !COEF && NT * 16 == NK * 32), "fused fc2: one workgroup holds whole fc1 rows, as wide as its own K");
    constexpr int KS = NK * 32 + 8;                          // LDS row stride of an activation plane (bf16 elements): rows 16 B apart in the banks
    constexpr int NTHR = 64 * NT;
    constexpr int A_CHUNKS = 3 * 16 * NK * 4;                // 16-byte chunks of the activation tile (3 planes x 16 rows x NK * 64 B)
    constexpr int A_PER = (A_CHUNKS + NTHR - 1) / NTHR;
    constexpr int TS = NT * 16 + 4;                          // row stride of the result tile (floats) for the coefficient epilogue
    __shared__ __attribute__((aligned(16))) unsigned short s_a[3 * 16 * KS > 2 * 16 * TS ? 3 * 16 * KS : 2 * 16 * TS];
    const int ngrp = (n + NT * 16 - 1) / (NT * 16);
    const int tm0 = blockIdx.x / ngrp, tg = blockIdx.x - tm0 * ngrp;
    const int tm_step = gridDim.x / ngrp, ntm = (M + 15) / 16;      // LOOP: a workgroup walks the row tiles tm0, tm0 + tm_step, ... with its weight fragments in registers
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fq = lane >> 4, fr = lane & 15;
    const int ct = tg * NT + wave;                           // this wave's 16-column tile (padded weight rows cover a tile that starts below n)
    // every weight fragment of the wave: one burst
    bf16x8 wv[NK][3];
    {
        const unsigned short *pw = W3 + (size_t)(ct * 16 + fr) * ldw + 8 * fq;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
#pragma unroll
            for (int p = 0; p < 3; ++p) wv[kt][p] = *reinterpret_cast<const bf16x8 *>(pw + p * w_plane + kt * GEMM_BK);
    }
    bf16x8 wv2[FUSE2 ? NK : 1][3];                         // fc2's fragments (its sixteen output "columns" are one real row + zero padding)
    if (FUSE2 && wave == 0) {
        const unsigned short *pw2 = f2.W3 + (size_t)fr * f2.ldw + 8 * fq;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
#pragma unroll
            for (int p = 0; p < 3; ++p) wv2[kt][p] = *reinterpret_cast<const bf16x8 *>(pw2 + p * f2.w_plane + kt * GEMM_BK);
    }
    int tm = tm0 - tm_step;
    do {                                                     // (one pass without LOOP: the code of a single tile, no loop-carried registers)
    tm += tm_step;
    if (LOOP && tm != tm0) __syncthreads();                  // everybody is done with the previous tile's LDS image
    // the activation tile (shared by the NT waves): global -> registers -> LDS
    {
        u32x4 stage[A_PER];
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            int c = t + j * NTHR;
            c = c < A_CHUNKS ? c : A_CHUNKS - 1;
            const int p = c / (16 * NK * 4), rem = c - p * (16 * NK * 4);
            const int r = rem / (NK * 4), ch = rem - r * (NK * 4);
            int grow = tm * 16 + r;
            grow = grow < M ? grow : M - 1;
            stage[j] = *reinterpret_cast<const u32x4 *>(Apl + p * a_plane + (size_t)grow * lda + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int c = t + j * NTHR;
            if (c < A_CHUNKS) {
                const int p = c / (16 * NK * 4), rem = c - p * (16 * NK * 4);
                const int r = rem / (NK * 4), ch = rem - r * (NK * 4);
                *reinterpret_cast<u32x4 *>(&s_a[(p * 16 + r) * KS + ch * 8]) = stage[j];
            }
        }
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_odd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        bf16x8 ap[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ap[p] = *reinterpret_cast<const bf16x8 *>(&s_a[(p * 16 + fr) * KS + kt * GEMM_BK + 8 * fq]);
        if (kt & 1) SB_STAGE(acc_odd, ap, wv[kt]);
        else SB_STAGE(acc, ap, wv[kt]);
    }
    const int m = tm * 16 + fr, nb = ct * 16 + fq * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nb);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = (acc[i] + acc_odd[i]) + bv[i];
        if (LEAKY) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
    }
    if (FUSE2) {
        // this workgroup's fc1 rows (all NT x 16 columns; the ones behind n are exact zeros: zero weight rows, zero bias) as planes into
        // the LDS tile the activation planes came through, then fc2 on them by the first wave
        __syncthreads();
        {
            const unsigned a0 = pack2_lat(v[0], v[1]), a1 = pack2_lat(v[2], v[3]);
            const float r0 = sub1(v[0], __uint_as_float(a0 << 16)), r1 = sub1(v[1], __uint_as_float(a0 & 0xFFFF0000u));
            const float r2 = sub1(v[2], __uint_as_float(a1 << 16)), r3 = sub1(v[3], __uint_as_float(a1 & 0xFFFF0000u));
            const unsigned b0 = pack2_lat(r0, r1), b1 = pack2_lat(r2, r3);
            const float s0 = sub1(r0, __uint_as_float(b0 << 16)), s1 = sub1(r1, __uint_as_float(b0 & 0xFFFF0000u));
            const float s2 = sub1(r2, __uint_as_float(b1 << 16)), s3 = sub1(r3, __uint_as_float(b1 & 0xFFFF0000u));
            unsigned short *d = &s_a[fr * KS + wave * 16 + fq * 4];
            *reinterpret_cast<uint2 *>(d) = make_uint2(a0, a1);
            *reinterpret_cast<uint2 *>(d + 16 * KS) = make_uint2(b0, b1);
            *reinterpret_cast<uint2 *>(d + 32 * KS) = make_uint2(pack2_lat(s0, s1), pack2_lat(s2, s3));
        }
        __syncthreads();
        if (wave != 0) continue;
        f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            bf16x8 ap[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) ap[p] = *reinterpret_cast<const bf16x8 *>(&s_a[(p * 16 + fr) * KS + kt * GEMM_BK + 8 * fq]);
            if (kt & 1) SB_STAGE(o, ap, wv2[kt]);
            else SB_STAGE(e, ap, wv2[kt]);
        }
        if (fq == 0 && m < M) {                              // output column 0 of the row
            const float y = (e[0] + o[0]) + f2.bias[0];
            f2.C[(size_t)m * f2.ldc] = y;
            f2.a12[(size_t)m * 32] = __builtin_fmaf(y, f2.attn_l[0], 0.f);
            f2.a12[(size_t)m * 32 + 16] = __builtin_fmaf(y, f2.attn_r[0], 0.f);
        }
        continue;
    }
    if (OPL) {
        if (m < M && nb + 3 < n) {
            store_planes4(Cpl + (size_t)m * ldcp + nb, c_plane, v);
        } else if (m < M) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (nb + i < n) store_planes1(Cpl + (size_t)m * ldcp + nb + i, c_plane, v[i]);
        }
    } else {
        if (m < M) {
            float *dst = C + (size_t)m * ldc + nb;
            if (nb + 3 < n) {
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < n) dst[i] = v[i];
            }
        }
    }
    if (COEF) {
        // the workgroup's result tile through LDS (the activation tile is not needed any more), then one thread per (row, head)
        float *s_t = reinterpret_cast<float *>(s_a);
        __syncthreads();
        *reinterpret_cast<f32x4 *>(&s_t[fr * TS + wave * 16 + fq * 4]) = v;
        __syncthreads();
        const int c0 = tg * NT * 16;                         // first column of the workgroup
        const int h0 = c0 / out_dim;                         // (the host launches this epilogue only where heads do not straddle workgroups)
        int nh = (NT * 16) / out_dim;
        if ((h0 + nh) * out_dim > n) nh = (n - c0) / out_dim;
        if (nh < 1 && c0 < n) nh = 1;
        for (int i = t; i < 16 * nh; i += NTHR) {
            const int r = i / nh, hl = i - r * nh, h = h0 + hl;
            const int mm = tm * 16 + r;
            if (mm >= M) continue;
            const float *fv = s_t + r * TS + (h * out_dim - c0);
            const float *al = attn_l + h * out_dim, *ar = attn_r + h * out_dim;
            float x1 = 0.f, x2 = 0.f;
            if (out_dim == 40) {
                coef40_lat(fv, al, ar, h & 1, x1, x2);
            } else {
                for (int d = 0; d < out_dim; ++d) {
                    x1 = __builtin_fmaf(fv[d], al[d], x1);
                    x2 = __builtin_fmaf(fv[d], ar[d], x2);
                }
            }
            a12[(size_t)mm * 32 + h] = x1;
            a12[(size_t)mm * 32 + 16 + h] = x2;
        }
    }
    } while (LOOP && tm + tm_step < ntm);                    // row tiles of the workgroup
}

}  // namespace lat

// Row-tile groups of an fc1 launch: one workgroup per (row tile, column group) while that fits the chip in one round (the kernels hold
// 150-200 registers: one workgroup per CU); beyond that (eight frames: 90 row tiles x 5 column groups) as many groups as give one
// workgroup per CU, each walking several row tiles with its weight fragments in registers -- a second round would fetch them again
// and pay the kernel's whole serial depth twice (fc1 of eight frames: 14.6 -> 11.8 us, 16.0 -> 11.4 on the 320-wide layer).  Only
// the fc1 forms: with the coefficient epilogue the loop form needs more registers than the wave has (29 spilled: 18.2 -> 20.5 us).
static int lat_row_groups(int ntm, int ngrp) {
    const int cu = device_cu_count();
    if (ntm * ngrp <= cu) return ntm;
    const int g = cu / ngrp;
    return g < 1 ? 1 : g < ntm ? g : ntm;
}

// Does a layer of this shape have an instantiation of k_lat_gemm?  fc1 (leaky, planes out) / fc2 (fp32 out + coefficients):
// K stages and columns per workgroup as the deployed network has them (train_skeleton_matching.py:40-57: 400 / 320 / 150 wide).
static int lat_gemm_form(int k_pad, int n, bool fc2, int out_dim) {
    const int nk = k_pad / GEMM_BK;
    if (!fc2) {
        if (nk == 13 && n % 16 == 0) return 1;               // 400 -> 400
        if (nk == 10 && n % 16 == 0) return 2;               // 320 -> 320
        if (nk == 5 && n <= 160) return 3;                   // 150 -> 150: one workgroup of ten waves per row tile
        return 0;
    }
    if (nk == 13 && out_dim == 40 && n % 80 == 0) return 4;  // 400 -> H x 40, coefficients from the epilogue
    if (nk == 10) return 5;                                  // 320 -> 5 x 30: fp32 rows only (80-column workgroups cut the 30-wide heads; the attention kernel computes a1 | a2)
    if (nk == 5 && n <= 16) return 6;                        // 150 -> 1, coefficients from the epilogue
    return 0;
}

bool lat_gemm_available(int k_pad, int n, bool fc2, int out_dim) { return lat_gemm_form(k_pad, n, fc2, out_dim) != 0; }

// fc1 (150 -> 150) and fc2 (150 -> 1) of the last layer as one launch?
bool lat_gemm_fusable(int k1_pad, int n1, int k2_pad, int n2, int out_dim2) {
    return lat_gemm_form(k1_pad, n1, false, 0) == 3 && lat_gemm_form(k2_pad, n2, true, out_dim2) == 6 && n2 == 1 && out_dim2 == 1 && k2_pad == 160 && n1 <= 160;
}

hipError_t launch_lat_gemm_fused(hipStream_t s, const unsigned short *Apl, int lda, size_t a_plane, const unsigned short *W3, size_t w_plane, int ldw,
                                 const float *bias, int m, int n, float slope, const unsigned short *W3b, size_t w_plane_b, int ldw_b, const float *bias_b,
                                 float *C2, int ldc2, const float *attn_l, const float *attn_r, float *a12) {
    if (m <= 0) return hipSuccess;
    lat::LatFc2 f2{W3b, w_plane_b, ldw_b, bias_b, C2, ldc2, attn_l, attn_r, a12};
    const int ntm = (m + 15) / 16;
    hipLaunchKernelGGL((lat::k_lat_gemm<5, 10, true, true, false, true>), dim3((unsigned)ntm), dim3(640), 0, s, Apl, lda, a_plane, W3, w_plane, ldw, bias,
                       nullptr, 0, nullptr, 0, 0, m, n, slope, nullptr, nullptr, nullptr, 0, f2);
    return hipGetLastError();
}

hipError_t launch_lat_gemm(hipStream_t s, const unsigned short *Apl, int lda, size_t a_plane, const unsigned short *W3, size_t w_plane, int ldw,
                           const float *bias, float *C, int ldc, unsigned short *Cpl, int ldcp, size_t c_plane, int m, int n, int k_pad, bool fc2,
                           float slope, const float *attn_l, const float *attn_r, float *a12, int out_dim, bool *coef_done) {
    if (coef_done) *coef_done = false;
    if (m <= 0 || n <= 0) return hipSuccess;
    const int form = lat_gemm_form(k_pad, n, fc2, out_dim);
    if (coef_done) *coef_done = form == 4 || form == 6;
    const int ntm = (m + 15) / 16;
#define MPE_LG(NK_, NT_, L_, O_, C_)                                                                                                         \
    hipLaunchKernelGGL((lat::k_lat_gemm<NK_, NT_, L_, O_, C_>), dim3((unsigned)(ntm * ((n + NT_ * 16 - 1) / (NT_ * 16)))), dim3(64 * NT_), 0, s, Apl, lda, \
                       a_plane, W3, w_plane, ldw, bias, C, ldc, Cpl, ldcp, c_plane, m, n, slope, attn_l, attn_r, a12, out_dim)
    // fc1 of the wide layers at several frames: more (row tile, column group) pairs than CUs -> as many row-tile groups as give one
    // workgroup per CU, each walking its row tiles with the weight fragments in registers (lat_row_groups)
#define MPE_LG_LOOP(NK_, NT_)                                                                                                                \
    do {                                                                                                                                     \
        const int ngrp_ = (n + NT_ * 16 - 1) / (NT_ * 16), g_ = lat_row_groups(ntm, ngrp_);                                                  \
        if (g_ < ntm)                                                                                                                        \
            hipLaunchKernelGGL((lat::k_lat_gemm<NK_, NT_, true, true, false, false, true>), dim3((unsigned)(g_ * ngrp_)), dim3(64 * NT_), 0, s, Apl, lda,     \
                               a_plane, W3, w_plane, ldw, bias, C, ldc, Cpl, ldcp, c_plane, m, n, slope, attn_l, attn_r, a12, out_dim);      \
        else MPE_LG(NK_, NT_, true, true, false);                                                                                            \
    } while (0)
    switch (form) {
    case 1: MPE_LG_LOOP(13, 5); break;
    case 2: MPE_LG_LOOP(10, 5); break;
    case 3: MPE_LG(5, 10, true, true, false); break;
    case 4: MPE_LG(13, 5, false, false, true); break;
    case 5: MPE_LG(10, 5, false, false, false); break;
    case 6: MPE_LG(5, 1, false, false, true); break;
    default: return hipErrorInvalidValue;
    }
#undef MPE_LG_LOOP
#undef MPE_LG
    return hipGetLastError();
}

}  // namespace mpe

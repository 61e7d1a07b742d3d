// Small-batch ("latency") forms of the hot path: the reference's own call pattern is one frame per call
// (test/metrics_from_model.py:120-300), and a 5 x 4 frame is a chain of small dependent launches whose cost is each launch's
// own serial depth, not its arithmetic.  The kernels here compute EXACTLY what the batch kernels compute (same products, same
// summation orders, same bits: tests/test_gpu_latency.py) with the shortest serial depth the arithmetic allows:
//
//   k_linear_lat_f64   nn.Linear (+LeakyReLU) with f64 running sums (utils/mlp.py:8-28; fc2 of GAT layer 0, gat2.py:55) for a
//                      handful of rows: one workgroup per 16-feature tile, the flush units (stage pairs; single stages in the
//                      maximum-accuracy mode) of the whole K dealt to eight waves, EVERY weight fragment of a wave requested
//                      before the first product (the batch form's small-batch kernel k_linear_sb_ks asks for one pair at a
//                      time: K / 64 / 8 dependent round trips to HBM per launch), weights streamed ONCE from the fp32 copy
//                      (4 B per weight; the batch kernels stream the three bf16 planes, 6 B) and split in registers into the
//                      same three planes, ordered f64 reduction through LDS
//
// What bounds them: the weight stream (116 MB per MLP pass at small batch: HBM / Infinity Cache bandwidth) plus one memory
// round trip per launch.  DESIGN.md 7.4 has the measurements and why these are launches and not one persistent kernel.
#include <cstdlib>

#include "mpe_internal.h"
#include "sb_common.h"

namespace mpe {
namespace lat {

using namespace sb;

constexpr int LF_WAVES = 16;                   // waves per workgroup
constexpr int LF_UPW = 3;                      // flush units per wave and round
constexpr int LF_ROUND = LF_WAVES * LF_UPW;    // units per round: K = 3072 at two stages per unit is one round
constexpr int LF_MAXPASS = 8;                  // row tiles of 16 (the weight fragments stay in registers across them)

// One workgroup = one 16-feature tile of the output, all rows (M <= 16 LF_MAXPASS).  Unit u = K stages [FL u, FL u + FL): one fp32
// chain started from zero, the six products per stage in the canonical order (sb_common.h) -- k_linear_sb's flush interval.  Wave w
// takes the units w, w + 8, ... of a round, parks each chain's result in LDS; after the round 256 threads add "their" element over
// the units IN UNIT ORDER into the f64 running sum (kept in LDS between rounds): the additions of k_linear_sb in the same order.
// What the kernel costs besides its weight stream is the split of the fragments (vector instructions, 4 cycles each): the weights'
// is unavoidable at 4 B per weight (44 instructions per 16 x 32 fragment = 2.7 us of vector issue for a 3072 x 3072 layer, under
// the stream); the activations' would be the same again in EVERY workgroup for the same few rows, so between the layers of a
// chain the activations travel as planes: APL = A is [3][rows][lda] bf16 (split by the producer's epilogue, one value per thread),
// OPL = the epilogue stores such planes at Cp (ldc elements per row, plane stride c_plane) instead of fp32 rows.
template <bool LEAKY, int FL, bool APL, bool OPL>
__global__ __launch_bounds__(64 * LF_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_linear_lat_f64(
    const void *__restrict__ Av, int lda, size_t a_plane, const float *__restrict__ W, int ldw, const float *__restrict__ bias,
    void *__restrict__ Cv, int ldc, size_t c_plane, int m_cap, const int32_t *__restrict__ d_m, int n, int k_pad, float slope) {
    __shared__ __attribute__((aligned(16))) float s_part[LF_ROUND * 256];          // [unit of the round][lane][4]
    __shared__ double s_run[LF_MAXPASS * 256];
    int M = m_cap;
    if (d_m) {
        const int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    if (M <= 0) return;
    const int tn = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fq = lane >> 4, fr = lane & 15;
    const int nk = k_pad / GEMM_BK, nu = (nk + FL - 1) / FL;
    const int npass = (M + 15) / 16 < LF_MAXPASS ? (M + 15) / 16 : LF_MAXPASS;
    const float *pw = W + (size_t)(tn * 16 + fr) * ldw + 8 * fq;           // stage kt: k = 32 kt + 8 fq + {0..7}
    constexpr int AV = APL ? 3 : 2;                                        // 16-byte pieces of an activation fragment
    // (No run-time branch between the first request and the last product of a round: behind a branch the compiler's wait-count
    // bookkeeping falls back to vmcnt(0) and the wave's requests serialise.)
    for (int u0 = 0; u0 < nu; u0 += LF_ROUND) {
        // stage kt of unit i of this wave (clamped: a stage behind the end repeats the last one and its products are dropped)
        auto stage_of = [&](int i, int s) {
            const int kt = FL * (u0 + wave + LF_WAVES * i) + s;
            return kt < nk ? kt : nk - 1;
        };
        // own units of the round: i = 0 .. cnt - 1
        const int left = nu - u0 - wave;
        const int cnt = left <= 0 ? 0 : (left + LF_WAVES - 1) / LF_WAVES < LF_UPW ? (left + LF_WAVES - 1) / LF_WAVES : LF_UPW;
        for (int pass = 0; pass < npass; ++pass) {
            int grow = pass * 16 + fr;
            grow = grow < M ? grow : M - 1;
            const size_t pa = (size_t)grow * lda + 8 * fq;
            // A SMALL loop, double-buffered through two named register sets (the unit being multiplied and the next one: with
            // sixteen waves 64 KB of fragments in flight per CU), not one straight line over all the wave's units: a kernel that runs
            // ONCE pays for every instruction line it fetches and for every product on a unit behind the end (measured: the fully
            // unrolled eight-wave form, 15 KB of code, took 12.6 us per launch -- 11.5 us for the 54-wide last layer on four
            // workgroups; a third of that code 5.6 us)
            f32x4 wv[2][FL][2], av[2][FL][AV];
            auto load = [&](int buf, int i) {
#pragma unroll
                for (int s = 0; s < FL; ++s) {
                    const int kt = stage_of(i, s);
                    const size_t ko = pa + (size_t)kt * GEMM_BK;
                    if (APL) {
                        const unsigned short *q = static_cast<const unsigned short *>(Av) + ko;
#pragma unroll
                        for (int p = 0; p < 3; ++p) av[buf][s][p] = *reinterpret_cast<const f32x4 *>(q + p * a_plane);
                    } else {
                        const float *q = static_cast<const float *>(Av) + ko;
                        av[buf][s][0] = *reinterpret_cast<const f32x4 *>(q);
                        av[buf][s][1] = *reinterpret_cast<const f32x4 *>(q + 4);
                    }
                    wv[buf][s][0] = *reinterpret_cast<const f32x4 *>(pw + kt * GEMM_BK);
                    wv[buf][s][1] = *reinterpret_cast<const f32x4 *>(pw + kt * GEMM_BK + 4);
                }
            };
            auto compute = [&](int buf, int i) {
                const int u = u0 + wave + LF_WAVES * i;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < FL; ++s) {
                    bf16x8 wp[3], ap[3];
                    split8_lat(wv[buf][s][0], wv[buf][s][1], wp[0], wp[1], wp[2]);
                    if (APL) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) ap[p] = __builtin_bit_cast(bf16x8, av[buf][s][p]);
                    } else {
                        split8_lat(av[buf][s][0], av[buf][s][1], ap[0], ap[1], ap[2]);
                    }
                    f32x4 nx = acc;
                    SB_STAGE(nx, ap, wp);
                    if (s == 0) {
                        acc = nx;
                    } else {                              // the second stage of the last unit of an odd stage count does not exist
                        const bool live = FL * u + s < nk;
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = live ? nx[c] : acc[c];
                    }
                }
                if (i < cnt) *reinterpret_cast<f32x4 *>(&s_part[((wave + LF_WAVES * i) * 64 + lane) * 4]) = acc;
            };
            load(0, 0);
#pragma unroll 1
            for (int i = 0; i < cnt; i += 2) {
                load(1, i + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(0, i);
                __builtin_amdgcn_sched_barrier(0);
                load(0, i + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(1, i + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (t < 256) {
                const int cnt_r = nu - u0 < LF_ROUND ? nu - u0 : LF_ROUND;
                double run = u0 ? s_run[pass * 256 + t] : 0.0;
                for (int j = 0; j < cnt_r; ++j) run += (double)s_part[j * 256 + t];      // unit order, as the tile kernel
                if (u0 + LF_ROUND < nu) {
                    s_run[pass * 256 + t] = run;
                } else {
                    const int el = t >> 2, i = t & 3;                    // element (lane el, component i) of the MFMA tile
                    const int m = pass * 16 + (el & 15), nb = tn * 16 + (el >> 4) * 4 + i;
                    if (m < M && nb < n) {
                        float v = (float)(run + (double)bias[nb]);
                        if (LEAKY) v = v > 0.f ? v : v * slope;
                        if (OPL) {
                            store_planes1(static_cast<unsigned short *>(Cv) + (size_t)m * ldc + nb, c_plane, v);   // as split8 makes them
                        } else {
                            static_cast<float *>(Cv)[(size_t)m * ldc + nb] = v;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

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

template <int NK, int NT, bool LEAKY, bool OPL, bool COEF, bool FUSE2 = false>
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
    const int tm = blockIdx.x / ngrp, tg = blockIdx.x - tm * ngrp;
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
        if (wave != 0) return;
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
        return;
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
}

}  // namespace lat

// rows a launch of the latency kernels may hold (beyond it the batch kernels are the better form anyway)
int linear_lat_max_rows() { return 16 * lat::LF_MAXPASS; }

// a_planes / c_planes: the activations come in / go out as three bf16 planes ([3][rows][ld], plane stride in elements) instead of
// fp32 rows (the chain of an MLP pass: fp32 rows in, planes between the layers, fp32 rows out)
hipError_t launch_linear_lat_f64(hipStream_t s, const void *A, int lda, size_t a_plane, const float *W, int ldw, const float *bias, void *C,
                                 int ldc, size_t c_plane, int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky, float slope,
                                 int flush_stages, bool a_planes, bool c_planes) {
    if (m_cap <= 0 || n <= 0) return hipSuccess;
    if (m_cap > linear_lat_max_rows() || (flush_stages != 1 && flush_stages != 2) || k_pad % GEMM_BK) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 15) / 16)), block(64 * lat::LF_WAVES);
#define MPE_LAT5(L_, F_, AP_, CP_) \
    hipLaunchKernelGGL((lat::k_linear_lat_f64<L_, F_, AP_, CP_>), grid, block, 0, s, A, lda, a_plane, W, ldw, bias, C, ldc, c_plane, m_cap, d_m, n, k_pad, slope)
#define MPE_LAT(L_, F_)                               \
    do {                                              \
        if (a_planes && c_planes) MPE_LAT5(L_, F_, true, true);    \
        else if (a_planes) MPE_LAT5(L_, F_, true, false);          \
        else if (c_planes) MPE_LAT5(L_, F_, false, true);          \
        else MPE_LAT5(L_, F_, false, false);                       \
    } while (0)
    if (leaky && flush_stages == 2) MPE_LAT(true, 2);
    else if (leaky) MPE_LAT(true, 1);
    else if (flush_stages == 2) MPE_LAT(false, 2);
    else MPE_LAT(false, 1);
#undef MPE_LAT
#undef MPE_LAT5
    return hipGetLastError();
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
    switch (form) {
    case 1: MPE_LG(13, 5, true, true, false); break;
    case 2: MPE_LG(10, 5, true, true, false); break;
    case 3: MPE_LG(5, 10, true, true, false); break;
    case 4: MPE_LG(13, 5, false, false, true); break;
    case 5: MPE_LG(10, 5, false, false, false); break;
    case 6: MPE_LG(5, 1, false, false, true); break;
    default: return hipErrorInvalidValue;
    }
#undef MPE_LG
    return hipGetLastError();
}

}  // namespace mpe

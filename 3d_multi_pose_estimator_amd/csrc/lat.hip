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

// ---------------------------------------------------------------------------------------------------------------------------
// k_mlp_chain: the WHOLE MLP (utils/mlp.py:8-28) for a handful of rows in ONE launch.  Nine dependent launches are nine floors of ~4.2 us
// and nine cold starts of the weight stream; here a workgroup takes 16-feature tiles of layer after layer from a shared ticket counter
// and hands its results to the next layer's workgroups through memory:
//   * NO workgroup ever waits for a workgroup that has not started: a tile is computed by whoever drew its ticket, and the only wait is
//     for "all tiles of the previous layer done" -- tiles that are in the hands of running workgroups.  The launch makes progress with
//     any number of resident workgroups (several contexts in flight share the chip; no co-residency argument, no cooperative launch).
//   * a tile's weight fragments do not depend on the previous layer: they are requested BEFORE the wait, so the stream of layer l + 1
//     starts while layer l is still being finished elsewhere.
//   * hand-off across the eight XCDs (their L2s are not coherent): the producer stores its 16 x 16 result tile as three bf16 planes with
//     write-through (sc1) stores, whole 128-byte lines per instruction, waits for them (vmcnt(0)), then adds 1 to the layer's counter
//     (agent scope); the consumer polls the counter with sc1 loads, passes a workgroup barrier and reads the planes with sc1 loads
//     (MI355X_MICROARCH.md, "hand-offs measured with sc1 loads in place of the acquire", third row).  No fence anywhere.
//   * every spin is bounded (a stuck launch raises a status bit and drains instead of hanging the queue).
// Arithmetic: k_linear_sb's for f64-sum launches -- flush units of FL stages, six products per stage in the canonical order, the units
// of a tile dealt to the eight waves, ordered f64 sum through LDS -- with the weights from the context's planes and the activations
// of layers >= 1 as the planes the producer made of its fp32 results (split8's planes of the same numbers): the same bits.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int MC_WAVES = 8;
constexpr int MC_SPW = 12;                     // K stages per wave (K <= 8 x 12 x 32 = 3072)
constexpr int MC_WSL = 10;                     // of which this many have their weight fragments requested before the wait (120 registers); the rest refill used slots
constexpr int MC_CTL_DONE = MPE_MAX_MLP_LAYERS;            // ctl: [0, 16) tickets, [16, 32) tiles done, [32] workgroups that left, [33] abort
constexpr int MC_CTL_EXIT = 2 * MPE_MAX_MLP_LAYERS, MC_CTL_ABORT = 2 * MPE_MAX_MLP_LAYERS + 1;

struct ChainLayer {
    const unsigned short *W3;      // [3][weight_rows(n)][ldw]
    size_t w_plane;
    int ldw;
    const float *bias;
    int n, leaky;
};

struct ChainArgs {
    ChainLayer L[MPE_MAX_MLP_LAYERS];
    int n_layers;
    const float *A0;               // fp32 input rows [rows][lda0]
    int lda0;
    unsigned short *act[2];        // activations between the layers: [3 planes][n tiles][16 rows][16 columns] bf16, plane stride act_plane
    size_t act_plane;
    int32_t *ctl;
    int32_t *status;               // the context's device status word (bit 2: a launch gave up waiting)
    int m_cap;
    const int32_t *d_m;
    float slope;
    float *y;                      // last layer, fp32 rows
    int ldy;
    DecodeEpi dec;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mc_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, 0x7FFFFFFF, 0x00020000);
}

// The twelve stages of a wave: activation fragments in four register slots, refilled as they are used (a slot.s next fragment has three
// stages of products to arrive in), straight-line, so that every product waits for its own fragments only.  L0: fp32 rows, split here.
template <int FL, bool L0>
__device__ __forceinline__ void mc_products(const float *__restrict__ pa0, const unsigned short *__restrict__ ain, size_t act_plane,
                                            bf16x8 (&wv)[MC_WSL][3], const unsigned short *__restrict__ pw, size_t w_plane, int nk, int nu, int wave,
                                            int lane, float *s_part) {
    constexpr int SL = 4;                                    // slots (six spill: 144 registers of weight fragments are live beside them)
    static_assert(SL % FL == 0, "a unit's stages sit in one half");
    const int fq = lane >> 4, fr = lane & 15;
    const __amdgpu_buffer_rsrc_t ra = mc_rsrc(ain);
    f32x4 slot[SL][L0 ? 2 : 3];
    auto stage_kc = [&](int q) {
        const int kt = FL * (wave + MC_WAVES * (q / FL)) + (q % FL);
        return kt < nk ? kt : nk - 1;
    };
    auto fetch = [&](int sl, int q) {
        const int kc = stage_kc(q);
        if (L0) {
            slot[sl][0] = *reinterpret_cast<const f32x4 *>(pa0 + kc * GEMM_BK);
            slot[sl][1] = *reinterpret_cast<const f32x4 *>(pa0 + kc * GEMM_BK + 4);
        } else {
            const int k0 = kc * GEMM_BK + 8 * fq;
            const unsigned off = (unsigned)((((k0 >> 4) * 16 + fr) * 16 + (k0 & 15)) * 2);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                slot[sl][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off + (unsigned)(p * act_plane * 2), 0, 16));   // sc1
        }
    };
#pragma unroll
    for (int q = 0; q < SL; ++q) fetch(q, q);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < MC_SPW; ++q) {
        const int u = wave + MC_WAVES * (q / FL), kt = FL * u + (q % FL);
        if (q % FL == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 ap[3];
        if (L0) {
            split8_lat(slot[q % SL][0], slot[q % SL][1], ap[0], ap[1], ap[2]);
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p) ap[p] = __builtin_bit_cast(bf16x8, slot[q % SL][p]);
        }
        if (q + SL < MC_SPW) fetch(q % SL, q + SL);
        f32x4 nx = acc;
        SB_STAGE(nx, ap, wv[q % MC_WSL]);
        if (q + MC_WSL < MC_SPW) {                           // this slot's next weight fragment (stages 10, 11 of a 3072-deep layer)
            const int kc = stage_kc(q + MC_WSL);
#pragma unroll
            for (int p = 0; p < 3; ++p) wv[q % MC_WSL][p] = *reinterpret_cast<const bf16x8 *>(pw + p * w_plane + kc * GEMM_BK);
        }
        const bool live = kt < nk;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = live ? nx[c] : acc[c];
        if (q % FL == FL - 1 && u < nu) *reinterpret_cast<f32x4 *>(&s_part[(u * 64 + lane) * 4]) = acc;
        __builtin_amdgcn_sched_barrier(0);                   // (left alone the scheduler hoists the refills of later stages and spills)
    }
}

template <int FL>
__global__ __launch_bounds__(64 * MC_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mlp_chain(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_part[];        // [units of a tile][lane][4]
    __shared__ float s_tile[256];
    __shared__ int s_t;
    int M = a.m_cap;
    if (a.d_m) {
        const int dm = *a.d_m;
        M = dm < a.m_cap ? dm : a.m_cap;
    }
    if (M <= 0) return;                                       // (nothing was drawn, nothing to reset)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fq = lane >> 4, fr = lane & 15;
    int grow = fr < M ? fr : M - 1;                           // (at most 16 rows: the caller routes larger batches elsewhere)
    bool aborted = false;
    int k_pad = a.L[0].ldw;
    for (int l = 0; l < a.n_layers && !aborted; ++l) {
        const ChainLayer &L = a.L[l];
        k_pad = L.ldw;
        const int nk = k_pad / GEMM_BK, nu = (nk + FL - 1) / FL;
        const int ntile = (L.n + 15) / 16;
        const bool last = l == a.n_layers - 1;
        for (;;) {
            if (t == 0) s_t = __hip_atomic_fetch_add(a.ctl + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int tile = s_t;
            __syncthreads();
            if (tile >= ntile) break;
            // this wave's weight fragments: stages FL u + s of the units u = wave, wave + 8, ... (requested before the wait)
            bf16x8 wv[MC_WSL][3];
            const unsigned short *pw = L.W3 + (size_t)(tile * 16 + fr) * L.ldw + 8 * fq;
#pragma unroll
            for (int q = 0; q < MC_WSL; ++q) {
                const int u = wave + MC_WAVES * (q / FL), kt = FL * u + (q % FL);
                const int kc = kt < nk ? kt : nk - 1;
#pragma unroll
                for (int p = 0; p < 3; ++p) wv[q][p] = *reinterpret_cast<const bf16x8 *>(pw + p * L.w_plane + kc * GEMM_BK);
            }
            __builtin_amdgcn_sched_barrier(0);               // (the burst stays in front of the wait)
            if (l > 0) {
                // every tile of the previous layer stored?  (one lane polls; bounded)
                if (t == 0) {
                    const int want = (a.L[l - 1].n + 15) / 16;
                    int spins = 0, ok = 1;
                    while (__hip_atomic_load(a.ctl + MC_CTL_DONE + l - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 22) || __hip_atomic_load(a.ctl + MC_CTL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            ok = 0;
                            break;
                        }
                    }
                    if (!ok) {
                        __hip_atomic_store(a.ctl + MC_CTL_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.status) atomicOr(a.status, 4);
                    }
                    s_t = ok;
                }
                __syncthreads();
                const int ok = s_t;
                __syncthreads();
                if (!ok) {
                    aborted = true;
                    break;
                }
            }
            // the products: one fp32 chain per unit, parked for the ordered sum
            if (l == 0) mc_products<FL, true>(a.A0 + (size_t)grow * a.lda0 + 8 * fq, nullptr, 0, wv, pw, L.w_plane, nk, nu, wave, lane, s_part);
            else mc_products<FL, false>(nullptr, a.act[(l - 1) & 1], a.act_plane, wv, pw, L.w_plane, nk, nu, wave, lane, s_part);
            __syncthreads();
            if (t < 256) {
                double run = 0.0;
                for (int j = 0; j < nu; ++j) run += (double)s_part[j * 256 + t];       // unit order, as the tile kernel
                const int el = t >> 2, i = t & 3;
                const int m = el & 15, c = (el >> 4) * 4 + i, nb = tile * 16 + c;
                float v = (float)(run + (double)L.bias[nb]);
                if (L.leaky) v = v > 0.f ? v : v * a.slope;
                if (last) {
                    if (m < M && nb < L.n) {
                        a.y[(size_t)m * a.ldy + nb] = v;
                        if (a.dec.poses) {
                            int lo = 0, hi = a.dec.n_frames;
                            while (hi - lo > 1) {
                                const int mid = (lo + hi) >> 1;
                                if (a.dec.person_off[mid] <= m) lo = mid;
                                else hi = mid;
                            }
                            const int pp = m - a.dec.person_off[lo];
                            if (pp < a.dec.pcap && nb < a.dec.n_out) a.dec.poses[((size_t)lo * a.dec.pcap + pp) * a.dec.n_out + nb] = v * a.dec.scale;
                        }
                    }
                } else {
                    s_tile[m * 16 + c] = v;
                }
            }
            __syncthreads();
            if (!last && wave == 0) {
                // the tile as three planes, 512 bytes each: one write-through store instruction per plane (four whole lines), then the count
                const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_tile[lane * 4]);
                const unsigned a0 = pack2_lat(v[0], v[1]), a1 = pack2_lat(v[2], v[3]);
                const float r0 = sub1(v[0], __uint_as_float(a0 << 16)), r1 = sub1(v[1], __uint_as_float(a0 & 0xFFFF0000u));
                const float r2 = sub1(v[2], __uint_as_float(a1 << 16)), r3 = sub1(v[3], __uint_as_float(a1 & 0xFFFF0000u));
                const unsigned b0 = pack2_lat(r0, r1), b1 = pack2_lat(r2, r3);
                const float s0 = sub1(r0, __uint_as_float(b0 << 16)), s1 = sub1(r1, __uint_as_float(b0 & 0xFFFF0000u));
                const float s2 = sub1(r2, __uint_as_float(b1 << 16)), s3 = sub1(r3, __uint_as_float(b1 & 0xFFFF0000u));
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const __amdgpu_buffer_rsrc_t ro = mc_rsrc(a.act[l & 1]);
                const unsigned off = (unsigned)((tile * 256 + lane * 4) * 2);
                __builtin_amdgcn_raw_buffer_store_b64((v2u){a0, a1}, ro, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b64((v2u){b0, b1}, ro, off + (unsigned)(a.act_plane * 2), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b64((v2u){pack2_lat(s0, s1), pack2_lat(s2, s3)}, ro, off + (unsigned)(2 * a.act_plane * 2), 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(a.ctl + MC_CTL_DONE + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the last workgroup to leave puts the counters back to zero for the next launch on this context
    __syncthreads();
    if (t == 0) {
        const int left = __hip_atomic_fetch_add(a.ctl + MC_CTL_EXIT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (int)gridDim.x - 1)
            for (int i = 0; i <= MC_CTL_ABORT; ++i) __hip_atomic_store(a.ctl + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// the whole MLP of a small batch in one launch (k_mlp_chain); layers: planes, strides, widths.  act: two buffers of 3 x act_plane bf16.
bool mlp_chain_available(int n_layers, const int *k_pad, const int *n, int m_cap) {
    if (m_cap > 16 || n_layers < 2 || n_layers > MPE_MAX_MLP_LAYERS) return false;
    for (int l = 0; l < n_layers; ++l)
        if (k_pad[l] > lat::MC_WAVES * lat::MC_SPW * GEMM_BK || k_pad[l] % GEMM_BK) return false;
    for (int l = 1; l < n_layers; ++l)
        if (k_pad[l] != (n[l - 1] + 15) / 16 * 16 && k_pad[l] != (n[l - 1] + 31) / 32 * 32) return false;
    return true;
}

hipError_t launch_mlp_chain(hipStream_t s, int n_layers, const unsigned short *const *W3, const size_t *w_plane, const int *ldw, const float *const *bias,
                            const int *n, const float *A0, int lda0, unsigned short *act0, unsigned short *act1, size_t act_plane, int32_t *ctl,
                            int32_t *status, int m_cap, const int32_t *d_m, float slope, float *y, int ldy, const DecodeEpi *dec, int flush_stages,
                            int n_workgroups) {
    lat::ChainArgs a{};
    int nu_max = 1;
    for (int l = 0; l < n_layers; ++l) {
        a.L[l] = lat::ChainLayer{W3[l], w_plane[l], ldw[l], bias[l], n[l], l != n_layers - 1 ? 1 : 0};
        const int nu = (ldw[l] / GEMM_BK + flush_stages - 1) / flush_stages;
        nu_max = nu > nu_max ? nu : nu_max;
    }
    a.n_layers = n_layers;
    a.A0 = A0;
    a.lda0 = lda0;
    a.act[0] = act0;
    a.act[1] = act1;
    a.act_plane = act_plane;
    a.ctl = ctl;
    a.status = status;
    a.m_cap = m_cap;
    a.d_m = d_m;
    a.slope = slope;
    a.y = y;
    a.ldy = ldy;
    if (dec) a.dec = *dec;
    const size_t shm = (size_t)nu_max * 1024;
    const void *fn = flush_stages == 2 ? reinterpret_cast<const void *>(lat::k_mlp_chain<2>) : reinterpret_cast<const void *>(lat::k_mlp_chain<1>);
    if (shm > 48 * 1024) {
        static PerDeviceFlag attr[2];
        PerDeviceFlag &f = attr[flush_stages == 2 ? 0 : 1];
        if (!f.test()) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            f.set();
        }
    }
    if (flush_stages == 2) hipLaunchKernelGGL(lat::k_mlp_chain<2>, dim3((unsigned)n_workgroups), dim3(64 * lat::MC_WAVES), shm, s, a);
    else hipLaunchKernelGGL(lat::k_mlp_chain<1>, dim3((unsigned)n_workgroups), dim3(64 * lat::MC_WAVES), shm, s, a);
    return hipGetLastError();
}

}  // namespace mpe

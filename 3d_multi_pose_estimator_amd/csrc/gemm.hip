// fp32 MFMA "NT" GEMM with fused bias + LeakyReLU epilogue:  C = act(A * W^T + b).
//
// Replaces every nn.Linear(+LeakyReLU) call site of the path: GraphAttention2 fc1/fc2
// (reference gat2.py:53-55) and PoseEstimatorMLP (reference utils/mlp.py:8-28).
//
// Kernels in this file (one numerical definition, several shapes)
//   k_linear_dma    default: tile 128 x 80|64 x 32, K stages written into LDS by
//                   global_load_lds_dwordx4, 4 waves x (32 rows x all features)
//   k_linear_skinny / k_linear_skinny_ks   one wave per 16x16 tile for small batches
//   k_linear_bf16   reduced precision (BASELINE configs[4]), not on the parity path
// The default-path kernels (k_linear_dma, k_linear_skinny*) walk K in 32-deep stages; inside a
// stage lane group q of the MFMA takes k = 8q..8q+3 (first four MFMAs) then 8q+4..8q+7, so a row
// gives the same bits whichever of them, and whatever batch size, computed it (tested).  ACC64 adds each stage's fp32 result into f64 running sums.
//
// gfx950 design (the tile, fragment layout and epilogue of the round-1 register-staged kernel, whose code left the
// library in round 5 -- git history has it; k_linear_dma's own header describes the LDS image the DMA writes)
//   * v_mfma_f32_16x16x4_f32 (exact f32 fma chain, 256 FLOP/clk/CU): the model's widths
//     (400, 320, 160, 912, 3072, 2048, 1024) are multiples of 16, not of 32/128, so the
//     16-wide tile wastes <2 % where a 128-wide tile would waste up to 22 %.
//   * workgroup = 4 waves, tile 128 rows x 80 features x 32 deep; wave w owns rows
//     [32w,32w+32) x all 80 features = 2 x 5 MFMA tiles (40 accumulator VGPRs).
//   * operands are swapped (A-operand = weight rows, B-operand = activation rows) so each
//     lane ends up with 4 consecutive output features of one row: one 16-byte store.
//   * LDS image is planar in the MFMA k-quarter and dense: plane q holds, for every tile
//     row r, the 8 k-values {8q..8q+7} as two 16-byte halves stored at chunk 2r + (h ^ s(r)),
//     s(r) = bit3(r) ^ bit2(r).  With that swizzle a lane fetches its K-stage share with two
//     conflict-free ds_read_b128 (4 LDS cycles each) and the global->LDS pass (rows x chunks
//     transposed over the lanes) writes conflict-free ds_write_b128 (8 cycles each).
//   * double-buffered LDS (52 KB -> 3 workgroups/CU), register prefetch of the next K stage,
//     one barrier per stage; XCD-aware workgroup order so the tiles that share operands run
//     on one XCD's L2.
#include <cstdlib>

#include "mpe_internal.h"

namespace mpe {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int NT = GEMM_BN / 16;                // 5 feature tiles per wave
constexpr int MT = 2;                           // 2 row tiles per wave

// ---------------------------------------------------------------------------------------
// LDS-DMA variant: the K stages are written straight into LDS by `global_load_lds_dwordx4`
// (no VGPR round trip, no ds_write).  The LDS image is row-major and dense -- row r of a tile
// is one 128-byte line, its eight 16-byte chunks stored at chunk position c ^ g(r) with
// g(r) = bit1(r) | bit2(r) << 2 -- so one wave-instruction moves 8 whole 128-B rows (lane i ->
// row i/8, position i%8; the swizzle is applied on the per-lane SOURCE address) and the
// fragment reads stay conflict-free ds_read_b128.  Wave w stages exactly the 32 activation
// rows it consumes; the 80 weight rows are spread over the four waves.  One barrier per
// stage: [wait own DMA + barrier] -> issue DMA of the next stage into the other buffer ->
// 80 MFMAs on this one.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int dma_swz(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// A12 (fc2 of a graph-attention layer with 40-wide attention heads, 80-wide feature tiles): the
// epilogue also emits the attention coefficients of gat2.py:57-58,
//   a1[row][head] = <ft2[row, head, :], attn_l[head]>,  a2 with attn_r,
// from the result values the lanes already hold (a tile covers exactly two heads), so the
// attention stage does not have to read ft2 a second time.  Summation order (the canonical order,
// mirrored by coef40() in gat.hip for the paths that compute the coefficients elsewhere): lane
// group q = 0..3 of the MFMA layout runs one fma chain over the features it owns in ascending
// order, then (s_q + s_q^1) + (s_q^2 + s_q^3).
// LDR (the default since round 3): the K stages are requested by a FIFTH wave of the workgroup
// (the loader wave; 320-thread workgroups) instead of by the four MFMA waves.  Measured with
// tools/dma_interf.hip: a staging instruction (LDS-DMA or plain global load alike) issued from a
// wave that also streams MFMAs costs ~40 cycles of that wave's MFMA issue when it is alone on its
// SIMD and ~120 cycles of MFMA-pipe time with two such waves on the SIMD (6.5 of them per wave and
// stage: MFMA pipe busy 0.65-0.75), whereas the same requests issued by a wave that issues no MFMA
// leave the MFMA waves of its SIMD at exactly the cycle count of the staging-free loop (busy 1.000).
// The loader waits for its stage (vmcnt(0)) in front of the stage barrier all five waves share.
template <bool LEAKY, bool ACC64, int NTT, bool A12 = false, int NL = 0>
__global__ __launch_bounds__(256 + 64 * NL, NL == 4 ? ((ACC64 || A12) ? 4 : 6) : NL == 2 ? 5 : (NTT > 10 ? 1 : NTT > 5 ? 2 : 3)) void k_linear_dma(const float *__restrict__ A, int lda,
                                                       const float *__restrict__ W, int ldw,
                                                       const float *__restrict__ bias, float *__restrict__ C,
                                                       int ldc, int m_cap, const int32_t *__restrict__ d_m, int n,
                                                       int k_pad, float slope, int ntn, int n_major,
                                                       const int32_t *__restrict__ a_rows,
                                                       const int32_t *__restrict__ c_rows,
                                                       const float *__restrict__ attn_l = nullptr,
                                                       const float *__restrict__ attn_r = nullptr,
                                                       float *__restrict__ a12 = nullptr,
                                                       const int32_t *__restrict__ grp_count = nullptr, int n_grp = 0,
                                                       int grp_stride = 0, long w_grp_stride = 0, int out_half = 0) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];   // 2 stages of (128 + 16 NTT) rows x 128 B
    constexpr int ROWF = 32;                   // floats per tile row (dense)
    constexpr int W_OFF = GEMM_BM * ROWF;      // weight rows follow the activation rows
    constexpr int STAGE = (GEMM_BM + NTT * 16) * ROWF;

    int M = m_cap;
    if (d_m) {
        int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    int ntm = (M + GEMM_BM - 1) / GEMM_BM;
    if (grp_count) {
        // grouped GEMM: group g multiplies its own rows (row list g) with its own weight matrix;
        // every 128-row tile belongs to one group, the device-side counts give the tile ranges
        ntm = 0;
        for (int g = 0; g < n_grp; ++g) ntm += (grp_count[g] + GEMM_BM - 1) / GEMM_BM;
    }
    const int bid = blockIdx.x, nwg = ntm * ntn;
    if (bid >= nwg) return;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int tm, tn;
    if (n_major) {
        // weight panels outer, but in bands of 8 row tiles: the ~96 workgroups an XCD runs at a
        // time then cover about 8 row tiles x 12 panels (22 MB of operands) instead of 32 x 3
        // (52 MB), which halves the Infinity-Cache traffic of the MLP layers
        constexpr int RB = 8;
        const int band = swz / (RB * ntn), rem = swz - band * (RB * ntn);
        const int rows = ntm - band * RB < RB ? ntm - band * RB : RB;
        tn = rem / rows;
        tm = band * RB + (rem - tn * rows);
    } else {
        tm = swz / ntn;
        tn = swz - tm * ntn;
    }
    constexpr int BN = NTT * 16;
    if (grp_count) {
        int g = 0, t0 = 0;
        for (;;) {
            const int nt_g = (grp_count[g] + GEMM_BM - 1) / GEMM_BM;
            if (tm < t0 + nt_g || g == n_grp - 1) break;
            t0 += nt_g;
            ++g;
        }
        tm -= t0;
        M = grp_count[g];
        a_rows += (size_t)g * grp_stride;
        c_rows += (size_t)g * grp_stride;
        W += (size_t)g * w_grp_stride;
    }
    const int m0 = tm * GEMM_BM, n0 = tn * BN;

    constexpr bool LDR = NL > 0;
    const int tid = threadIdx.x, wave = LDR ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, lane = tid & 63;
    const int dr = lane >> 3, dp = lane & 7;          // DMA role: row within the 8-row group, chunk position
    const int nk = k_pad / GEMM_BK;

    if (LDR && wave >= 4) {
        // ---- loader wave(s): the 16 activation groups and 2*NTT weight groups of every stage, dealt
        // round-robin to the NL loader waves ----
        constexpr int NW = NTT * 2;
        constexpr int NLD = NL > 0 ? NL : 1;          // (NL == 0 never runs this block)
        constexpr int NA_L = 16 / NLD, NW_L = (NW + NLD - 1) / NLD;
        const int li = wave - 4;
        __builtin_amdgcn_s_setprio(3);         // +0.7 %: the loaders' requests are not queued behind older MFMA waves
        const float *la[NA_L], *lw[NW_L];
#pragma unroll
        for (int g = 0; g < NA_L; ++g) {
            const int row = (li * NA_L + g) * 8 + dr;
            int grow = m0 + row;
            grow = grow < M ? grow : M - 1;
            if (a_rows) grow = a_rows[grow];
            la[g] = A + (size_t)grow * lda + ((dp ^ dma_swz(row)) << 2);
        }
#pragma unroll
        for (int g = 0; g < NW_L; ++g) {
            int grp = li + NLD * g;
            if (grp > NW - 1) grp = NW - 1;
            const int row = grp * 8 + dr;
            lw[g] = W + (size_t)(n0 + row) * ldw + ((dp ^ dma_swz(row)) << 2);
        }
        auto fill = [&](int kt, int buf) {
            const int koff = kt * GEMM_BK;
            float *base = lds + buf * STAGE;
#pragma unroll
            for (int g = 0; g < NA_L; ++g)
                __builtin_amdgcn_global_load_lds((glb_void *)(la[g] + koff), (lds_void *)(base + (li * NA_L + g) * 8 * ROWF), 16, 0, 0);
#pragma unroll
            for (int g = 0; g < NW_L; ++g)
                if ((g + 1) * NLD <= NW || li + NLD * g < NW)
                    __builtin_amdgcn_global_load_lds((glb_void *)(lw[g] + koff), (lds_void *)(base + W_OFF + (li + NLD * g) * 8 * ROWF), 16, 0, 0);
        };
        fill(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                   // vmcnt(0): stage kt has landed; the MFMA waves are done with the other buffer
            if (kt + 1 < nk) fill(kt + 1, (kt + 1) & 1);
        }
        return;
    }

    // per-lane source pointers: 4 activation groups (rows 32w + 8g + dr), up to 3 weight groups
    const float *a_src[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = wave * 32 + g * 8 + dr;
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        if (a_rows) grow = a_rows[grow];           // gathered rows (grouped layer-0 GEMM)
        a_src[g] = A + (size_t)grow * lda + ((dp ^ dma_swz(row)) << 2);
    }
    // weight rows: 2*NTT groups of 8, dealt round-robin to the four waves
    constexpr int WG = NTT * 2, WPW = (WG + 3) / 4;
    const float *w_src[WPW];
#pragma unroll
    for (int g = 0; g < WPW; ++g) {
        int grp = wave + 4 * g;
        if (grp > WG - 1) grp = WG - 1;
        const int row = grp * 8 + dr;
        w_src[g] = W + (size_t)(n0 + row) * ldw + ((dp ^ dma_swz(row)) << 2);
    }

    const int fq = lane >> 4, fr = lane & 15;
    const int fsw = dma_swz(fr);
    int a_rd[MT], w_rd[NTT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * ROWF;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) w_rd[nt] = W_OFF + (nt * 16 + fr) * ROWF;
    const int c0 = ((fq * 2 + 0) ^ fsw) << 2, c1 = ((fq * 2 + 1) ^ fsw) << 2;

    f32x4 acc[NTT][MT];
    double run[ACC64 ? NTT : 1][ACC64 ? MT : 1][4];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ACC64) {
#pragma unroll
                for (int i = 0; i < 4; ++i) run[ACC64 ? nt : 0][ACC64 ? mt : 0][i] = 0.0;
            }
        }

    auto issue = [&](int kt, int buf) {
        if (LDR) return;                       // the loader wave stages
        const int koff = kt * GEMM_BK;
        float *base = lds + buf * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((glb_void *)(a_src[g] + koff),
                                             (lds_void *)(base + (wave * 32 + g * 8) * ROWF), 16, 0, 0);
#pragma unroll
        for (int g = 0; g < WPW; ++g)
            if ((g + 1) * 4 <= WG || wave + 4 * g < WG)
                __builtin_amdgcn_global_load_lds((glb_void *)(w_src[g] + koff),
                                                 (lds_void *)(base + W_OFF + (wave + 4 * g) * 8 * ROWF), 16, 0, 0);
    };

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // own DMA landed (vmcnt 0) + everybody's, and everybody
                                               // is done reading the other buffer
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const int cur = (kt & 1) * STAGE;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int co = hh ? c1 : c0;
            f32x4 af[MT], wf[NTT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4 *>(&lds[cur + a_rd[mt] + co]);
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) wf[nt] = *reinterpret_cast<const f32x4 *>(&lds[cur + w_rd[nt] + co]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s], af[mt][s], acc[nt][mt], 0, 0, 0);
        }
        if (ACC64) {
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) run[ACC64 ? nt : 0][ACC64 ? mt : 0][i] += (double)acc[nt][mt][i];
                    acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
    }

    float pl[2][MT], pr[2][MT];          // A12: per (head of the tile, row tile) partial dot products
    if (A12) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) pl[h][mt] = pr[h][mt] = 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
        const int nb = n0 + nt * 16 + fq * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nb);
        f32x4 al = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
        if (A12 && nb + 3 < n) {
            al = *reinterpret_cast<const f32x4 *>(attn_l + nb);
            ar = *reinterpret_cast<const f32x4 *>(attn_r + nb);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wave * 32 + mt * 16 + fr;
            f32x4 v;
            if (ACC64) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (float)(run[ACC64 ? nt : 0][ACC64 ? mt : 0][i] + (double)bv[i]);
            } else {
                v = acc[nt][mt] + bv;
            }
            if (LEAKY) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
            }
            if (A12) {
                // a 4-feature group never straddles the 40-feature head boundary of the tile
                const int hp = (nt * 16 + fq * 4) >= 40 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float l1 = __builtin_fmaf(v[i], al[i], hp ? pl[1][mt] : pl[0][mt]);
                    const float r1 = __builtin_fmaf(v[i], ar[i], hp ? pr[1][mt] : pr[0][mt]);
                    if (hp) { pl[1][mt] = l1; pr[1][mt] = r1; } else { pl[0][mt] = l1; pr[0][mt] = r1; }
                }
            }
            if (m >= M) continue;
            const int mo = c_rows ? c_rows[m] : m;
            if (out_half) {                    // fp16 rows for the attention stage (configs[4]); ldc counts halves
                _Float16 *dh = reinterpret_cast<_Float16 *>(C) + (size_t)mo * ldc + nb;
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                if (nb + 3 < n) {
                    *reinterpret_cast<h4 *>(dh) = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (nb + i < n) dh[i] = (_Float16)v[i];
                }
                continue;
            }
            float *dst = C + (size_t)mo * ldc + nb;
            if (nb + 3 < n) {
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < n) dst[i] = v[i];
            }
        }
    }
    if (A12) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float x = pl[h][mt], y = pr[h][mt];
                x = x + __shfl_xor(x, 16);
                y = y + __shfl_xor(y, 16);
                x = x + __shfl_xor(x, 32);
                y = y + __shfl_xor(y, 32);
                const int m = m0 + wave * 32 + mt * 16 + fr;
                const int head = (n0 / 40) + h;
                if (fq == 0 && m < M && head * 40 < n) {
                    const int mo = c_rows ? c_rows[m] : m;
                    a12[(size_t)mo * 32 + head] = x;
                    a12[(size_t)mo * 32 + 16 + head] = y;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------
// Latency variant for small batches (online use: one frame per call).  With a handful of rows a
// 128x80 tile leaves the chip empty and every workgroup walks K serially at 80 MFMAs per
// stage, most of them on padding.  Here one WAVE owns one 16x16 output tile and streams its
// 16 activation rows and 16 weight rows straight from global memory into MFMA fragments (no
// LDS, no barriers), SK_DEPTH K-stages in flight per wave.  The k order inside a 32-deep stage,
// the fp32 chain and the f64 flush cadence are exactly those of k_linear_dma, so a batch of one
// frame and a batch of a thousand give bit-identical rows.  Small M makes this kernel
// weight-bandwidth bound: every weight element is read once.
// ---------------------------------------------------------------------------------------
constexpr int SK_DEPTH = 8;

template <bool LEAKY, bool ACC64>
__global__ __launch_bounds__(256) void k_linear_skinny(const float *__restrict__ A, int lda,
                                                        const float *__restrict__ W, int ldw,
                                                        const float *__restrict__ bias, float *__restrict__ C,
                                                        int ldc, int m_cap, const int32_t *__restrict__ d_m, int n,
                                                        int k_pad, float slope, int nt16,
                                                        const int32_t *__restrict__ a_rows,
                                                        const int32_t *__restrict__ c_rows, int out_half) {
    int M = m_cap;
    if (d_m) {
        int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);      // the 4 waves of a block share activation rows
    const int tm = gw / nt16, tn = gw - tm * nt16;
    if (tm * 16 >= M) return;                                // no barriers below: waves may leave early
    const int fq = lane >> 4, fr = lane & 15;

    int grow = tm * 16 + fr;
    grow = grow < M ? grow : M - 1;
    if (a_rows) grow = a_rows[grow];
    const float *pa = A + (size_t)grow * lda + 8 * fq;       // stage kt: k = 32 kt + 8 fq + {0..7}
    const float *pw = W + (size_t)(tn * 16 + fr) * ldw + 8 * fq;

    const int nk = k_pad / GEMM_BK;
    f32x4 ra[SK_DEPTH][2], rw[SK_DEPTH][2];
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d) {
        const int ko = (d < nk ? d : nk - 1) * GEMM_BK;
        ra[d][0] = *reinterpret_cast<const f32x4 *>(pa + ko);
        ra[d][1] = *reinterpret_cast<const f32x4 *>(pa + ko + 4);
        rw[d][0] = *reinterpret_cast<const f32x4 *>(pw + ko);
        rw[d][1] = *reinterpret_cast<const f32x4 *>(pw + ko + 4);
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    double run[4] = {0.0, 0.0, 0.0, 0.0};
    auto stage = [&](const f32x4 &a0, const f32x4 &a1, const f32x4 &w0, const f32x4 &w1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[s], a0[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s], a1[s], acc, 0, 0, 0);
        if (ACC64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) run[i] += (double)acc[i];
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    int kt0 = 0;
    for (; kt0 + SK_DEPTH <= nk; kt0 += SK_DEPTH) {
#pragma unroll
        for (int d = 0; d < SK_DEPTH; ++d) {
            stage(ra[d][0], ra[d][1], rw[d][0], rw[d][1]);
            int kn = kt0 + d + SK_DEPTH;                     // refill the slot in place (clamped: no branch)
            kn = (kn < nk ? kn : nk - 1) * GEMM_BK;
            ra[d][0] = *reinterpret_cast<const f32x4 *>(pa + kn);
            ra[d][1] = *reinterpret_cast<const f32x4 *>(pa + kn + 4);
            rw[d][0] = *reinterpret_cast<const f32x4 *>(pw + kn);
            rw[d][1] = *reinterpret_cast<const f32x4 *>(pw + kn + 4);
        }
    }
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d)
        if (kt0 + d < nk) stage(ra[d][0], ra[d][1], rw[d][0], rw[d][1]);

    const int m = tm * 16 + fr;
    if (m >= M) return;
    const int nb = tn * 16 + fq * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nb);
    f32x4 v;
    if (ACC64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)(run[i] + (double)bv[i]);
    } else {
        v = acc + bv;
    }
    if (LEAKY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
    }
    const int mo = c_rows ? c_rows[m] : m;
    if (out_half) {
        _Float16 *dh = reinterpret_cast<_Float16 *>(C) + (size_t)mo * ldc + nb;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < n) dh[i] = (_Float16)v[i];
        return;
    }
    float *dst = C + (size_t)mo * ldc + nb;
    if (nb + 3 < n) {
        *reinterpret_cast<f32x4 *>(dst) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < n) dst[i] = v[i];
    }
}

// ---------------------------------------------------------------------------------------
// Latency variant with f64 running sums (the MLP at small batch): with ACC64 every 32-deep
// stage is its own fp32 MFMA chain, so the stages of one 16x16 tile are independent until the
// f64 sum.  The KS waves of a workgroup take the stages round-robin, park their fp32 stage
// results in LDS (1 KB per stage), and after one barrier each of 256 threads adds "its" element
// over the stages IN STAGE ORDER in f64 -- the same additions in the same order as
// k_linear_dma<ACC64>, hence the same bits, with the serial MFMA chain cut KS-fold.
// ---------------------------------------------------------------------------------------
constexpr int SKS_DEPTH = 4;

template <bool LEAKY, int KS>
__global__ __launch_bounds__(64 * KS) void k_linear_skinny_ks(const float *__restrict__ A, int lda,
                                                             const float *__restrict__ W, int ldw,
                                                             const float *__restrict__ bias, float *__restrict__ C,
                                                             int ldc, int m_cap, const int32_t *__restrict__ d_m,
                                                             int n, int k_pad, float slope, int nt16,
                                                             const int32_t *__restrict__ a_rows,
                                                             const int32_t *__restrict__ c_rows, int out_half) {
    extern __shared__ __attribute__((aligned(16))) float s_part[];       // [nk][64 lanes][4]
    int M = m_cap;
    if (d_m) {
        int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int tm = blockIdx.x / nt16, tn = blockIdx.x - tm * nt16;
    if (tm * 16 >= M) return;                                // whole workgroup leaves: no barrier reached
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15;
    int grow = tm * 16 + fr;
    grow = grow < M ? grow : M - 1;
    if (a_rows) grow = a_rows[grow];
    const float *pa = A + (size_t)grow * lda + 8 * fq;
    const float *pw = W + (size_t)(tn * 16 + fr) * ldw + 8 * fq;
    const int nk = k_pad / GEMM_BK;
    const int mine = nk > wave ? (nk - wave + KS - 1) / KS : 0;          // stages wave, wave+KS, ...

    f32x4 ra[SKS_DEPTH][2], rw[SKS_DEPTH][2];
    auto load = [&](int d, int i) {                          // i-th own stage (clamped: no branch)
        int ii = i < mine ? i : mine - 1;
        ii = ii < 0 ? 0 : ii;
        const int ko = (wave + ii * KS < nk ? wave + ii * KS : nk - 1) * GEMM_BK;
        ra[d][0] = *reinterpret_cast<const f32x4 *>(pa + ko);
        ra[d][1] = *reinterpret_cast<const f32x4 *>(pa + ko + 4);
        rw[d][0] = *reinterpret_cast<const f32x4 *>(pw + ko);
        rw[d][1] = *reinterpret_cast<const f32x4 *>(pw + ko + 4);
    };
#pragma unroll
    for (int d = 0; d < SKS_DEPTH; ++d) load(d, d);
    auto stage = [&](int d, int i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rw[d][0][s], ra[d][0][s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rw[d][1][s], ra[d][1][s], acc, 0, 0, 0);
        *reinterpret_cast<f32x4 *>(&s_part[((wave + i * KS) * 64 + lane) * 4]) = acc;
    };
    int i0 = 0;
    for (; i0 + SKS_DEPTH <= mine; i0 += SKS_DEPTH) {
#pragma unroll
        for (int d = 0; d < SKS_DEPTH; ++d) {
            stage(d, i0 + d);
            load(d, i0 + d + SKS_DEPTH);
        }
    }
#pragma unroll
    for (int d = 0; d < SKS_DEPTH; ++d)
        if (i0 + d < mine) stage(d, i0 + d);
    __syncthreads();

    const int t = threadIdx.x;
    if (t >= 256) return;
    double run = 0.0;
    for (int s = 0; s < nk; ++s) run += (double)s_part[s * 256 + t];     // stage order, as the tile kernel
    const int el = t >> 2, i = t & 3;                        // element (lane el, component i) of the MFMA tile
    const int m = tm * 16 + (el & 15), nb = tn * 16 + (el >> 4) * 4 + i;
    if (m >= M || nb >= n) return;
    float v = (float)(run + (double)bias[nb]);
    if (LEAKY) v = v > 0.f ? v : v * slope;
    const int mo = c_rows ? c_rows[m] : m;
    if (out_half) reinterpret_cast<_Float16 *>(C)[(size_t)mo * ldc + nb] = (_Float16)v;
    else C[(size_t)mo * ldc + nb] = v;
}

// ---------------------------------------------------------------------------------------
// bf16 variant (reduced precision, BASELINE.json configs[4]: "bf16 MLP MFMA GEMM"):
// C = act(bf16(A) * W_bf16^T + b) with fp32 accumulation on v_mfma_f32_16x16x32_bf16
// (16x the fp32 MFMA rate).  Activations stay fp32 in memory and are rounded to bf16 while
// they are staged (v_cvt_pk_bf16_f32); weights are converted once at upload.  Tile 128x80x128:
// a 256-B LDS row holds the 128 k of one tile row as sixteen 16-byte chunks stored at chunk
// position c ^ (row & 15), which makes both the ds_write_b128 of the staging pass and the
// ds_read_b128 fragment reads conflict-free.  Single LDS buffer (52 KB, 3 WG/CU), register
// prefetch of the next stage, two barriers per stage.  Not used on the parity path.
// ---------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) short;
constexpr int BF_BK = 128;
constexpr int BF_ROWB = BF_BK * 2;                       // bytes per tile row
constexpr int BF_A_PASSES = GEMM_BM * 16 / 256;          // 16 chunks per row, 256 threads -> 8
constexpr int BF_W_PASSES = GEMM_BN * 16 / 256;          // 5

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v = {(__bf16)lo, (__bf16)hi};                    // v_cvt_pk_bf16_f32, round to nearest even
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return u;
}

template <bool LEAKY, bool OUT16>
__global__ __launch_bounds__(256, 2) void k_linear_bf16(const float *__restrict__ A, int lda,
                                                        const unsigned short *__restrict__ Wb, int ldw,
                                                        const float *__restrict__ bias, float *__restrict__ C,
                                                        int ldc, int m_cap, const int32_t *__restrict__ d_m, int n,
                                                        int k_pad, float slope, int ntn, int n_major, int k_lim,
                                                        const int32_t *__restrict__ a_rows,
                                                        const int32_t *__restrict__ c_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[(GEMM_BM + GEMM_BN) * BF_ROWB];
    int M = m_cap;
    if (d_m) {
        int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int ntm = (M + GEMM_BM - 1) / GEMM_BM;
    const int bid = blockIdx.x, nwg = ntm * ntn;
    if (bid >= nwg) return;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int tm, tn;
    if (n_major) {
        tn = swz / ntm;
        tm = swz - tn * ntm;
    } else {
        tm = swz / ntn;
        tn = swz - tm * ntn;
    }
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sc = tid & 15, sr = tid >> 4;              // staging role: chunk 0..15, row 0..15 (+16 per pass)

    const float *a_src[BF_A_PASSES];
    int a_dst[BF_A_PASSES];
#pragma unroll
    for (int p = 0; p < BF_A_PASSES; ++p) {
        const int row = p * 16 + sr;
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        if (a_rows) grow = a_rows[grow];
        a_src[p] = A + (size_t)grow * lda + sc * 8;
        a_dst[p] = row * BF_ROWB + ((sc ^ (row & 15)) << 4);
    }
    const unsigned short *w_src[BF_W_PASSES];
    int w_dst[BF_W_PASSES];
#pragma unroll
    for (int p = 0; p < BF_W_PASSES; ++p) {
        const int row = p * 16 + sr;
        w_src[p] = Wb + (size_t)(n0 + row) * ldw + sc * 8;
        w_dst[p] = (GEMM_BM + row) * BF_ROWB + ((sc ^ (row & 15)) << 4);
    }
    const int fq = lane >> 4, fr = lane & 15;
    int a_rd[MT], w_rd[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * BF_ROWB;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w_rd[nt] = (GEMM_BM + nt * 16 + fr) * BF_ROWB;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 pa[BF_A_PASSES], pw[BF_W_PASSES];
    const int nk = k_pad / BF_BK;

    auto load = [&](int kt) {
        const int koff = kt * BF_BK;
        // activation rows are valid (and zero padded) up to k_lim columns only; the rest of the
        // 128-deep stage is zeros
        const bool in_row = koff + sc * 8 < k_lim;
#pragma unroll
        for (int p = 0; p < BF_A_PASSES; ++p) {
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            if (in_row) {
                lo = *reinterpret_cast<const f32x4 *>(a_src[p] + koff);
                hi = *reinterpret_cast<const f32x4 *>(a_src[p] + koff + 4);
            }
            pa[p] = (u32x4){pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]),
                            pack_bf16(hi[2], hi[3])};
        }
#pragma unroll
        for (int p = 0; p < BF_W_PASSES; ++p) pw[p] = *reinterpret_cast<const u32x4 *>(w_src[p] + koff);
    };
    auto store = [&]() {
#pragma unroll
        for (int p = 0; p < BF_A_PASSES; ++p) *reinterpret_cast<u32x4 *>(&lds[a_dst[p]]) = pa[p];
#pragma unroll
        for (int p = 0; p < BF_W_PASSES; ++p) *reinterpret_cast<u32x4 *>(&lds[w_dst[p]]) = pw[p];
    };

    load(0);
    for (int kt = 0; kt < nk; ++kt) {
        store();
        __syncthreads();
        if (kt + 1 < nk) load(kt + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int co = ((s * 4 + fq) ^ fr) << 4;
            bf16x8 af[MT], wf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const bf16x8 *>(&lds[a_rd[mt] + co]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const bf16x8 *>(&lds[w_rd[nt] + co]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nb = n0 + nt * 16 + fq * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + nb);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wave * 32 + mt * 16 + fr;
            if (m >= M) continue;
            f32x4 v = acc[nt][mt] + bv;
            if (LEAKY) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
            }
            const int mo = c_rows ? c_rows[m] : m;
            if (OUT16) {                       // fp16 rows for the attention stage (ldc in halves)
                _Float16 *dst = reinterpret_cast<_Float16 *>(C) + (size_t)mo * ldc + nb;
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                if (nb + 3 < n) {
                    *reinterpret_cast<h4 *>(dst) = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (nb + i < n) dst[i] = (_Float16)v[i];
                }
                continue;
            }
            float *dst = C + (size_t)mo * ldc + nb;
            if (nb + 3 < n) {
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < n) dst[i] = v[i];
            }
        }
    }
}

hipError_t launch_linear_bf16(hipStream_t s, const float *A, int lda, const unsigned short *Wb, int ldw,
                              const float *bias, float *C, int ldc, int m_cap, const int32_t *d_m, int n, int k_pad,
                              bool leaky, float slope, int k_lim, bool out_half, const int32_t *a_rows,
                              const int32_t *c_rows) {
    if (m_cap <= 0 || n <= 0) return hipSuccess;
    const int ntm = (m_cap + GEMM_BM - 1) / GEMM_BM;
    const int ntn = (n + GEMM_BN - 1) / GEMM_BN;
    const int n_major = (size_t)n * k_pad * 2 > (size_t)(2u << 20) ? 1 : 0;
    if (k_lim <= 0 || k_lim > k_pad) k_lim = k_pad;
    dim3 grid(ntm * ntn), block(256);
#define MPE_LAUNCH_BF(L_, H_)                                                                                    \
    hipLaunchKernelGGL((k_linear_bf16<L_, H_>), grid, block, 0, s, A, lda, Wb, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, \
                       slope, ntn, n_major, k_lim, a_rows, c_rows)
    if (leaky && out_half) MPE_LAUNCH_BF(true, true);
    else if (leaky) MPE_LAUNCH_BF(true, false);
    else if (out_half) MPE_LAUNCH_BF(false, true);
    else MPE_LAUNCH_BF(false, false);
#undef MPE_LAUNCH_BF
    return hipGetLastError();
}

static size_t dma_lds_bytes(int ntt) { return (size_t)2 * (GEMM_BM + ntt * 16) * 32 * sizeof(float); }

bool linear_uses_tile_kernel(int m_cap, int n) {
    static const int skinny_waves = getenv("MPE_SKINNY_WAVES") ? atoi(getenv("MPE_SKINNY_WAVES")) : 1024;
    return (long)((m_cap + 15) / 16) * ((n + 15) / 16) > skinny_waves;
}

// ---------------------------------------------------------------------------------------
// Launch side of k_linear_dma.
//
// NL = loader waves per workgroup (0, 2 or 4; the template header says why they exist).  What
// each instantiation gets, measured on the default bench (profiles/r03_gemm_loader_waves.txt):
//   plain fp32 chain, NTT 4|5      NL 4   65 VGPRs, 3 workgroups x 8 waves per CU       -4 %
//   fc2 + attention coefficients   NL 4   95 VGPRs (the epilogue's partial sums), 2 workgroups x 8 waves per CU: -5 %;
//                                         3 x 8 waves needs 80 VGPRs and spills (+3.5 %), two loader waves +6 %
//   f64 running sums, NTT 4 (MLP)  NL 4   128 VGPRs, 2 workgroups x 8 waves per CU      -6 %
//   f64 running sums, NTT 5        NL 0   160 VGPRs: the round-2 form (layer-0 fc2, head rows only)
// (MPE_GEMM_LOADER=0, the round-2 form everywhere, and the wide tiles of MPE_GEMM_BN left the library in round 5.)
// ---------------------------------------------------------------------------------------
struct DmaLaunch {
    const float *A;
    int lda;
    const float *W;
    int ldw;
    const float *bias;
    float *C;
    int ldc, m_cap;
    const int32_t *d_m;
    int n, k_pad;
    float slope;
    int ntn, n_major;
    const int32_t *a_rows, *c_rows;
    const float *attn_l = nullptr, *attn_r = nullptr;
    float *a12 = nullptr;
    const int32_t *grp_count = nullptr;
    int n_grp = 0, grp_stride = 0;
    long w_grp_stride = 0;
    int out_half = 0;          // fc2 of a graph-attention layer in the fp16-attention mode: C holds fp16 rows, ldc in halves
};

template <bool L, bool A64, int N, bool A12, int NL>
static void launch_dma(hipStream_t s, int grid, const DmaLaunch &a) {
    hipLaunchKernelGGL((k_linear_dma<L, A64, N, A12, NL>), dim3(grid), dim3(256 + 64 * NL), dma_lds_bytes(N), s, a.A, a.lda, a.W,
                       a.ldw, a.bias, a.C, a.ldc, a.m_cap, a.d_m, a.n, a.k_pad, a.slope, a.ntn, a.n_major, a.a_rows, a.c_rows,
                       a.attn_l, a.attn_r, a.a12, a.grp_count, a.n_grp, a.grp_stride, a.w_grp_stride, a.out_half);
}

// plain fp32 chain (no f64 sums, no coefficient epilogue), NTT 4 or 5
template <int N>
static void launch_dma_plain(hipStream_t s, int grid, const DmaLaunch &a, bool leaky) {
    if (leaky) launch_dma<true, false, N, false, 4>(s, grid, a);
    else launch_dma<false, false, N, false, 4>(s, grid, a);
}

template <int N>
static void launch_dma_acc64(hipStream_t s, int grid, const DmaLaunch &a, bool leaky) {
    if constexpr (N == 4) {
        if (leaky) launch_dma<true, true, 4, false, 4>(s, grid, a);
        else launch_dma<false, true, 4, false, 4>(s, grid, a);
    } else {
        if (leaky) launch_dma<true, true, N, false, 0>(s, grid, a);
        else launch_dma<false, true, N, false, 0>(s, grid, a);
    }
}

hipError_t launch_linear_grouped(hipStream_t s, const float *A, int lda, const float *W, int ldw, long w_grp_stride,
                                 const float *bias, float *C, int ldc, int m_cap, const int32_t *grp_count, int n_grp,
                                 const int32_t *row_lists, int grp_stride, int n, int k_pad, bool leaky, float slope, bool acc64) {
    if (m_cap <= 0 || n <= 0 || n_grp <= 0) return hipSuccess;
    const int ntm_cap = (m_cap + GEMM_BM - 1) / GEMM_BM + n_grp;       // sum of per-group ceilings
    // (f64 running sums -- the GAT's explicit f64-sum mode -- on 64-wide tiles, as launch_linear: the per-camera launches of a small
    // batch take that form too, so a row keeps its bits whatever the batch)
    DmaLaunch a{A, lda, W, ldw, bias, C, ldc, m_cap, nullptr, n, k_pad, slope, acc64 ? (n + 63) / 64 : (n + 79) / 80, 0, row_lists, row_lists};
    a.grp_count = grp_count;
    a.n_grp = n_grp;
    a.grp_stride = grp_stride;
    a.w_grp_stride = w_grp_stride;
    if (acc64) launch_dma_acc64<4>(s, ntm_cap * a.ntn, a, leaky);
    else launch_dma_plain<5>(s, ntm_cap * a.ntn, a, leaky);
    return hipGetLastError();
}

hipError_t launch_linear(hipStream_t s, const float *A, int lda, const float *W, int ldw, const float *bias,
                         float *C, int ldc, int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky,
                         float slope, bool acc64, const int32_t *a_rows, const int32_t *c_rows,
                         const AttnCoef *coef, bool *coef_done, bool out_half) {
    if (coef_done) *coef_done = false;
    if (m_cap <= 0 || n <= 0) return hipSuccess;
    const int ntm = (m_cap + GEMM_BM - 1) / GEMM_BM;
    const int n_major = (size_t)n * k_pad * sizeof(float) > (size_t)(2u << 20) ? 1 : 0;
    // small batches: one wave per 16x16 tile while that still leaves SIMDs idle (k_linear_skinny)
    static const int skinny_waves = getenv("MPE_SKINNY_WAVES") ? atoi(getenv("MPE_SKINNY_WAVES")) : 1024;
    const int nt16 = (n + 15) / 16;
    const long waves16 = (long)((m_cap + 15) / 16) * nt16;
    // narrow outputs (one 64-wide feature tile would do <= 1/4 useful work: the 54-wide last MLP layer, the 1-wide
    // last fc2 of the GAT) stay on the wave-per-16x16-tile kernels at any batch size: same bits by construction
    static const int narrow_on = getenv("MPE_GEMM_NARROW") ? atoi(getenv("MPE_GEMM_NARROW")) : 1;
    const bool narrow = narrow_on && nt16 <= (acc64 ? 4 : 1);
    if ((waves16 <= skinny_waves || narrow) && acc64 && k_pad / GEMM_BK <= 128 && k_pad / GEMM_BK >= 8) {
        // K split over the waves of a workgroup, ordered f64 reduction through LDS
        const size_t shm = (size_t)(k_pad / GEMM_BK) * 1024;
        static PerDeviceFlag attr_done;
        if (!attr_done.test()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_skinny_ks<true, 8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_skinny_ks<false, 8>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            attr_done.set();
        }
        dim3 kgrid((unsigned)waves16), kblock(512);
        if (leaky)
            hipLaunchKernelGGL((k_linear_skinny_ks<true, 8>), kgrid, kblock, shm, s, A, lda, W, ldw, bias, C, ldc, m_cap,
                               d_m, n, k_pad, slope, nt16, a_rows, c_rows, out_half ? 1 : 0);
        else
            hipLaunchKernelGGL((k_linear_skinny_ks<false, 8>), kgrid, kblock, shm, s, A, lda, W, ldw, bias, C, ldc, m_cap,
                               d_m, n, k_pad, slope, nt16, a_rows, c_rows, out_half ? 1 : 0);
        return hipGetLastError();
    }
    if (waves16 <= skinny_waves || narrow) {
        dim3 sgrid((unsigned)((waves16 + 3) / 4)), sblock(256);
#define MPE_LAUNCH_SK(L_, A_)                                                                                  \
    hipLaunchKernelGGL((k_linear_skinny<L_, A_>), sgrid, sblock, 0, s, A, lda, W, ldw, bias, C, ldc, m_cap, d_m, n, \
                       k_pad, slope, nt16, a_rows, c_rows, out_half ? 1 : 0)
        if (leaky && acc64) MPE_LAUNCH_SK(true, true);
        else if (leaky) MPE_LAUNCH_SK(true, false);
        else if (acc64) MPE_LAUNCH_SK(false, true);
        else MPE_LAUNCH_SK(false, false);
#undef MPE_LAUNCH_SK
        return hipGetLastError();
    }
    // Feature-tile width: 64 or 80, whichever leaves fewer (tiles on the busiest CU) x width; the MLP layers at a few thousand
    // person rows balance exactly with 64.  (160- and 208-wide tiles -- fewer staged bytes and DMA requests per MFMA, +10 % in the
    // isolated loop of tools/mfma_peak.hip -- did not pay in the full kernel and left the library in round 5.)  The padded weight
    // rows cover any tile that starts below n.
    DmaLaunch a{A, lda, W, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, slope, 0, n_major, a_rows, c_rows};
    a.out_half = out_half ? 1 : 0;
    // attention coefficients in the epilogue: 40-wide heads on 80-wide tiles (two heads per tile)
    if (coef && coef->out_dim == 40 && n == coef->heads * 40 && !leaky) {
        a.ntn = (n + 79) / 80;
        a.attn_l = coef->attn_l;
        a.attn_r = coef->attn_r;
        a.a12 = coef->a12;
        const int grid = ntm * a.ntn;
        if (acc64) launch_dma<false, true, 5, true, 0>(s, grid, a);
        else launch_dma<false, false, 5, true, 4>(s, grid, a);
        if (coef_done) *coef_done = true;
        return hipGetLastError();
    }
    const int tiles64 = ntm * ((n + 63) / 64), tiles80 = ntm * ((n + 79) / 80);
    const bool wide = (double)((tiles80 + 255) / 256) * 80 < (double)((tiles64 + 255) / 256) * 64;        // (a tie goes to 64)
    a.ntn = wide ? (n + 79) / 80 : (n + 63) / 64;
    const int grid = ntm * a.ntn;
    if (!wide) {
        if (acc64) launch_dma_acc64<4>(s, grid, a, leaky);
        else launch_dma_plain<4>(s, grid, a, leaky);
    } else {
        if (acc64) launch_dma_acc64<5>(s, grid, a, leaky);
        else launch_dma_plain<5>(s, grid, a, leaky);
    }
    return hipGetLastError();
}

}  // namespace mpe

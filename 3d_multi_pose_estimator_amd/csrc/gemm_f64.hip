// nn.Linear (+ LeakyReLU) with EXACT products and f64 accumulation on the f64 matrix pipe: C = act(A W^T + b), fp32 in, fp32 out.
//
// The reference-exact form of the MLP (utils/mlp.py:8-28; mpe_set_precision MLP 5): every fp32 x fp32 product is exact in f64 and
// the whole K sum is carried in f64 by v_mfma_f64_16x16x4_f64, so a layer's output is the correctly rounded fp32 of its exact value
// up to the f64 summation order (relative 1e-16) -- the network "evaluated in f64 with fp32 rounding between layers" that the
// parity tests call `exact` (tests/test_gpu_stages.py::_exact_mlp), which the reference's own torch-CPU MLP sits 1.6-6e-3 mm away
// from.  This is the mode in which the north star's "3D joints within 1e-3 mm" holds against the exact network on every rig
// (test_mlp_within_1e_3_mm_of_the_exact_network).  It is NOT the fast path: the f64 matrix pipe peaks at 78.6 TFLOP/s, the MLP
// launches take ~3x the split-bf16 form's time (bench.py reports both: 163 k against 267 k frames/s).
//
// One kernel for every batch size (a row has the same bits alone and in a batch of thousands: k ascending in steps of four, four
// products per instruction).  Tile 128 rows x 64 features x 32 deep, four waves of 32 rows x 64 features (8 accumulator tiles =
// 64 registers); operands staged through LDS as fp32 with a 36-float row stride (the column reads of the MFMA layout -- lane =
// (k, row) -- then touch 64 different banks) and widened to f64 in registers; next stage prefetched into registers while this one
// is multiplied.  Lane layout (MI355X_MICROARCH.md): A / B operand lane l = (row l & 15, k l >> 4), one f64 each; result register
// i of lane l = (row (l >> 4) + 4 i, column l & 15); operands swapped as in the other GEMMs (A operand = weight rows), so a lane
// holds features (l >> 4) + 4 i of activation row l & 15.
#include "mpe_internal.h"

namespace mpe {
namespace f64mm {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) double f64x4;

constexpr int BM = 128, BN = 64, BK = 32, LDSF = 36;      // LDS row stride in floats

template <bool LEAKY>
__global__ __launch_bounds__(256, 2) void k_linear_f64(const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
                                                       const float *__restrict__ bias, float *__restrict__ C, int ldc, int m_cap,
                                                       const int32_t *__restrict__ d_m, int n, int k_pad, double slope, int ntn) {
    __shared__ __attribute__((aligned(16))) float s_a[BM * LDSF];
    __shared__ __attribute__((aligned(16))) float s_w[BN * LDSF];
    int M = m_cap;
    if (d_m) {
        const int dm = *d_m;
        M = dm < m_cap ? dm : m_cap;
    }
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;                                   // whole workgroup leaves: no barrier reached
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fk = lane >> 4;
    // staging: thread t moves float4 (row t / 8 + 32 j, chunk t % 8) of the activation tile (j = 0..3) and of the weight tile (j = 0..1)
    const int sr = tid >> 3, sc = (tid & 7) * 4;
    const float *pa[4], *pw[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int row = m0 + sr + 32 * j;
        row = row < M ? row : M - 1;                       // rows behind the batch repeat its last row (never stored)
        pa[j] = A + (size_t)row * lda + sc;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) pw[j] = W + (size_t)(n0 + sr + 32 * j) * ldw + sc;      // (weight rows are padded far beyond n)
    f32x4 ra[4], rw[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4 *>(pa[j] + k0);
#pragma unroll
        for (int j = 0; j < 2; ++j) rw[j] = *reinterpret_cast<const f32x4 *>(pw[j] + k0);
    };
    auto park = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(&s_a[(sr + 32 * j) * LDSF + sc]) = ra[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(&s_w[(sr + 32 * j) * LDSF + sc]) = rw[j];
    };
    f64x4 acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int nk = k_pad / BK;
    fetch(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                                   // everybody is done reading the previous stage
        park();
        __syncthreads();
        if (kt + 1 < nk) fetch((kt + 1) * BK);             // in flight while this stage is multiplied
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            double ad[2], wd[4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ad[mt] = (double)s_a[(wave * 32 + mt * 16 + fr) * LDSF + 4 * s + fk];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wd[nt] = (double)s_w[(nt * 16 + fr) * LDSF + 4 * s + fk];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wd[nt], ad[mt], acc[nt][mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wave * 32 + mt * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = n0 + nt * 16 + fk + 4 * i;
                if (f >= n) continue;
                double v = acc[nt][mt][i] + (double)bias[f];
                if (LEAKY) v = v > 0.0 ? v : v * slope;               // in f64 as well: one rounding per output, at the very end
                C[(size_t)m * ldc + f] = (float)v;
            }
    }
}

}  // namespace f64mm

hipError_t launch_linear_f64(hipStream_t s, const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc,
                             int m_cap, const int32_t *d_m, int n, int k_pad, bool leaky, float slope) {
    if (m_cap <= 0 || n <= 0) return hipSuccess;
    const int ntm = (m_cap + f64mm::BM - 1) / f64mm::BM, ntn = (n + f64mm::BN - 1) / f64mm::BN;
    const dim3 grid((unsigned)(ntm * ntn)), block(256);
    // The slope arrives as the fp32 number the C ABI carries; the network's own parameter is the decimal the caller wrote
    // (nn.LeakyReLU(0.1): utils/mlp.py:11), which an f64 evaluation of the NETWORK multiplies by as a double.  Take that decimal
    // (seven digits) when it rounds to the fp32 value given, the fp32 value itself otherwise.  This is the rule include/mpe.h
    // documents for MLP mode 5: the mode evaluates the network in f64 ("f64-evaluated network", what tests call the exact network),
    // it does not replay the reference's fp32 kernel, which multiplies by (float)0.1.
    double slope_d = (double)slope;
    const double dec = __builtin_nearbyint(slope_d * 1e7) / 1e7;
    if ((float)dec == slope) slope_d = dec;
    if (leaky)
        hipLaunchKernelGGL((f64mm::k_linear_f64<true>), grid, block, 0, s, A, lda, W, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, slope_d, ntn);
    else
        hipLaunchKernelGGL((f64mm::k_linear_f64<false>), grid, block, 0, s, A, lda, W, ldw, bias, C, ldc, m_cap, d_m, n, k_pad, slope_d, ntn);
    return hipGetLastError();
}

}  // namespace mpe

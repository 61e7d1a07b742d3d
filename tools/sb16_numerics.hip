// Numerical experiment for the split-bf16 GEMM idea (companion of tools/sb16_gemm.hip): how far from the
// exactly evaluated dot products does each arithmetic land, on operands shaped like the MLP's (K up to 3072)?
//
//   fp32 MFMA chain + f64 flush per 32-deep stage      the production arithmetic of the MLP launches (k_linear_dma<ACC64>)
//   fp32 MFMA chain, no flush                          the production arithmetic of the GAT launches
//   split-bf16, 6 products, f64 flush every S stages   a = a1 + a2 + a3 (bf16 each, exact), products a_i w_j with i + j <= 4 on
//                                                      v_mfma_f32_16x16x32_bf16, fp32 accumulators; S = 1, 2, 4, 8, never
//   split-bf16, 3 products                             a1w1 + a1w2 + a2w1
//
// One wave computes one 16 x 16 tile; 64 tiles with independent random operands.  Errors in units of the fp32 ulp of the
// result's magnitude scale (the largest |y| of the tile), max and rms over all outputs.
//   hipcc --offload-arch=gfx950 -O3 tools/sb16_numerics.hip -o tools/sb16_numerics && tools/sb16_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

__device__ __forceinline__ unsigned short bf16_rn(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// A [16][K] rows, W [16][K] features (both fp32, per tile).  mode: 0 = fp32 MFMA; 1 = split-bf16 6 products; 2 = 3 products.
// flush = stages (of 32 k) between f64 flushes; 0 = never
__global__ __launch_bounds__(64) void k_tile(const float *A, const float *W, int K, int mode, int flush, double *out) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const float *a = A + (size_t)tile * 16 * K, *w = W + (size_t)tile * 16 * K;
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    double run[4] = {0, 0, 0, 0};
    int since = 0;
    for (int k0 = 0; k0 < K; k0 += 32) {
        if (mode == 0) {
            // the order of k_linear_dma inside a stage: two halves of 16, lane group q owns k = 4q .. 4q+3 of each half, step s
            // multiplies k = half * 16 + 4 q' ... -- one chain per accumulator, 8 MFMAs of 4 k each
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k = k0 + s * 4 + q;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(size_t)r * K + k], a[(size_t)r * K + k], acc, 0, 0, 0);
            }
        } else {
            bf16x8 af[3], wf[3];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + q * 8 + j;
                float x = a[(size_t)r * K + k], y = w[(size_t)r * K + k];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned short hx = bf16_rn(x), hy = bf16_rn(y);
                    af[p][j] = (short)hx;
                    wf[p][j] = (short)hy;
                    x -= bf16_f(hx);                                   // exact
                    y -= bf16_f(hy);
                }
            }
            const int np = mode == 1 ? 3 : 2;
            // least significant products first
            for (int s = np - 1; s >= 0; --s)                         // s = pa + pw
                for (int pa = 0; pa <= s; ++pa) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s - pa], af[pa], acc, 0, 0, 0);
        }
        if (flush > 0 && ++since == flush) {
#pragma unroll
            for (int i = 0; i < 4; ++i) run[i] += (double)acc[i];
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            since = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[((size_t)tile * 16 + (q * 4 + i)) * 16 + r] = run[i] + (double)acc[i];     // [feature][row]
}

static double urand(unsigned long long &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
}

int main() {
    const int T = 64;
    const int Ks[3] = {1280, 3072, 416};
    for (int K : Ks) {
        std::vector<float> A((size_t)T * 16 * K), W((size_t)T * 16 * K);
        unsigned long long seed = 12345 + K;
        // activations after LeakyReLU(0.1): mostly positive, magnitude ~1; weights uniform in +-1/sqrt(K) x 3 (unit-variance outputs)
        for (auto &v : A) { double u = urand(seed) * 2 - 0.6; v = (float)(u > 0 ? u : 0.1 * u); }
        const double wb = 3.0 / std::sqrt((double)K);
        for (auto &v : W) v = (float)((urand(seed) * 2 - 1) * wb);
        std::vector<double> exact((size_t)T * 256);
        double scale = 0;
        for (int t = 0; t < T; ++t)
            for (int f = 0; f < 16; ++f)
                for (int r = 0; r < 16; ++r) {
                    long double s = 0;
                    for (int k = 0; k < K; ++k) s += (long double)W[((size_t)t * 16 + f) * K + k] * (long double)A[((size_t)t * 16 + r) * K + k];
                    exact[((size_t)t * 16 + f) * 16 + r] = (double)s;
                    scale = std::fmax(scale, std::fabs((double)s));
                }
        const double ulp = std::ldexp(1.0, (int)std::floor(std::log2(scale)) - 23);
        float *dA, *dW;
        double *dO;
        hipMalloc(&dA, A.size() * 4);
        hipMalloc(&dW, W.size() * 4);
        hipMalloc(&dO, exact.size() * 8);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        printf("K = %d, largest |y| %.3f, fp32 ulp at that scale %.3e; errors of the result ROUNDED TO fp32, in those ulps (max / rms over %d outputs)\n", K, scale, ulp, T * 256);
        struct V { const char *name; int mode, flush; } vs[] = {
            {"fp32 MFMA chain, f64 flush per stage (production MLP)", 0, 1}, {"fp32 MFMA chain, no flush (production GAT)", 0, 0},
            {"split-bf16 x6, f64 flush per stage", 1, 1}, {"split-bf16 x6, flush every 2 stages", 1, 2}, {"split-bf16 x6, flush every 4 stages", 1, 4},
            {"split-bf16 x6, flush every 8 stages", 1, 8}, {"split-bf16 x6, no flush", 1, 0}, {"split-bf16 x3, f64 flush per stage", 2, 1}};
        std::vector<double> got(exact.size());
        for (auto &v : vs) {
            hipLaunchKernelGGL(k_tile, dim3(T), dim3(64), 0, 0, dA, dW, K, v.mode, v.flush, dO);
            hipMemcpy(got.data(), dO, got.size() * 8, hipMemcpyDeviceToHost);
            double mx = 0, sq = 0;
            for (size_t i = 0; i < got.size(); ++i) {
                const double e = ((double)(float)got[i] - exact[i]) / ulp;
                mx = std::fmax(mx, std::fabs(e));
                sq += e * e;
            }
            printf("  %-58s max %8.2f  rms %7.3f\n", v.name, mx, std::sqrt(sq / got.size()));
        }
        hipFree(dA);
        hipFree(dW);
        hipFree(dO);
    }
    return 0;
}

// Numerical experiment for the 32x32x16 form of the split-bf16 GEMM (companion of tools/sb16_numerics.hip):
// v_mfma_f32_32x32x16_bf16 adds 16 products per instruction where v_mfma_f32_16x16x32_bf16 adds 32, so a 32-deep K stage becomes
// two instructions per plane pair and the in-stage summation order changes.  Same operands, same error measure as
// sb16_numerics.hip (errors of the result rounded to fp32, in fp32 ulps of the largest |y|, max / rms over 16 384 outputs
// against long-double dot products).
//
//   inst   16 = v_mfma_f32_16x16x32_bf16 (one wave per 16 x 16 tile), 32 = v_mfma_f32_32x32x16_bf16 (one wave per 32 x 32 tile)
//   order  0 = "half-major": the six products of k-half 0, then the six of k-half 1 (a 16-deep stage in the canonical order)
//          1 = "plane-major": per plane pair both halves, least significant pair first
//          (16x16x32 has one order: the canonical six)
//   acc    0 = one fp32 chain per flush interval (production MLP form)
//          1 = even / odd stages in two chains, added at the end (production GAT form; flush = 0)
//          2 = SPLIT ACCUMULATORS: the five low-order products (<= 2^-8 of the leading one) in a chain that is never flushed,
//              the leading product a1 w1 in its own chain flushed every `flush` stages -- the roundings that matter are then one
//              per leading MFMA instead of one per MFMA
//   flush  stages of 32 k between f64 flushes (0 = never)
//
//   hipcc --offload-arch=gfx950 -O3 tools/sb32_numerics.hip -o tools/sb32_numerics && tools/sb32_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

__device__ __forceinline__ unsigned short bf16_rn(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void split3(const float *src, bf16x8 (&p)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = src[j];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned short h = bf16_rn(x);
            p[q][j] = (short)h;
            x -= bf16_f(h);                                    // exact
        }
    }
}

// TS = 16: A [16][K], W [16][K] per tile, v_mfma_f32_16x16x32_bf16; TS = 32: [32][K] each, v_mfma_f32_32x32x16_bf16
template <int TS>
__global__ __launch_bounds__(64) void k_tile(const float *A, const float *W, int K, int order, int accmode, int flush, double *out) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const float *a = A + (size_t)tile * TS * K, *w = W + (size_t)tile * TS * K;
    constexpr int NV = TS == 16 ? 4 : 16;
    typedef typename std::conditional<TS == 16, f32x4, f32x16>::type acc_t;
    acc_t acc, acc2;                     // acc2: the odd-stage chain (accmode 1) or the never-flushed low-order chain (accmode 2)
    double run[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { acc[i] = 0.f; acc2[i] = 0.f; run[i] = 0.0; }
    int since = 0, stage = 0;
    auto mfma = [&](const bf16x8 &wf, const bf16x8 &af, acc_t &c) {
        if constexpr (TS == 16) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, c, 0, 0, 0);
    };
    for (int k0 = 0; k0 < K; k0 += 32, ++stage) {
        acc_t &main = (accmode == 1 && (stage & 1)) ? acc2 : acc;
        acc_t &low = accmode == 2 ? acc2 : main;
        if constexpr (TS == 16) {
            const int r = lane & 15, q = lane >> 4;
            bf16x8 af[3], wf[3];
            split3(a + (size_t)r * K + k0 + q * 8, af);
            split3(w + (size_t)r * K + k0 + q * 8, wf);
            mfma(wf[2], af[0], low);
            mfma(wf[1], af[1], low);
            mfma(wf[1], af[0], low);
            mfma(wf[0], af[2], low);
            mfma(wf[0], af[1], low);
            mfma(wf[0], af[0], main);
        } else {
            const int r = lane & 31, g = lane >> 5;
            bf16x8 af[2][3], wf[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                split3(a + (size_t)r * K + k0 + 16 * h + 8 * g, af[h]);
                split3(w + (size_t)r * K + k0 + 16 * h + 8 * g, wf[h]);
            }
            if (order == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    mfma(wf[h][2], af[h][0], low);
                    mfma(wf[h][1], af[h][1], low);
                    mfma(wf[h][1], af[h][0], low);
                    mfma(wf[h][0], af[h][2], low);
                    mfma(wf[h][0], af[h][1], low);
                    mfma(wf[h][0], af[h][0], main);
                }
            } else {
                const int pw[6] = {2, 1, 1, 0, 0, 0}, pa[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int h = 0; h < 2; ++h) mfma(wf[h][pw[s]], af[h][pa[s]], s == 5 ? main : low);
            }
        }
        if (flush > 0 && ++since == flush) {
#pragma unroll
            for (int i = 0; i < NV; ++i) { run[i] += (double)acc[i]; acc[i] = 0.f; }
            since = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double v;
        if (accmode == 1) v = (double)(acc[i] + acc2[i]);                  // fp32 add of the two chains, as the kernel's epilogue
        else if (accmode == 2) v = (run[i] + (double)acc[i]) + (double)acc2[i];
        else v = run[i] + (double)acc[i];
        int f, r;
        if (TS == 16) { f = (lane >> 4) * 4 + i; r = lane & 15; }
        else { f = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3); r = lane & 31; }
        out[((size_t)tile * TS + f) * TS + r] = v;                         // [feature][row]
    }
}

static double urand(unsigned long long &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
}

template <int TS>
static void run_ts(int K) {
    const int T = TS == 16 ? 64 : 16;
    std::vector<float> A((size_t)T * TS * K), W((size_t)T * TS * K);
    unsigned long long seed = 12345 + K;
    // activations after LeakyReLU(0.1): mostly positive, magnitude ~1; weights uniform in +-3/sqrt(K) (outputs of unit scale)
    for (auto &v : A) { double u = urand(seed) * 2 - 0.6; v = (float)(u > 0 ? u : 0.1 * u); }
    const double wb = 3.0 / std::sqrt((double)K);
    for (auto &v : W) v = (float)((urand(seed) * 2 - 1) * wb);
    std::vector<double> exact((size_t)T * TS * TS);
    double scale = 0;
    for (int t = 0; t < T; ++t)
        for (int f = 0; f < TS; ++f)
            for (int r = 0; r < TS; ++r) {
                long double s = 0;
                for (int k = 0; k < K; ++k) s += (long double)W[((size_t)t * TS + f) * K + k] * (long double)A[((size_t)t * TS + r) * K + k];
                exact[((size_t)t * TS + f) * TS + r] = (double)s;
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
    printf("K = %d, %s, largest |y| %.3f, ulp %.3e (max / rms over %zu outputs)\n", K, TS == 16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16",
           scale, ulp, exact.size());
    struct V { const char *name; int order, acc, flush; };
    std::vector<V> vs;
    if (TS == 16) {
        vs = {{"x6, one chain, flush per stage", 0, 0, 1}, {"x6, one chain, flush every 2 stages (production MLP)", 0, 0, 2},
              {"x6, one chain, flush every 4 stages", 0, 0, 4}, {"x6, one chain, no flush", 0, 0, 0},
              {"x6, even / odd chains, no flush (production GAT)", 0, 1, 0},
              {"x6, split accumulators, leading chain flushed per stage", 0, 2, 1}, {"x6, split accumulators, flush every 2 stages", 0, 2, 2},
              {"x6, split accumulators, flush every 4 stages", 0, 2, 4}, {"x6, split accumulators, flush every 8 stages", 0, 2, 8},
              {"x6, split accumulators, no flush", 0, 2, 0}};
    } else {
        vs = {{"x6 half-major, one chain, flush per stage", 0, 0, 1}, {"x6 half-major, one chain, flush every 2 stages", 0, 0, 2},
              {"x6 half-major, one chain, flush every 4 stages", 0, 0, 4}, {"x6 half-major, one chain, no flush", 0, 0, 0},
              {"x6 half-major, even / odd chains, no flush", 0, 1, 0},
              {"x6 plane-major, one chain, flush per stage", 1, 0, 1}, {"x6 plane-major, one chain, flush every 2 stages", 1, 0, 2},
              {"x6 plane-major, one chain, flush every 4 stages", 1, 0, 4}, {"x6 plane-major, one chain, no flush", 1, 0, 0},
              {"x6 plane-major, even / odd chains, no flush", 1, 1, 0},
              {"x6 half-major, split accumulators, flush per stage", 0, 2, 1}, {"x6 half-major, split accumulators, flush every 2 stages", 0, 2, 2},
              {"x6 half-major, split accumulators, flush every 4 stages", 0, 2, 4}, {"x6 half-major, split accumulators, no flush", 0, 2, 0}};
    }
    std::vector<double> got(exact.size());
    for (auto &v : vs) {
        hipLaunchKernelGGL((k_tile<TS>), dim3(T), dim3(64), 0, 0, dA, dW, K, v.order, v.acc, v.flush, dO);
        hipMemcpy(got.data(), dO, got.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0, sq = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            const double e = ((double)(float)got[i] - exact[i]) / ulp;
            mx = std::fmax(mx, std::fabs(e));
            sq += e * e;
        }
        printf("  %-62s max %8.2f  rms %7.3f\n", v.name, mx, std::sqrt(sq / got.size()));
    }
    hipFree(dA);
    hipFree(dW);
    hipFree(dO);
}

int main() {
    const int Ks[3] = {416, 1280, 3072};
    for (int K : Ks) {
        run_ts<16>(K);
        run_ts<32>(K);
    }
    return 0;
}

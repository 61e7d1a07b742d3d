// VERDICT r5 item 4, the clock question: does the i8 matrix pipe keep its 2x rate over the bf16 pipe when every CU runs it on
// random operand bits (the batch GEMMs are power-bound: DESIGN.md 7.1)?  Pure matrix loops in the production kernel's geometry
// (eight MFMA waves per CU = two per SIMD, no memory traffic), per 16 x 16 x 64 block of a tile:
//   bf16  the split-bf16 form of today: 2 stages x 6 products of v_mfma_f32_16x16x32_bf16 into one fp32 accumulator
//   i8    the sliced form: 10 products (i + j < 4) of v_mfma_i32_16x16x64_i8 into four int32 level accumulators
// Reported: time per block and wave, the shader clock the loop held (s_memtime / s_memrealtime stamps), blocks per second chip-wide.
// hipcc --offload-arch=gfx950 -O3 tools/i8_clock.hip -o tools/i8_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NT = 4;        // independent 16 x 16 tiles per wave (the production wave owns 2 x 4)

__global__ __launch_bounds__(512) void k_bf16(const unsigned *__restrict__ rnd, float *__restrict__ out, unsigned long long *__restrict__ stamps, int iters) {
    bf16x8 a[3], w[3];
    unsigned *pa = reinterpret_cast<unsigned *>(a), *pw = reinterpret_cast<unsigned *>(w);
    for (int i = 0; i < 12; ++i) {
        pa[i] = rnd[(threadIdx.x * 24 + i) & 65535] & 0xBF7FBF7Fu;        // (finite bf16 values)
        pw[i] = rnd[(threadIdx.x * 24 + 12 + i + blockIdx.x) & 65535] & 0xBF7FBF7Fu;
    }
    f32x4 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if ((threadIdx.x & 63) == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], a[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[0], acc[t], 0, 0, 0);
            }
    }
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k_i8(const unsigned *__restrict__ rnd, float *__restrict__ out, unsigned long long *__restrict__ stamps, int iters) {
    i32x4 a[4], w[4];                   // four slices of 16 i8 values per lane (K = 64: 16 bytes per lane and operand)
    unsigned *pa = reinterpret_cast<unsigned *>(a), *pw = reinterpret_cast<unsigned *>(w);
    for (int i = 0; i < 16; ++i) {
        pa[i] = rnd[(threadIdx.x * 32 + i) & 65535];
        pw[i] = rnd[(threadIdx.x * 32 + 16 + i + blockIdx.x) & 65535];
    }
    i32x4 acc[NT][4];
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 4; ++l) acc[t][l] = (i32x4){0, 0, 0, 0};
    unsigned long long t0 = 0, r0 = 0;
    if ((threadIdx.x & 63) == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4 - i; ++j)
                    acc[t][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[j], a[i], acc[t][i + j], 0, 0, 0);
    }
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
    int s = 0;
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 4; ++l) s += acc[t][l][0] + acc[t][l][1] + acc[t][l][2] + acc[t][l][3];
    out[blockIdx.x * 512 + threadIdx.x] = (float)s;
}

int main(int argc, char **argv) {
    int cus = 256;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    std::vector<unsigned> h(65536);
    unsigned x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
    unsigned *rnd;
    float *out;
    unsigned long long *stamps;
    CHK(hipMalloc(&rnd, h.size() * 4));
    CHK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&out, (size_t)cus * 512 * 4));
    CHK(hipMalloc(&stamps, (size_t)cus * 8 * 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::vector<unsigned long long> st((size_t)cus * 16);
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) CHK(hipMemset(rnd, 0, h.size() * 4));
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0, 0));
                if (kind == 0) hipLaunchKernelGGL(k_bf16, dim3(cus), dim3(512), 0, 0, rnd, out, stamps, iters);
                else hipLaunchKernelGGL(k_i8, dim3(cus), dim3(512), 0, 0, rnd, out, stamps, iters);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep < 2) continue;
                CHK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
                double clk = 0;
                for (int i = 0; i < cus * 8; ++i) clk += (double)st[2 * i] / (double)st[2 * i + 1] * 0.1;      // shader cycles per 10 ns -> GHz
                clk /= cus * 8;
                const double blocks = (double)iters * NT * cus * 8;             // 16 x 16 x 64 blocks
                const int per_block = kind == 0 ? 12 : 10;
                printf("%-5s %-14s %8.3f ms  %7.2f ns per 16x16x64 block and wave  clock %.3f GHz  %5.1f matrix cycles per instruction  %.1f T blocks/s\n",
                       kind == 0 ? "bf16" : "i8", zero ? "zero operands" : "random bits", ms, ms * 1e6 / ((double)iters * NT), clk,
                       ms * 1e6 / ((double)iters * NT) * clk / per_block * 2 /* two waves per SIMD share the pipe */ / 2, blocks / (ms * 1e-3) * 1e-12);
            }
        }
    }
    return 0;
}

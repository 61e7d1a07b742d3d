// Diagnostic (round 3): how do staging instructions interfere with the fp32 MFMA stream?
//
// Every wave runs the GEMM's MFMA stream from registers (80 independent-enough
// v_mfma_f32_16x16x4_f32 per "stage", 10 accumulators) and, once per stage, issues P staging
// instructions of one kind:
//   kind 0  nothing (ceiling of this loop)
//   kind 1  global_load_lds_dwordx4  (LDS-DMA, 1 KiB per wave-instruction, 8 rows x 128 B)
//   kind 2  global_load_dwordx4 into VGPRs (same addresses)
//   kind 3  ds_read_b128 (fragment-style LDS reads)
//   kind 4  the LDS-DMA pieces issued by a FIFTH wave of the workgroup (loader wave) instead of
//           the four MFMA waves: 4P pieces per stage from that one wave
// Nothing waits for the data beyond a bound on requests in flight, there is no barrier and no
// dependence between the loads and the MFMAs: whatever the MFMA rate loses is issue / datapath
// interference, not latency.  Prints TFLOP/s (wall), the MFMA-pipe busy share in shader CYCLES
// (median wave: s_memtime around the loop) and the in-kernel clock (cycles / s_memrealtime), which
// separates issue interference from DVFS.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_interf.hip -o tools/dma_interf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int KIND, int P>
__global__ __launch_bounds__(KIND == 4 ? 320 : 256, 2) void k_interf(float *out, int iters, const float *src, unsigned span_mask,
                                                                     const float *rnd, long long *clk) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];       // 32 KiB ring for the DMA pieces / reads
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[r] = rnd[(tid * 16 + r) & 4095];
        b[r] = rnd[(tid * 16 + 8 + r + blockIdx.x) & 4095];
    }
    f32x4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = rnd[i & 4095];
    __syncthreads();
    // staging source: this wave's window walks through `span_floats` (8 rows x 32 floats per piece, row stride 416)
    const unsigned lane_off = (lane >> 3) * 416 + (lane & 7) * 4;
    const unsigned base = (blockIdx.x * 5u + wave) * 8u * 416u * 8u;
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    f32x4 ring[KIND == 2 ? P : 1];
#pragma unroll
    for (int p = 0; p < (KIND == 2 ? P : 1); ++p) ring[p] = sink;
    if (KIND == 4 && wave == 4) {
        // loader wave: 4P pieces per stage, bounded in flight
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 4 * P; ++p) {
                const unsigned off = (base + ((unsigned)it * 4 * P + p) * 8u * 416u) & span_mask;
                __builtin_amdgcn_global_load_lds((glb_void_t *)(src + off + lane_off),
                                                 (lds_void_t *)(lds + ((p & 31) * 256)), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const long long t_c0 = (long long)__builtin_readcyclecounter(), t_w0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const unsigned off = (base + ((unsigned)it * P + p) * 8u * 416u) & span_mask;
                __builtin_amdgcn_global_load_lds((glb_void_t *)(src + off + lane_off),
                                                 (lds_void_t *)(lds + (wave * 8 + (p & 7)) * 256), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if (KIND == 2) {
            // plain loads into a ring of P destination registers, consumed one stage later (the
            // compiler places the wait; a hand-issued asm load here would leave its destination
            // unprotected -- the first version of this tool faulted on exactly that)
#pragma unroll
            for (int p = 0; p < P; ++p) asm volatile("" ::"v"(ring[p]));
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const unsigned off = (base + ((unsigned)it * P + p) * 8u * 416u) & span_mask;
                ring[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + off + lane_off));
            }
        } else if (KIND == 3) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(&lds[((wave * 8 + p) & 31) * 256 + lane * 4]);
                asm volatile("" ::"v"(v));
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 10; ++i) sink += acc[i];
    asm volatile("" ::"v"(sink));
    const long long t_c1 = (long long)__builtin_readcyclecounter(), t_w1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (clk && lane == 0) {
        clk[(blockIdx.x * 4 + wave) * 2 + 0] = t_c1 - t_c0;      // shader cycles of the loop
        clk[(blockIdx.x * 4 + wave) * 2 + 1] = t_w1 - t_w0;      // 100 MHz ticks
        clk[2048 * 8 + (blockIdx.x * 4 + wave) * 2 + 0] = t_w0;   // absolute start / end (100 MHz)
        clk[2048 * 8 + (blockIdx.x * 4 + wave) * 2 + 1] = t_w1;
    }
    out[blockIdx.x * 256 + (tid & 255)] = sink[0] + sink[1] + sink[2] + sink[3];
}

static float *g_out, *g_src, *g_rnd;
static unsigned g_span;
static long long *g_clk;
static double g_busy, g_ghz;

template <int KIND, int P>
double run(int wg_per_cu, int iters) {
    const int grid = 256 * wg_per_cu;
    const int threads = KIND == 4 ? 320 : 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_interf<KIND, P>), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_interf<KIND, P>), dim3(grid), dim3(threads), 40 * 1024, 0, g_out, 50, g_src, g_span - 1, g_rnd, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_interf<KIND, P>), dim3(grid), dim3(threads), 40 * 1024, 0, g_out, iters, g_src, g_span - 1, g_rnd, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)grid * 8);
    hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int i = 0; i < grid * 4; ++i) {
        cyc.push_back((double)h[2 * i]);
        ghz.push_back(h[2 * i + 1] > 0 ? (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0) / 1e0 : 0.0);   // cycles per ns
    }
    std::vector<long long> ab((size_t)grid * 8);
    hipMemcpy(ab.data(), g_clk + 2048 * 8, ab.size() * 8, hipMemcpyDeviceToHost);
    long long t_first = ab[0], t_last = ab[1];
    std::vector<double> starts;
    for (int i = 0; i < grid * 4; ++i) {
        t_first = std::min(t_first, ab[2 * i]);
        t_last = std::max(t_last, ab[2 * i + 1]);
    }
    for (int i = 0; i < grid * 4; ++i) starts.push_back((double)(ab[2 * i] - t_first) / 100.0);   // us
    std::sort(starts.begin(), starts.end());
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    if (getenv("DMA_INTERF_DIST"))
        printf("    [kind %d P %d wg %d] wave loop cycles min %.0f med %.0f p90 %.0f max %.0f | loop start us med %.1f p90 %.1f max %.1f | span %.1f us, event %.1f us\n",
               KIND, P, wg_per_cu, cyc[0], cyc[cyc.size() / 2], cyc[cyc.size() * 9 / 10], cyc.back(), starts[starts.size() / 2],
               starts[starts.size() * 9 / 10], starts.back(), (double)(t_last - t_first) / 100.0, ms * 1e3);
    // MFMA-pipe busy share in CYCLES: the waves of a SIMD need wg_per_cu * iters * 80 * 32 cycles of it
    g_busy = (double)wg_per_cu * iters * 80 * 32 / cyc[cyc.size() / 2];
    g_ghz = ghz[ghz.size() / 2];
    const double flop = (double)grid * 4 * iters * 80 * 2048.0;
    return flop / ms / 1e9;
}

template <int KIND>
void sweep(const char *name, int wg, int iters, double base) {
    double t[5], bz[5], gz[5];
    t[0] = run<KIND, 1>(wg, iters); bz[0] = g_busy; gz[0] = g_ghz;
    t[1] = run<KIND, 2>(wg, iters); bz[1] = g_busy; gz[1] = g_ghz;
    t[2] = run<KIND, 4>(wg, iters); bz[2] = g_busy; gz[2] = g_ghz;
    t[3] = run<KIND, 7>(wg, iters); bz[3] = g_busy; gz[3] = g_ghz;
    t[4] = run<KIND, 12>(wg, iters); bz[4] = g_busy; gz[4] = g_ghz;
    const int ps[5] = {1, 2, 4, 7, 12};
    printf("%-28s wg/CU %d:", name, wg);
    (void)base;
    for (int i = 0; i < 5; ++i) printf("  P=%2d %6.1f TF busy %.3f @%.2f GHz |", ps[i], t[i], bz[i], gz[i]);
    printf("\n");
}

int main(int argc, char **argv) {
    const long span_mb = argc > 1 ? atol(argv[1]) : 512;
    g_span = (unsigned)(span_mb * 1024 * 1024 / 4);      // power of two
    hipMalloc(&g_out, (size_t)256 * 8 * 256 * sizeof(float));
    hipMalloc(&g_src, (size_t)g_span * 4 + (1 << 20));
    hipMalloc(&g_rnd, 4096 * sizeof(float));
    hipMalloc(&g_clk, (size_t)2 * 2048 * 8 * sizeof(long long));
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(g_rnd, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipMemset(g_src, 0, (size_t)g_span * 4 + (1 << 20));
    const int iters = 3000;
    printf("staging source span %ld MB\n", span_mb);
    for (int wg = 1; wg <= 3; ++wg) {
        const double base = run<0, 1>(wg, iters);
        printf("no staging                   wg/CU %d: %6.1f TF busy %.3f @%.2f GHz\n", wg, base, g_busy, g_ghz);
        sweep<1>("LDS-DMA dwordx4", wg, iters, base);
        sweep<2>("global_load_dwordx4 -> VGPR", wg, iters, base);
        sweep<3>("ds_read_b128", wg, iters, base);
        if (wg <= 2) sweep<4>("LDS-DMA from a loader wave", wg, iters, base);
    }
    return 0;
}

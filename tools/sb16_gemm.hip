// Timing model (wrong data, right data flow) of a SPLIT-bf16 GEMM in the production kernel's shape: fp32 operands as three bf16
// planes each (a = a1 + a2 + a3 exactly), products a_i b_j on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, the six
// significant ones (i + j <= 4: a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) or fewer.  VERDICT r3 item 3 asks what such a kernel would
// reach before anything is built into the library: the bf16 pipe is 16x the fp32 one, six products break even at 0.67 PFLOP/s.
//
// Same skeleton as k_linear_dma: tile 128 x 80 x 32, 4 MFMA waves (32 rows x 80 features each) + 4 loader waves that issue the
// LDS-DMA (global_load_lds_dwordx4) of the next K stage, double-buffered LDS, one barrier per stage.  A stage holds the three
// planes of both operands: (128 + 80) rows x 64 B x 3 = 39 KB (fp32 kernel: 26 KB), so two workgroups per CU.  Rows are 64 B
// (32 bf16); a row's four 16-byte chunks sit at position c ^ ((row >> 2) & 3): conflict-free ds_read_b128 fragment reads.
//
//   hipcc --offload-arch=gfx950 -O3 tools/sb16_gemm.hip -o tools/sb16_gemm && tools/sb16_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#ifndef SB_SWZ_OLD
#define SB_SWZ_OLD 0
#endif
#ifndef SB_ORDER
#define SB_ORDER 0
#endif
#ifndef SB_RANDOM
#define SB_RANDOM 0
#endif

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int BM = 128, BN = 80, BK = 32;
constexpr int ROW_B = BK * 2;                         // bytes per tile row
constexpr int PLANE_A = BM * ROW_B, PLANE_W = BN * ROW_B;
constexpr int STAGE_B = 3 * PLANE_A + 3 * PLANE_W;    // 39936 bytes

#if SB_SWZ_OLD
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }      // the first version: 2-way conflicts in the real ds_read_b128 lane groups
#else
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }      // gemm_sb16.hip: w_swz
#endif

// NPROD: 6 = fp32-accurate split, 3 = a1b1 + a1b2 + a2b1 (~16 significant bits), 1 = plain bf16 (planes 0 only)
// STAGING: 1 = LDS-DMA from loader waves, 0 = no staging at all (the MFMA + fragment-read ceiling)
template <int NPROD, int STAGING>
__global__ __launch_bounds__(512, 2) void k_sb16(float *out, int nk, const unsigned short *A, const unsigned short *W, size_t plane_a,
                                                 size_t plane_w, int ld, int rows_a, int share, int n_planes) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = ((blockIdx.x / share) * BM) % rows_a, n0 = (blockIdx.x % share) * BN;
    constexpr int NP = NPROD == 1 ? 1 : NPROD == 3 ? 2 : 3;       // planes that are staged
    if (wave >= 4) {
        // ---- loader waves: one wave-instruction = 1 KiB = 16 rows x 64 B of one plane ----
        const int li = wave - 4;
        const int r = lane >> 2, c = lane & 3;
        __builtin_amdgcn_s_setprio(3);
        constexpr int GA = BM / 16, GW = BN / 16;                  // 8 and 5 groups per plane
        constexpr int NI = NP * (GA + GW);                         // DMA instructions per stage
        for (int kt = 0; kt <= nk; ++kt) {
            if (kt > 0) __syncthreads();                           // stage kt-1 landed (vmcnt 0) and the other buffer is free
            if (kt == nk) break;
            if (STAGING) {
                unsigned char *base = lds + (kt & 1) * STAGE_B;
                const size_t koff = (size_t)kt * BK;
#pragma unroll
                for (int i = 0; i < (NI + 3) / 4; ++i) {
                    const int q = li + 4 * i;
                    if (q < NI) {
                        const int p = q / (GA + GW), g = q - p * (GA + GW);
                        if (g < GA) {
                            const int row = g * 16 + r;
                            const unsigned short *src = A + p * plane_a + (size_t)(m0 + row) * ld + koff + ((c ^ swz(row)) << 3);
                            __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(base + p * PLANE_A + g * 16 * ROW_B), 16, 0, 0);
                        } else {
                            const int row = (g - GA) * 16 + r;
                            const unsigned short *src = W + p * plane_w + (size_t)(n0 + row) * ld + koff + ((c ^ swz(row)) << 3);
                            __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(base + 3 * PLANE_A + p * PLANE_W + (g - GA) * 16 * ROW_B), 16, 0, 0);
                        }
                    }
                }
            }
        }
        return;
    }
    // ---- MFMA waves ----
    const int fr = lane & 15, fq = lane >> 4;
    int a_rd[2], w_rd[5];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wave * 32 + mt * 16 + fr;
        a_rd[mt] = row * ROW_B + ((fq ^ swz(row)) << 4);
    }
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int row = nt * 16 + fr;
        w_rd[nt] = 3 * PLANE_A + row * ROW_B + ((fq ^ swz(row)) << 4);
    }
    f32x4 acc[5][2];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const unsigned char *cur = lds + (kt & 1) * STAGE_B;
        bf16x8 af[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[p][mt] = *reinterpret_cast<const bf16x8 *>(cur + p * PLANE_A + a_rd[mt]);
#if SB_ORDER == 1
        // production order (gemm_sb16.hip SB_STAGE): the six products of one accumulator back to back, weight fragments per column tile
        if (NPROD == 6) {
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                bf16x8 wp[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) wp[p] = *reinterpret_cast<const bf16x8 *>(cur + p * PLANE_W + w_rd[nt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[2], af[0][mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[1], af[1][mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[1], af[0][mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[0], af[2][mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[0], af[1][mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[0], af[0][mt], acc[nt][mt], 0, 0, 0);
                }
            }
            continue;
        }
#endif
#pragma unroll
        for (int pw = 0; pw < NP; ++pw) {
            bf16x8 wf[5];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) wf[nt] = *reinterpret_cast<const bf16x8 *>(cur + pw * PLANE_W + w_rd[nt]);
            // products a_pa * w_pw with pa + pw <= NP - 1 (planes are ordered by significance): 3 + 2 + 1 = 6, 2 + 1 = 3, or 1
#pragma unroll
            for (int pa = 0; pa + pw < NP; ++pa)
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[pa][mt], acc[nt][mt], 0, 0, 0);
        }
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 5; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) sum += acc[nt][mt];
    out[(size_t)blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <int NPROD, int STAGING>
static void run(const char *what, int M, int N, int K) {
    const int nk = K / BK, share = N / BN, tiles_m = M / BM;
    const int grid = tiles_m * share;
    const size_t plane_a = (size_t)(M + BM) * K, plane_w = (size_t)(N + BN) * K;
    unsigned short *A, *W;
    float *out;
    hipMalloc(&A, 3 * plane_a * 2);
    hipMalloc(&W, 3 * plane_w * 2);
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMemset(A, 0x3c, 3 * plane_a * 2);               // 0x3c3c = 0.0115 as bf16: finite, non-trivial operand bits
    hipMemset(W, 0x3c, 3 * plane_w * 2);
#if SB_RANDOM
    {   // operands with random mantissas and signs (the matrix pipe's power, hence the clock, depends on the operand bits)
        std::vector<unsigned short> h(1 << 22);
        unsigned x = 12345u;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x80FFu) | 0x3C00u | ((x >> 9) & 0x0300u)); }
        for (size_t o = 0; o < 3 * plane_a * 2; o += h.size() * 2) hipMemcpy((char *)A + o, h.data(), std::min(h.size() * 2, 3 * plane_a * 2 - o), hipMemcpyHostToDevice);
        for (size_t o = 0; o < 3 * plane_w * 2; o += h.size() * 2) hipMemcpy((char *)W + o, h.data(), std::min(h.size() * 2, 3 * plane_w * 2 - o), hipMemcpyHostToDevice);
    }
#endif
    const void *fn = reinterpret_cast<const void *>(k_sb16<NPROD, STAGING>);
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        hipLaunchKernelGGL((k_sb16<NPROD, STAGING>), dim3(grid), dim3(512), 2 * STAGE_B, 0, out, nk, A, W, plane_a, plane_w, K, M, share, 3);
    };
    launch();
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double eq = 2.0 * (double)tiles_m * BM * (double)share * BN * K;          // fp32-equivalent FLOP of the GEMM
    printf("%-46s M=%6d N=%4d K=%4d  %d products%s: %.3f ms  %.1f fp32-equivalent TFLOP/s (bf16 MFMA rate %.2f PFLOP/s)%s\n", what, M, N, K, NPROD,
           STAGING ? "" : ", NO staging", ms, eq / ms / 1e9, eq * NPROD / ms / 1e12, hipGetLastError() == hipSuccess ? "" : "  [launch error]");
    hipFree(A);
    hipFree(W);
    hipFree(out);
}

int main() {
    // the two shapes that carry the step: GAT fc1 / fc2 (180 k rows, K = 416 padded, N = 400) and the MLP's big layers (4 k rows)
    const int shapes[3][3] = {{180224, 400, 416}, {180224, 400, 3328}, {4096, 3040, 3072}};
    printf("SB_ORDER %d SB_RANDOM %d\n", SB_ORDER, SB_RANDOM);
    for (auto &s : shapes) {
        run<6, 1>("split-bf16, 6 products, LDS-DMA staging", s[0], s[1], s[2]);
        run<6, 0>("split-bf16, 6 products, no staging (ceiling)", s[0], s[1], s[2]);
        run<3, 1>("3 products (a1b1+a1b2+a2b1), staging", s[0], s[1], s[2]);
        run<1, 1>("plain bf16 (1 product), staging", s[0], s[1], s[2]);
        run<1, 0>("plain bf16, no staging", s[0], s[1], s[2]);
    }
    return 0;
}

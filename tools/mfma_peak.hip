// Diagnostic: what rate does a pure v_mfma_f32_16x16x4_f32 loop reach on this board (no memory
// traffic at all)?  Gives the practical ceiling the GEMM's 157.3 TFLOP/s "peak" should be read
// against.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// same loop with eight different random operand pairs per lane (data toggling as in a real GEMM)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_rand(float *out, int iters, const float *rnd) {
    f32x4 acc[NACC];
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[r] = rnd[(threadIdx.x * 16 + r) & 4095];
        b[r] = rnd[(threadIdx.x * 16 + 8 + r + blockIdx.x) & 4095];
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float a, float b) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
void run(int blocks_per_cu, int cus) {
    const int grid = cus * blocks_per_cu, iters = 4000;
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 8 * NACC * 2048.0;
    printf("accumulators/wave %2d, workgroups/CU %d: %.1f TFLOP/s (%.2f ms)\n", NACC, blocks_per_cu, flop / ms / 1e9, ms);
    hipFree(out);
}

// GEMM-shaped inner loop without global traffic: per "stage" 2 x (2 + 5) ds_read_b128 fragment
// loads from LDS, then 2 x 40 MFMAs on them (the k_linear_dma loop minus DMA and barrier).
// PIPE = 0: loads, wait, MFMAs (what the compiler makes of the GEMM source);
// PIPE = 1: the loads of the next half-stage are issued before the MFMAs of the current one.
template <int PIPE>
__global__ __launch_bounds__(256, 3) void k_lds_mfma(float *out, int iters, const float *rnd) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 6656];
    for (int i = threadIdx.x; i < 2 * 6656; i += 256) lds[i] = rnd[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15;
    int a_rd[2], w_rd[5];
    for (int mt = 0; mt < 2; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * 32;
    for (int nt = 0; nt < 5; ++nt) w_rd[nt] = 4096 + (nt * 16 + fr) * 32;
    const int sw = ((fr >> 1) & 1) | (((fr >> 2) & 1) << 2);
    const int c0 = ((fq * 2 + 0) ^ sw) << 2, c1 = ((fq * 2 + 1) ^ sw) << 2;
    f32x4 acc[5][2];
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 af[2][2], wf[2][5];
    auto load = [&](int slot, int base, int co) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[slot][mt] = *reinterpret_cast<const f32x4 *>(&lds[base + a_rd[mt] + co]);
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) wf[slot][nt] = *reinterpret_cast<const f32x4 *>(&lds[base + w_rd[nt] + co]);
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][nt][s], af[slot][mt][s], acc[nt][mt], 0, 0, 0);
    };
    if (PIPE == 0) {
        for (int it = 0; it < iters; ++it) {
            const int base = (it & 1) * 6656;
            load(0, base, c0);
            mfmas(0);
            load(0, base, c1);
            mfmas(0);
        }
    } else {
        load(0, 0, c0);
        for (int it = 0; it < iters; ++it) {
            const int base = (it & 1) * 6656, nbase = ((it + 1) & 1) * 6656;
            load(1, base, c1);
            mfmas(0);
            load(0, nbase, c0);
            mfmas(1);
        }
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) sum += acc[nt][mt];
    out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

// the same loop with the real staging: per stage one barrier and 7 (6) global_load_lds_dwordx4
// per wave that bring the next 32-deep stage of a 128-row activation tile and an 80-row weight
// tile from global memory (MODE 1), or the barrier alone (MODE 0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
template <int MODE>
__global__ __launch_bounds__(256, MODE == 2 ? 2 : 3) void k_stage_mfma(float *out, int nk, const float *A, const float *W, int ld, int rows_a, int share, long long *clk = nullptr) {
    const long long t_c0 = (long long)__builtin_readcyclecounter(), t_w0 = (long long)__builtin_amdgcn_s_memrealtime();
    constexpr int NBUF = MODE == 2 ? 3 : 2;
    __shared__ __attribute__((aligned(1024))) float lds[NBUF * 6656];
    for (int i = threadIdx.x; i < NBUF * 6656; i += 256) lds[i] = A[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15, dr = lane >> 3, dp = lane & 7;
    auto swz = [](int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); };
    int a_rd[2], w_rd[5];
    for (int mt = 0; mt < 2; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * 32;
    for (int nt = 0; nt < 5; ++nt) w_rd[nt] = 4096 + (nt * 16 + fr) * 32;
    const int c0 = ((fq * 2 + 0) ^ swz(fr)) << 2, c1 = ((fq * 2 + 1) ^ swz(fr)) << 2;
    const float *a_src[4], *w_src[3];
    // `share` consecutive workgroups read the same activation tile (as the feature tiles of one
    // row block do in the GEMM), each a different weight tile
    const int m0 = ((blockIdx.x / share) * 128) % rows_a;
    for (int g = 0; g < 4; ++g) {
        const int row = wave * 32 + g * 8 + dr;
        a_src[g] = A + (size_t)(m0 + row) * ld + ((dp ^ swz(row)) << 2);
    }
    for (int g = 0; g < 3; ++g) {
        int grp = wave + 4 * g;
        if (grp > 9) grp = 9;
        const int row = grp * 8 + dr;
        w_src[g] = W + (size_t)((blockIdx.x % share) * 80 + row) * ld + ((dp ^ swz(row)) << 2);
    }
    const bool w_third = wave < 2;
    auto issue = [&](int kt, int buf) {
        const int koff = kt * 32;
        float *base = lds + buf * 6656;
        if (MODE != 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(a_src[g] + koff), (lds_void_t *)(base + (wave * 32 + g * 8) * 32), 16, 0, 0);
        }
        if (MODE == 3) return;
#pragma unroll
        for (int g = 0; g < 2; ++g)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[g] + koff), (lds_void_t *)(base + 4096 + (wave + 4 * g) * 8 * 32), 16, 0, 0);
        if (w_third)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[2] + koff), (lds_void_t *)(base + 4096 + (wave + 8) * 8 * 32), 16, 0, 0);
    };
    f32x4 acc[5][2];
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (MODE >= 1) issue(0, 0);
    if (MODE == 2 && nk > 1) issue(1, 1);
    int cbuf = 0, nbuf = 2;
    for (int kt = 0; kt < nk; ++kt) {
        int cur;
        if (MODE == 2) {
            if (kt + 1 < nk) {
                if (w_third) __builtin_amdgcn_s_waitcnt(0xF77);
                else __builtin_amdgcn_s_waitcnt(0xF76);
            } else {
                __builtin_amdgcn_s_waitcnt(0xF70);
            }
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) issue(kt + 2, nbuf);
            cur = cbuf * 6656;
            cbuf = cbuf == 2 ? 0 : cbuf + 1;
            nbuf = nbuf == 2 ? 0 : nbuf + 1;
        } else {
            __syncthreads();
            if ((MODE == 1 || MODE >= 3) && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
            cur = (kt & 1) * 6656;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int co = hh ? c1 : c0;
            f32x4 af[2], wf[5];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const f32x4 *>(&lds[cur + a_rd[mt] + co]);
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) wf[nt] = *reinterpret_cast<const f32x4 *>(&lds[cur + w_rd[nt] + co]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s], af[mt][s], acc[nt][mt], 0, 0, 0);
        }
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) sum += acc[nt][mt];
    out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = (long long)__builtin_readcyclecounter() - t_c0;
        clk[1] = (long long)__builtin_amdgcn_s_memrealtime() - t_w0;
    }
}

// cross-stage software pipeline: three LDS buffers; the barrier at the top of stage kt certifies
// stage kt+1 (requested a whole stage earlier), so the first-half fragments of stage kt+1 are
// read while the second half of stage kt is still in the MFMA pipe and no wave starts a stage
// with empty registers
__global__ __launch_bounds__(256, 2) void k_stage_pipe(float *out, int nk, const float *A, const float *W, int ld, int rows_a, int share) {
    __shared__ __attribute__((aligned(1024))) float lds[3 * 6656];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15, dr = lane >> 3, dp = lane & 7;
    auto swz = [](int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); };
    int a_rd[2], w_rd[5];
    for (int mt = 0; mt < 2; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * 32;
    for (int nt = 0; nt < 5; ++nt) w_rd[nt] = 4096 + (nt * 16 + fr) * 32;
    const int c0 = ((fq * 2 + 0) ^ swz(fr)) << 2, c1 = ((fq * 2 + 1) ^ swz(fr)) << 2;
    const float *a_src[4], *w_src[3];
    const int m0 = ((blockIdx.x / share) * 128) % rows_a;
    for (int g = 0; g < 4; ++g) {
        const int row = wave * 32 + g * 8 + dr;
        a_src[g] = A + (size_t)(m0 + row) * ld + ((dp ^ swz(row)) << 2);
    }
    for (int g = 0; g < 3; ++g) {
        int grp = wave + 4 * g;
        if (grp > 9) grp = 9;
        const int row = grp * 8 + dr;
        w_src[g] = W + (size_t)((blockIdx.x % share) * 80 + row) * ld + ((dp ^ swz(row)) << 2);
    }
    const bool w_third = wave < 2;
    auto issue = [&](int kt, int buf) {
        const int koff = kt * 32;
        float *base = lds + buf * 6656;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(a_src[g] + koff), (lds_void_t *)(base + (wave * 32 + g * 8) * 32), 16, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[g] + koff), (lds_void_t *)(base + 4096 + (wave + 4 * g) * 8 * 32), 16, 0, 0);
        if (w_third)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[2] + koff), (lds_void_t *)(base + 4096 + (wave + 8) * 8 * 32), 16, 0, 0);
    };
    f32x4 acc[5][2];
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 af[2][2], wf[2][5];
    auto load = [&](int slot, int base, int co) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[slot][mt] = *reinterpret_cast<const f32x4 *>(&lds[base + a_rd[mt] + co]);
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) wf[slot][nt] = *reinterpret_cast<const f32x4 *>(&lds[base + w_rd[nt] + co]);
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][nt][s], af[slot][mt][s], acc[nt][mt], 0, 0, 0);
    };
    issue(0, 0);
    __syncthreads();
    if (nk > 1) issue(1, 1);
    load(0, 0, c0);
    int cb = 0;                                   // buffer of the current stage
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                          // stage kt+1 landed everywhere; buffer of stage kt-1 is free
        const int nb = cb == 2 ? 0 : cb + 1, pb = cb == 0 ? 2 : cb - 1;
        if (kt + 2 < nk) issue(kt + 2, pb);
        load(1, cb * 6656, c1);
        mfmas(0);
        if (kt + 1 < nk) load(0, nb * 6656, c0);
        mfmas(1);
        cb = nb;
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) sum += acc[nt][mt];
    out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

void run_pipe(int blocks_per_cu, int cus, int nk, int share, int rows_a) {
    const int grid = cus * blocks_per_cu, ld = nk * 32;
    float *out, *A, *W;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&A, (size_t)(rows_a + 128) * ld * sizeof(float));
    hipMalloc(&W, (size_t)(share * 80) * ld * sizeof(float));
    hipMemset(A, 0x3c, (size_t)(rows_a + 128) * ld * sizeof(float));
    hipMemset(W, 0x3c, (size_t)(share * 80) * ld * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_stage_pipe, dim3(grid), dim3(256), 0, 0, out, nk, A, W, ld, rows_a, share);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_stage_pipe, dim3(grid), dim3(256), 0, 0, out, nk, A, W, ld, rows_a, share);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * grid * 4 * (double)nk * 80 * 2048.0;
    printf("cross-stage pipeline (3 buffers, fragments read one half-stage ahead), K = %d, %d rows, workgroups/CU %d: %.1f TFLOP/s (%.2f ms)\n", nk * 32, rows_a, blocks_per_cu, flop / ms / 1e9, ms / 4);
    hipFree(out);
    hipFree(A);
    hipFree(W);
}


// wider feature tiles: NT MFMA tiles (16 features each) per wave, i.e. fewer staged bytes and
// DMA requests per MFMA (128 x 16*NT tile, double-buffered, plain schedule)
template <int NT>
__global__ __launch_bounds__(256, NT > 10 ? 1 : NT > 5 ? 2 : 3) void k_stage_wide(float *out, int nk, const float *A, const float *W, int ld, int rows_a, int share, float *Cout = nullptr, int ldc = 0) {
    constexpr int BN = NT * 16, STG = (128 + BN) * 32, WOFF = 128 * 32, WG = NT * 2;
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15, dr = lane >> 3, dp = lane & 7;
    auto swz = [](int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); };
    int a_rd[2], w_rd[NT];
    for (int mt = 0; mt < 2; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * 32;
    for (int nt = 0; nt < NT; ++nt) w_rd[nt] = WOFF + (nt * 16 + fr) * 32;
    const int c0 = ((fq * 2 + 0) ^ swz(fr)) << 2, c1 = ((fq * 2 + 1) ^ swz(fr)) << 2;
    constexpr int WPW = (WG + 3) / 4;                  // weight groups per wave (last may be partial)
    const float *a_src[4], *w_src[WPW];
    const int m0 = ((blockIdx.x / share) * 128) % rows_a;
    for (int g = 0; g < 4; ++g) {
        const int row = wave * 32 + g * 8 + dr;
        a_src[g] = A + (size_t)(m0 + row) * ld + ((dp ^ swz(row)) << 2);
    }
    for (int g = 0; g < WPW; ++g) {
        int grp = wave + 4 * g;
        if (grp > WG - 1) grp = WG - 1;
        const int row = grp * 8 + dr;
        w_src[g] = W + (size_t)((blockIdx.x % share) * BN + row) * ld + ((dp ^ swz(row)) << 2);
    }
    auto issue = [&](int kt, int buf) {
        const int koff = kt * 32;
        float *base = lds + buf * STG;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(a_src[g] + koff), (lds_void_t *)(base + (wave * 32 + g * 8) * 32), 16, 0, 0);
#pragma unroll
        for (int g = 0; g < WPW; ++g)
            if (wave + 4 * g < WG)
                __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[g] + koff), (lds_void_t *)(base + WOFF + (wave + 4 * g) * 8 * 32), 16, 0, 0);
    };
    f32x4 acc[NT][2];
    for (int nt = 0; nt < NT; ++nt)
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const int cur = (kt & 1) * STG;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int co = hh ? c1 : c0;
            f32x4 af[2], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const f32x4 *>(&lds[cur + a_rd[mt] + co]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const f32x4 *>(&lds[cur + w_rd[nt] + co]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s], af[mt][s], acc[nt][mt], 0, 0, 0);
        }
    }
    if (Cout) {                                    // the GEMM's epilogue: every lane stores 4 consecutive features
        const int n0 = (blockIdx.x % share) * BN;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 v = acc[nt][mt];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * 0.15f;
                *reinterpret_cast<f32x4 *>(Cout + (size_t)(m0 + wave * 32 + mt * 16 + fr) * ldc + n0 + nt * 16 + fq * 4) = v;
            }
        return;
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int nt = 0; nt < NT; ++nt)
        for (int mt = 0; mt < 2; ++mt) sum += acc[nt][mt];
    out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <int NT>
void run_wide(int blocks_per_cu, int cus, int nk, int share, int rows_a, bool write_c = false) {
    const int grid = cus * blocks_per_cu, ld = nk * 32, BN = NT * 16;
    const size_t shm = (size_t)2 * (128 + BN) * 32 * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_stage_wide<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    float *out, *A, *W;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&A, (size_t)(rows_a + 128) * ld * sizeof(float));
    hipMalloc(&W, (size_t)(share * BN) * ld * sizeof(float));
    hipMemset(A, 0x3c, (size_t)(rows_a + 128) * ld * sizeof(float));
    hipMemset(W, 0x3c, (size_t)(share * BN) * ld * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float *Cc = nullptr;
    const int ldc = share * BN;
    if (write_c) hipMalloc(&Cc, (size_t)(rows_a + 128) * ldc * sizeof(float));
    hipLaunchKernelGGL(k_stage_wide<NT>, dim3(grid), dim3(256), shm, 0, out, nk, A, W, ld, rows_a, share, Cc, ldc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_stage_wide<NT>, dim3(grid), dim3(256), shm, 0, out, nk, A, W, ld, rows_a, share, Cc, ldc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * grid * 4 * (double)nk * 16 * NT * 2048.0;
    printf("128 x %d tile (LDS %zu KB), K = %d, %d rows%s, workgroups/CU %d: %.1f TFLOP/s (%.2f ms) [%s]\n", BN, shm / 1024, nk * 32, rows_a, write_c ? ", result tile written" : "", blocks_per_cu, flop / ms / 1e9, ms / 4, hipGetErrorString(hipGetLastError()));
    if (Cc) hipFree(Cc);
    hipFree(out);
    hipFree(A);
    hipFree(W);
}

// operand fill: MFMA_PEAK_RANDOM=1 -> pseudo-random floats in [-0.5, 0.5) instead of the constant byte
// pattern (the switching activity of the multipliers, hence power and clock, depends on the operand bits)
__global__ void k_fill_rand(float *p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (float)(x >> 8) / 16777216.f - 0.5f;
    }
}
static void fill_operand(float *p, size_t n, unsigned seed) {
    if (getenv("MFMA_PEAK_RANDOM")) hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, p, n, seed);
    else hipMemset(p, 0x3c, n * sizeof(float));
    hipDeviceSynchronize();
}

// deeper prefetch at the same LDS footprint: 16-deep half-stages in a ring of four buffers (4 x 13.3 KB
// = the 53 KB of the two 32-deep buffers), three half-stages in flight behind counted vmcnt waits and
// raw barriers.  Rows are 64 B; chunk position = chunk ^ ((row >> 2) & 3) keeps the fragment reads
// conflict-free (rows four apart share banks).
template <int SPLIT>
__global__ __launch_bounds__(256, 3) void k_stage_deep(float *out, int nk, const float *A, const float *W, int ld, int rows_a, int share) {
    constexpr int HS = (128 + 80) * 16;             // floats per half-stage buffer
    __shared__ __attribute__((aligned(1024))) float lds[4 * HS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15, dr = lane >> 2, dp = lane & 3;   // DMA role: row within a 16-row group, chunk position
    auto swz = [](int row) { return (row >> 2) & 3; };
    int a_rd[2], w_rd[5];
    for (int mt = 0; mt < 2; ++mt) a_rd[mt] = (wave * 32 + mt * 16 + fr) * 16 + ((fq ^ swz(fr)) << 2);
    for (int nt = 0; nt < 5; ++nt) w_rd[nt] = 128 * 16 + (nt * 16 + fr) * 16 + ((fq ^ swz(fr)) << 2);
    const float *a_src[2], *w_src[2];
    const int m0 = ((blockIdx.x / share) * 128) % rows_a;
    for (int g = 0; g < 2; ++g) {
        const int row = wave * 32 + g * 16 + dr;
        a_src[g] = A + (size_t)(m0 + row) * ld + ((dp ^ swz(row)) << (SPLIT ? 3 : 2));
    }
    // weight rows: 5 groups of 16; wave 0 takes groups 0 and 4, waves 1..3 one group each
    for (int g = 0; g < 2; ++g) {
        const int grp = g == 0 ? wave : 4;
        const int row = grp * 16 + dr;
        w_src[g] = W + (size_t)((blockIdx.x % share) * 80 + row) * ld + ((dp ^ swz(row)) << (SPLIT ? 3 : 2));
    }
    const bool extra = wave == 0;
    auto issue = [&](int h, int buf) {
        // SPLIT: the half-stage takes the even / odd 16-byte chunks of its 32-deep stage (the chunk
        // partition of the production kernels: bit-identical sums), i.e. 16-byte pieces at a 32-byte stride
        const int koff = SPLIT ? (h >> 1) * 32 + (h & 1) * 4 : h * 16;
        float *base = lds + buf * HS;
#pragma unroll
        for (int g = 0; g < 2; ++g)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(a_src[g] + koff), (lds_void_t *)(base + (wave * 32 + g * 16) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[0] + koff), (lds_void_t *)(base + 128 * 16 + wave * 16 * 16), 16, 0, 0);
        if (extra)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[1] + koff), (lds_void_t *)(base + 128 * 16 + 4 * 16 * 16), 16, 0, 0);
    };
    f32x4 acc[5][2];
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nh = nk * 2;
    issue(0, 0);
    if (nh > 1) issue(1, 1);
    if (nh > 2) issue(2, 2);
    for (int h = 0; h < nh; ++h) {
        const int left = nh - 1 - h;                 // half-stages issued after h that may stay in flight (at most 2)
        if (left >= 2) {
            if (extra) __builtin_amdgcn_s_waitcnt(0xF78);
            else __builtin_amdgcn_s_waitcnt(0xF76);
        } else if (left == 1) {
            if (extra) __builtin_amdgcn_s_waitcnt(0xF74);
            else __builtin_amdgcn_s_waitcnt(0xF73);
        } else {
            __builtin_amdgcn_s_waitcnt(0xF70);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (h + 3 < nh) issue(h + 3, (h + 3) & 3);
        const int cur = (h & 3) * HS;
        f32x4 af[2], wf[5];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const f32x4 *>(&lds[cur + a_rd[mt]]);
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) wf[nt] = *reinterpret_cast<const f32x4 *>(&lds[cur + w_rd[nt]]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s], af[mt][s], acc[nt][mt], 0, 0, 0);
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int nt = 0; nt < 5; ++nt)
        for (int mt = 0; mt < 2; ++mt) sum += acc[nt][mt];
    out[blockIdx.x * 256 + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
}

template <int SPLIT>
void run_deep(int blocks_per_cu, int cus, int nk, int share, int rows_a) {
    const int grid = cus * blocks_per_cu, ld = nk * 32;
    float *out, *A, *W;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&A, (size_t)(rows_a + 128) * ld * sizeof(float));
    hipMalloc(&W, (size_t)(share * 80) * ld * sizeof(float));
    fill_operand(A, (size_t)(rows_a + 128) * ld, 1u);
    fill_operand(W, (size_t)(share * 80) * ld, 2u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_stage_deep<SPLIT>, dim3(grid), dim3(256), 0, 0, out, nk, A, W, ld, rows_a, share);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_stage_deep<SPLIT>, dim3(grid), dim3(256), 0, 0, out, nk, A, W, ld, rows_a, share);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * grid * 4 * (double)nk * 80 * 2048.0;
    printf("16-deep half-stages%s, ring of 4 buffers, 3 in flight, K = %d, %d share, %d rows, workgroups/CU %d: %.1f TFLOP/s (%.2f ms) [%s]\n", SPLIT ? " (even / odd chunks)" : "", nk * 32, share, rows_a, blocks_per_cu, flop / ms / 1e9, ms / 4, hipGetErrorString(hipGetLastError()));
    hipFree(out);
    hipFree(A);
    hipFree(W);
}

// fc1 -> LeakyReLU -> fc2 of one GAT layer in ONE launch (the round-1 review's item 7), as a timing model:
// a workgroup owns 64 rows; phase 1 computes h2 = leaky(A W1^T) in five 80-column passes (A and W1 tiles
// staged per 32-deep stage as in the production kernel) and keeps the 64 x 400 h2 tile in LDS (row stride
// 420 floats: conflict-free fragment reads); phase 2 computes h2 W2^T in five passes with only the W2 tile
// staged, its A fragments read from the h2 tile.  107.5 KB + 36 KB of LDS: one workgroup per CU.
__global__ __launch_bounds__(256, 1) void k_fused_fc12(float *out, int nk, const float *A, const float *W1, const float *W2, int ld, float *C) {
    constexpr int HSTR = 420, ROWS = 64, STG = (ROWS + 80) * 32;
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    float *h2 = lds;                                   // [64][420]
    float *stg = lds + ROWS * HSTR + 64;               // 2 stages of (64 + 80) rows x 32 floats (1 KiB aligned below)
    stg = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(stg) + 1023) & ~(uintptr_t)1023);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fq = lane >> 4, fr = lane & 15, dr = lane >> 3, dp = lane & 7;
    auto swz = [](int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); };
    const int m0 = blockIdx.x * ROWS;
    const int a_rd = (wave * 16 + fr) * 32;
    int w_rd[5];
    for (int nt = 0; nt < 5; ++nt) w_rd[nt] = ROWS * 32 + (nt * 16 + fr) * 32;
    const int c0 = ((fq * 2 + 0) ^ swz(fr)) << 2, c1 = ((fq * 2 + 1) ^ swz(fr)) << 2;
    const float *a_src[2];
    for (int g = 0; g < 2; ++g) {
        const int row = wave * 16 + g * 8 + dr;
        a_src[g] = A + (size_t)(m0 + row) * ld + ((dp ^ swz(row)) << 2);
    }
    for (int i = threadIdx.x; i < ROWS * HSTR; i += 256) h2[i] = 0.f;      // incl. the K padding columns
    __syncthreads();
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int phase = 0; phase < 2; ++phase) {
        const float *W = phase ? W2 : W1;
        for (int pass = 0; pass < 5; ++pass) {
            // weight rows of this pass: 10 groups of 8, dealt round-robin to the four waves
            const float *w_src[3];
            for (int g = 0; g < 3; ++g) {
                int grp = wave + 4 * g;
                if (grp > 9) grp = 9;
                const int row = grp * 8 + dr;
                w_src[g] = W + (size_t)(pass * 80 + row) * ld + ((dp ^ swz(row)) << 2);
            }
            auto issue = [&](int kt, int buf) {
                const int koff = kt * 32;
                float *base = stg + buf * STG;
                if (phase == 0) {
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        __builtin_amdgcn_global_load_lds((glb_void_t *)(a_src[g] + koff), (lds_void_t *)(base + (wave * 16 + g * 8) * 32), 16, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    if (wave + 4 * g < 10)
                        __builtin_amdgcn_global_load_lds((glb_void_t *)(w_src[g] + koff), (lds_void_t *)(base + ROWS * 32 + (wave + 4 * g) * 8 * 32), 16, 0, 0);
            };
            f32x4 acc[5];
            for (int nt = 0; nt < 5; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            issue(0, 0);
            for (int kt = 0; kt < nk; ++kt) {
                __syncthreads();
                if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
                const float *cur = stg + (kt & 1) * STG;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    f32x4 af, wf[5];
                    if (phase == 0) af = *reinterpret_cast<const f32x4 *>(cur + a_rd + (hh ? c1 : c0));
                    else af = *reinterpret_cast<const f32x4 *>(h2 + (wave * 16 + fr) * HSTR + kt * 32 + (fq * 2 + hh) * 4);
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) wf[nt] = *reinterpret_cast<const f32x4 *>(cur + w_rd[nt] + (hh ? c1 : c0));
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int nt = 0; nt < 5; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s], af[s], acc[nt], 0, 0, 0);
                }
            }
            __syncthreads();
            if (phase == 0) {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    f32x4 v = acc[nt];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * 0.15f;
                    *reinterpret_cast<f32x4 *>(h2 + (wave * 16 + fr) * HSTR + pass * 80 + nt * 16 + fq * 4) = v;
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    *reinterpret_cast<f32x4 *>(C + (size_t)(m0 + wave * 16 + fr) * 400 + pass * 80 + nt * 16 + fq * 4) = acc[nt];
                    total += acc[nt];
                }
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}

void run_fused_fc12(int cus, int rows) {
    const int nk = 13, ld = nk * 32, grid = rows / 64;
    const size_t shm = (size_t)(64 * 420 + 64 + 256 + 2 * (64 + 80) * 32) * sizeof(float) + 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_fc12), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    float *out, *A, *W1, *W2, *C;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&A, (size_t)(rows + 128) * ld * sizeof(float));
    hipMalloc(&W1, (size_t)(400 + 208) * ld * sizeof(float));
    hipMalloc(&W2, (size_t)(400 + 208) * ld * sizeof(float));
    hipMalloc(&C, (size_t)(rows + 128) * 400 * sizeof(float));
    fill_operand(A, (size_t)(rows + 128) * ld, 1u);
    fill_operand(W1, (size_t)(400 + 208) * ld, 2u);
    fill_operand(W2, (size_t)(400 + 208) * ld, 3u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fused_fc12, dim3(grid), dim3(256), shm, 0, out, nk, A, W1, W2, ld, C);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_fused_fc12, dim3(grid), dim3(256), shm, 0, out, nk, A, W1, W2, ld, C);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * grid * 2.0 * 5 * nk * (64.0 * 80 * 32 * 2);
    printf("fc1 -> fc2 fused, 64-row h2 tile in LDS (%zu KB, 1 workgroup/CU), K = N = 400 (padded 416), %d rows: %.1f TFLOP/s (%.2f ms for both GEMMs) [%s]\n",
           shm / 1024, rows, flop / ms / 1e9, ms / 4, hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(A); hipFree(W1); hipFree(W2); hipFree(C);
}

template <int MODE>
void run_stage(int blocks_per_cu, int cus, int nk, int share = 5, int rows_a = 65536) {
    const int grid = cus * blocks_per_cu, ld = nk * 32;
    float *out, *A, *W;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&A, (size_t)(rows_a + 128) * ld * sizeof(float));
    hipMalloc(&W, (size_t)(share * 80) * ld * sizeof(float));
    fill_operand(A, (size_t)(rows_a + 128) * ld, 1u);     // default 0x3c3c3c3c = 0.0115 as float
    fill_operand(W, (size_t)(share * 80) * ld, 2u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_stage_mfma<MODE>, dim3(grid), dim3(256), 0, 0, out, nk, A, W, ld, rows_a, share);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    long long *clk;
    hipMalloc(&clk, 16);
    hipMemset(clk, 0, 16);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_stage_mfma<MODE>, dim3(grid), dim3(256), 0, 0, out, nk, A, W, ld, rows_a, share, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long hclk[2];
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("   [one workgroup: %lld shader cycles in %lld wall ticks at %d kHz -> shader clock %.0f MHz]\n", hclk[0], hclk[1], wall_khz, wall_khz > 0 ? (double)hclk[0] / ((double)hclk[1] / wall_khz) / 1e3 : 0.0);
    const double flop = 4.0 * grid * 4 * (double)nk * 80 * 2048.0;
    printf("%s, K = %d, %d workgroups share an activation tile, %d rows, workgroups/CU %d: %.1f TFLOP/s (%.2f ms)\n", MODE == 3 ? "barrier + LDS-DMA of the activation tile only (4 per wave)" : MODE == 4 ? "barrier + LDS-DMA of the weight tile only (2.5 per wave)" : MODE == 2 ? "barrier + LDS-DMA, 3 buffers, 2 stages in flight" : MODE ? "barrier + LDS-DMA staging per stage" : "barrier per stage", nk * 32, share, rows_a, blocks_per_cu, flop / ms / 1e9, ms / 4);
    hipFree(out);
    hipFree(A);
    hipFree(W);
}

template <int PIPE>
void run_lds(int blocks_per_cu, int cus, const float *rnd) {
    const int grid = cus * blocks_per_cu, iters = 3000;
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lds_mfma<PIPE>, dim3(grid), dim3(256), 0, 0, out, 100, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_lds_mfma<PIPE>, dim3(grid), dim3(256), 0, 0, out, iters, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 80 * 2048.0;
    printf("LDS fragments + MFMA, %s, workgroups/CU %d: %.1f TFLOP/s (%.2f ms)\n", PIPE ? "reads one half-stage ahead" : "reads then MFMAs", blocks_per_cu, flop / ms / 1e9, ms);
    hipFree(out);
}

template <int NACC>
void run_rand(int blocks_per_cu, int cus) {
    const int grid = cus * blocks_per_cu, iters = 4000;
    float *out, *rnd;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&rnd, 4096 * sizeof(float));
    float h[4096];
    unsigned x = 12345;
    for (int i = 0; i < 4096; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = (float)(x >> 8) / 16777216.f - 0.5f;
    }
    hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma_rand<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, rnd);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_rand<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, rnd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * 8 * NACC * 2048.0;
        printf("random operands, accumulators/wave %2d, workgroups/CU %d: %.1f TFLOP/s (%.2f ms)\n", NACC, blocks_per_cu, flop / ms / 1e9, ms);
    }
    hipFree(out);
    hipFree(rnd);
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    if (argc > 1 && !strcmp(argv[1], "fused")) {
        // the two GEMMs of a GAT layer: two launches of the production loop vs one fused launch
        run_stage<1>(27, p.multiProcessorCount, 13, 5, 180224);
        run_wide<5>(27, p.multiProcessorCount, 13, 5, 180224, true);
        run_fused_fc12(p.multiProcessorCount, 180224);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "deep")) {
        // prefetch-depth comparison only
        run_stage<1>(3, p.multiProcessorCount, 96, 5, 8192);
        run_deep<0>(3, p.multiProcessorCount, 96, 5, 8192);
        run_deep<1>(3, p.multiProcessorCount, 96, 5, 8192);
        run_stage<1>(6, p.multiProcessorCount, 96, 38, 4096);
        run_deep<0>(6, p.multiProcessorCount, 96, 38, 4096);
        run_deep<1>(6, p.multiProcessorCount, 96, 38, 4096);
        run_stage<1>(27, p.multiProcessorCount, 13, 5, 180224);
        run_deep<0>(27, p.multiProcessorCount, 13, 5, 180224);
        run_deep<1>(27, p.multiProcessorCount, 13, 5, 180224);
        run_stage<1>(27, p.multiProcessorCount, 13, 5, 8192);
        run_deep<0>(27, p.multiProcessorCount, 13, 5, 8192);
        run_deep<1>(27, p.multiProcessorCount, 13, 5, 8192);
        return 0;
    }
    run<10>(1, p.multiProcessorCount);
    run<10>(2, p.multiProcessorCount);
    run<10>(3, p.multiProcessorCount);
    run<4>(3, p.multiProcessorCount);
    run<1>(4, p.multiProcessorCount);
    run_rand<10>(3, p.multiProcessorCount);
    run_rand<10>(1, p.multiProcessorCount);
    {
        float *rnd;
        hipMalloc(&rnd, 4096 * sizeof(float));
        float h[4096];
        unsigned x = 777;
        for (int i = 0; i < 4096; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = (float)(x >> 8) / 16777216.f - 0.5f;
        }
        hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
        for (int b = 1; b <= 3; ++b) run_lds<0>(b, p.multiProcessorCount, rnd);
        for (int b = 1; b <= 3; ++b) run_lds<1>(b, p.multiProcessorCount, rnd);
        run_stage<0>(3, p.multiProcessorCount, 96);
        run_stage<1>(3, p.multiProcessorCount, 96, 5, 65536);       // no reuse across workgroups: HBM bound
        run_stage<1>(6, p.multiProcessorCount, 96, 38, 4096);       // MLP-like: 4096 rows x 3040 features
        run_stage<1>(27, p.multiProcessorCount, 13, 5, 180224);     // GAT-like: 180k rows x 400 features
        run_stage<1>(27, p.multiProcessorCount, 13, 5, 8192);       // same tiles, operands cache resident
        run_stage<1>(2, p.multiProcessorCount, 96, 5, 8192);
        run_stage<2>(2, p.multiProcessorCount, 96, 5, 8192);
        run_stage<2>(26, p.multiProcessorCount, 13, 5, 8192);
        run_stage<1>(1, p.multiProcessorCount, 96, 5, 8192);
        run_stage<3>(3, p.multiProcessorCount, 96, 5, 8192);
        run_pipe(2, p.multiProcessorCount, 96, 5, 8192);
        run_wide<5>(3, p.multiProcessorCount, 96, 5, 8192);
        run_wide<10>(2, p.multiProcessorCount, 96, 5, 8192);
        run_wide<10>(1, p.multiProcessorCount, 96, 5, 8192);
        run_wide<13>(1, p.multiProcessorCount, 96, 2, 8192);
        run_wide<10>(16, p.multiProcessorCount, 13, 5, 180224);
        run_wide<13>(14, p.multiProcessorCount, 13, 2, 180224);
        run_wide<5>(27, p.multiProcessorCount, 13, 5, 180224);
        run_wide<5>(27, p.multiProcessorCount, 13, 5, 180224, true);
        run_wide<10>(16, p.multiProcessorCount, 13, 2, 180224, true);
        run_pipe(1, p.multiProcessorCount, 96, 5, 8192);
        run_pipe(26, p.multiProcessorCount, 13, 5, 180224);
        run_stage<4>(3, p.multiProcessorCount, 96, 5, 8192);
    }
    return 0;
}

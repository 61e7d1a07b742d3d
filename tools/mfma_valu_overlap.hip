// Do the vector instructions of one wave run under the MFMAs of the other wave of the same SIMD?  One 512-thread workgroup per CU
// (two waves per SIMD): waves 0-3 issue v_mfma_f32_16x16x32_bf16 (chains of six on one accumulator, as the split-bf16 GEMM), waves
// 4-7 issue plain v_add_f32 / v_cvt_f64_f32 + v_add_f64 streams.  Times: MFMA waves alone, vector waves alone, both together.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

template <int VOP>
__global__ __launch_bounds__(512) void k(float *out, int mfma_iters, int valu_iters, int who) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (!(who & 1)) return;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x + i); b[i] = (short)(0x3c10 + i * 7 + threadIdx.x); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);      // 48 MFMAs per iteration
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    } else {
        if (!(who & 2)) return;
        float f[16], g[16], h[16];
        double d[16];
        for (int i = 0; i < 16; ++i) { f[i] = 1.0f + i + threadIdx.x; d[i] = f[i] * 0.5; g[i] = f[i] * 3.f; h[i] = 0.f; }
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (VOP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
                if (VOP == 1) {
                    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[(i + 1) & 15]) : "v"(d[i]));
                }
                if (VOP == 2) {              // TwoSum of x = f[i] into (hi, lo) = (g[i], h[i]): seven fp32 instructions
                    float sm, bb, t1, t2;
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(sm) : "v"(g[i]), "v"(f[i]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(bb) : "v"(sm), "v"(g[i]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t1) : "v"(sm), "v"(bb));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t1) : "v"(g[i]), "v"(t1));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t2) : "v"(f[i]), "v"(bb));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(t1), "v"(t2));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(h[i]) : "v"(t1));
                    g[i] = sm;
                }
            }
        }
        for (int i = 0; i < 16; ++i) r += f[i] + (float)d[i] + g[i] + h[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int VOP>
static float run(float *out, int mi, int vi, int who) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VOP>, dim3(256), dim3(512), 0, 0, out, mi, vi, who);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VOP>, dim3(256), dim3(512), 0, 0, out, mi, vi, who);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    const int mi = 20000;                 // 48 MFMAs x 16 cycles = 768 cycles per iteration
    // the flush of the split-bf16 GEMM at its real ratio: 16 accumulators per 48 MFMAs, as f64 (cvt + add) or as fp32 TwoSum
    for (int vop = 1; vop <= 2; ++vop) {
        const float tm = vop == 1 ? run<1>(out, mi, mi, 1) : run<2>(out, mi, mi, 1);
        const float tv = vop == 1 ? run<1>(out, mi, mi, 2) : run<2>(out, mi, mi, 2);
        const float tb = vop == 1 ? run<1>(out, mi, mi, 3) : run<2>(out, mi, mi, 3);
        printf("flush of 16 accumulators per 48 MFMAs, %s: MFMA waves alone %.3f ms, flush waves alone %.3f ms, both %.3f ms\n",
               vop == 1 ? "v_cvt_f64_f32 + v_add_f64 (32 instructions)" : "fp32 TwoSum (112 instructions)        ", tm, tv, tb);
    }
    for (int vop = 0; vop < 2; ++vop) {
        // vector instructions per iteration: 16 (v_add_f32) or 32 (cvt + add f64)
        for (int ratio = 1; ratio <= 4; ratio *= 2) {
            const int vi = mi * 3 * ratio;      // vector work growing against the MFMA work
            const float tm = vop ? run<1>(out, mi, vi, 1) : run<0>(out, mi, vi, 1);
            const float tv = vop ? run<1>(out, mi, vi, 2) : run<0>(out, mi, vi, 2);
            const float tb = vop ? run<1>(out, mi, vi, 3) : run<0>(out, mi, vi, 3);
            printf("%s  vector iterations per MFMA iteration %2d: MFMA waves alone %.3f ms, vector waves alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n",
                   vop ? "v_cvt_f64_f32 + v_add_f64" : "v_add_f32              ", 3 * ratio, tm, tv, tb, tm + tv, tm > tv ? tm : tv);
        }
    }
    return 0;
}

// Issue cost (cycles per wave-instruction, one wave per SIMD, independent operands) of the vector instructions the f64 flush of the
// split-bf16 GEMM is made of, and of the candidates to replace them.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float f[16];
    double d[16];
    for (int i = 0; i < 16; ++i) { f[i] = seed + i + threadIdx.x; d[i] = f[i] * 0.5; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
            if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 4) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
            if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 15]), "v"(f[(i + 2) & 15]));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
            if (OP == 7) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 15]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
template <int OP>
void run(const char *name, float *out) {
    const int iters = 4096;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    float c;
    hipMemcpy(&c, out + (1 << 20), 4, hipMemcpyDeviceToHost);
    printf("%-20s %.2f cycles per wave-instruction (s_memtime ticks; one wave per SIMD)\n", name, c / (iters * 16.0));
}
int main() {
    float *out;
    hipMalloc(&out, ((1 << 20) + 16) * 4);
    run<0>("v_add_f32", out); run<1>("v_add_f64", out); run<2>("v_cvt_f64_f32", out); run<3>("v_cvt_f32_f64", out);
    run<4>("v_fma_f64", out); run<5>("v_cvt_pk_bf16_f32", out); run<6>("v_pk_add_f32", out); run<7>("v_lshlrev_b32", out);
    return 0;
}

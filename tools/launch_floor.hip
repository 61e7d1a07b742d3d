// What a dependent launch costs by its shape: a chain of N launches of a kernel that does (almost) nothing, per-launch time by
// workgroup size, LDS bytes per workgroup and grid size.  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern "C" __global__ void k_nop(float *out, int n) {
    extern __shared__ float s[];
    if (n < 0) {                      // never: keeps the LDS allocation alive
        s[threadIdx.x] = out[threadIdx.x];
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x + 1) % blockDim.x];
    }
}
// the same with `work` dependent vector instructions per wave (a stand-in for straight-line code that runs once)
template <int WORK>
__global__ void k_work(float *out, int n) {
    extern __shared__ float s[];
    float x = (float)threadIdx.x;
#pragma unroll
    for (int i = 0; i < WORK; ++i) x = __builtin_fmaf(x, 1.0001f, (float)i);
    if (n < 0 || x == 12345.678f) out[threadIdx.x] = x + s[0];
}
#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    float *d;
    CHK(hipMalloc(&d, 1 << 20));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipFuncSetAttribute((const void *)k_nop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 2000;
    printf("%-10s %8s %8s %8s  us per dependent launch\n", "kernel", "threads", "lds KB", "grid");
    for (int threads : {64, 256, 512, 1024})
        for (int lds : {0, 16, 32, 48, 64, 96, 160})
            for (int grid : {4, 64, 256}) {
                if (lds > 64 && threads < 512) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    CHK(hipEventRecord(e0, st));
                    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_nop, dim3(grid), dim3(threads), (size_t)lds * 1024, st, d, 1);
                    CHK(hipEventRecord(e1, st));
                    CHK(hipStreamSynchronize(st));
                    float ms;
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) printf("%-10s %8d %8d %8d  %7.2f\n", "nop", threads, lds, grid, ms * 1e3 / N);
                }
            }
#define WORK(W_)                                                                                                    \
    for (int threads : {256, 1024}) {                                                                               \
        for (int rep = 0; rep < 2; ++rep) {                                                                         \
            CHK(hipEventRecord(e0, st));                                                                            \
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_work<W_>, dim3(64), dim3(threads), 0, st, d, 1);       \
            CHK(hipEventRecord(e1, st));                                                                            \
            CHK(hipStreamSynchronize(st));                                                                          \
            float ms;                                                                                               \
            CHK(hipEventElapsedTime(&ms, e0, e1));                                                                  \
            if (rep) printf("work%-6d %8d %8d %8d  %7.2f\n", W_, threads, 0, 64, ms * 1e3 / N);                      \
        }                                                                                                           \
    }
    WORK(256) WORK(1024) WORK(4096)
    return 0;
}

// GPU-side cost of a dependent kernel launch on one stream: N back-to-back launches of (a) an empty kernel, (b) a kernel that asks for
// 160 KB of LDS and 512 registers (the z192 footprint) but returns at once, timed with events; and the same through a captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void empty_k(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(256) void big_k(int* p) {
    extern __shared__ char smem[];
    if (p && threadIdx.x == 9999) { smem[threadIdx.x] = 1; p[0] = smem[0]; }
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(big_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, st, nullptr);
                if (mode == 1) hipLaunchKernelGGL(big_k, dim3(256), dim3(256), 160 * 1024, st, nullptr);
                if (mode == 2) hipLaunchKernelGGL(empty_k, dim3(2048), dim3(256), 0, st, nullptr);
                if (mode == 3) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, st, nullptr);
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const char* names[4] = {"empty kernel, 256 x 256 threads", "160 KB LDS kernel, 256 x 256 threads", "empty kernel, 2048 x 256 threads", "empty kernel, 1 x 64 threads"};
        printf("%-40s %6.2f us per dependent launch\n", names[mode], best * 1e3f / N);
    }
    return 0;
}

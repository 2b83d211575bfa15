// Does v_exp_f32 run beside plain VALU instructions of the SAME wave (separate transcendental pipe) or does it hold the VALU for its
// whole quarter-rate duration?  One wave per SIMD, straight-line bodies, s_memrealtime-free: wall time over many iterations, ratios only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define E4 "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
#define F4 "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
#define EF "v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n" \
           "v_exp_f32 %1, %1\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n" \
           "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %4, %4, %4, %4\n" \
           "v_exp_f32 %3, %3\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
#define E1F1 "v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %7, %7\n"
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = a + 1, c = a + 2, d = a + 3, e = 0.5f, f = 0.25f, g = 0.125f, h = 0.75f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(E4 E4 E4 E4 OPS);                 // 16 exp
        if (MODE == 1) asm volatile(F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 OPS);     // 48 fma
        if (MODE == 2) asm volatile(EF EF EF EF OPS);                 // 16 exp + 48 fma, one exp every 4th instruction
        if (MODE == 3) asm volatile(E4 E4 E4 E4 F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 OPS);   // 16 exp then 48 fma
        if (MODE == 4) asm volatile(E1F1 E1F1 E1F1 E1F1 OPS);         // 16 exp + 16 fma alternating
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[5] = {"16 exp", "48 fma", "16 exp + 48 fma (exp every 4th)", "16 exp then 48 fma", "16 exp + 16 fma alternating"};
    float t[5];
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 5; ++m) {
            CK(hipEventRecord(e0));
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t[m], e0, e1));
        }
    for (int m = 0; m < 5; ++m) printf("%-36s %8.3f ms   %6.2f ns per body\n", names[m], t[m], t[m] * 1e6 / iters);
    printf("per instruction (ns): exp %.3f   fma %.3f   ->  exp / fma = %.2f;  interleaved body / (exp body + fma body) = %.2f\n",
           t[0] * 1e6 / iters / 16, t[1] * 1e6 / iters / 48, (t[0] / 16) / (t[1] / 48), t[2] / (t[0] + t[1]));
    return 0;
}

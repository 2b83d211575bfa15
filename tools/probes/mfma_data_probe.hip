// Does the issue interval / the sustained clock of v_mfma_f32_32x32x16_{f16,bf16} depend on the operand DATA?
// One wave per SIMD, 12 independent accumulators, 256 workgroups; operands from a buffer filled with
//   0 zeros | 1 ones | 2 multiples of 1/64 in [-1.56, 1.55] (the pattern of the round-1 probes) | 3 uniform +-0.06 (fp16
//   denormals present, as in real weights) | 4 uniform +-0.06 with |x| < 2^-12 forced to +-2^-12 (no denormals) | 5 N(0,1)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void mfma_kernel(const uint4* src, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 a[4], b[3];
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 7 + wave * 64 + lane + i * 256) & 16383];
    for (int i = 0; i < 3; ++i) b[i] = src[(blockIdx.x * 11 + wave * 64 + lane + i * 256 + 4096) & 16383];
    f32x16 c[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[i >> 2]), c[i], 0, 0, 0);
            else c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[i >> 2]), c[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, const uint4* src, float* out, unsigned long long* cyc) {
    const int iters = 4000, nb = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mfma_kernel<KIND>), dim3(nb), dim3(256), 0, 0, src, out, cyc, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nb;
    const double n = (double)iters * 12;
    printf("  %-34s %6.2f ticks per MFMA   %7.1f TFLOP/s   [%.2f GHz]\n", name, avg / n, 32768.0 * n * 4 * nb / best * 1e-9, avg / best * 1e-6);
}

int main() {
    uint4* src; float* out; unsigned long long* cyc;
    const size_t n16 = 16384 * 8;
    CK(hipMalloc(&src, n16 * 2)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 256 * 8));
    std::mt19937 rng(7); std::uniform_real_distribution<float> ud(-1.f, 1.f); std::normal_distribution<float> nd(0.f, 1.f);
    for (int bf = 0; bf < 2; ++bf)
    for (int pat = 0; pat < 6; ++pat) {
        std::vector<unsigned short> h(n16);
        for (size_t i = 0; i < n16; ++i) {
            float v = 0.f;
            if (pat == 1) v = 1.f;
            if (pat == 2) v = (((int)(i * 2654435761u >> 20) % 200) - 100) / 64.0f;
            if (pat == 3 || pat == 4) { v = 0.06f * ud(rng); if (pat == 4 && fabsf(v) < 2.44140625e-4f) v = v < 0 ? -2.44140625e-4f : 2.44140625e-4f; }
            if (pat == 5) v = nd(rng);
            if (bf) { unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
            else { _Float16 f = (_Float16)v; memcpy(&h[i], &f, 2); }
        }
        CK(hipMemcpy(src, h.data(), n16 * 2, hipMemcpyHostToDevice));
        const char* names[6] = {"zeros", "ones", "k/64 in [-1.56,1.55]", "uniform +-0.06 (denormals)", "uniform +-0.06, no denormals", "N(0,1)"};
        char buf[128]; snprintf(buf, sizeof buf, "%s %s", bf ? "bf16" : "f16 ", names[pat]);
        if (bf) run<1>(buf, src, out, cyc); else run<0>(buf, src, out, cyc);
    }
    return 0;
}

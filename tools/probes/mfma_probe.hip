// MFMA issue-rate probe: cycles per instruction per SIMD for v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 with
// 1 or 2 waves per SIMD (independent accumulators, random-ish operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NACC>
__global__ void mfma_kernel(const _Float16* src, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = *reinterpret_cast<const f16x8*>(src + (lane + i * 64) * 8); b[i] = *reinterpret_cast<const f16x8*>(src + (lane + i * 64 + 256) * 8); }
    f32x4 c4[NACC]; f32x16 c16[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 4; ++r) c4[i][r] = 0.f; for (int r = 0; r < 16; ++r) c16[i][r] = 0.f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], c4[i], 0, 0, 0);
            else c16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 3], c16[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 4; ++r) s += c4[i][r]; for (int r = 0; r < 16; ++r) s += c16[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC>
void run(const char* name, int threads, const _Float16* src, float* out, unsigned long long* cyc) {
    const int iters = 2000, nb = 256;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((mfma_kernel<KIND, NACC>), dim3(nb), dim3(threads), 0, 0, src, out, cyc, iters); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nb;
    const double per_simd = (double)iters * NACC * (threads / 256.0);      // MFMAs issued per SIMD (threads/256 waves per SIMD)
    const double flop = KIND == 0 ? 16384.0 : 32768.0;
    printf("%-22s %d waves/SIMD, %2d accumulators: %6.2f clk per MFMA per SIMD  (%5.1f %% of the 1024 FLOP/clk/SIMD peak)\n", name, threads / 256, NACC,
           avg / per_simd, 100.0 * flop / (avg / per_simd) / 1024.0);
}

int main() {
    _Float16* src; float* out; unsigned long long* cyc;
    CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 256 * 8));
    std::vector<_Float16> h(1 << 19); for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 64.0f);
    CK(hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice));
    run<0, 12>("16x16x32 f16", 256, src, out, cyc);
    run<0, 12>("16x16x32 f16", 512, src, out, cyc);
    run<0, 4>("16x16x32 f16", 512, src, out, cyc);
    run<1, 6>("32x32x16 f16", 256, src, out, cyc);
    run<1, 6>("32x32x16 f16", 512, src, out, cyc);
    run<1, 2>("32x32x16 f16", 512, src, out, cyc);
    return 0;
}

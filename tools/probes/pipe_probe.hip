// Can ONE wave per SIMD keep the matrix pipe busy while it also issues its own fragment ds_reads (software-pipelined by
// one k-step) under hipcc?  4 waves / workgroup, 1 workgroup / CU, wave tile 128(M) x 96(N) as 4 x 3 tiles of
// v_mfma_f32_32x32x16_f16 (192 accumulator registers), fragments double-buffered in registers (7 ds_read_b128 per k16 step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: reads + MFMA interleaved, 1: MFMA only, 2: reads only
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void pipe_kernel(const _Float16* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(src)[i & 4095];
    __syncthreads();
    // swizzled like the GEMM operand tiles: row r (128 B), 16-byte chunk c stored at c ^ ((r >> 1) & 7)  -> conflict-free b128 reads
    const int frow = lane & 31, fkey = (frow >> 1) & 7;
    const char* base = smem + wave * 4096 + frow * 128;
    int choff[4];
    for (int ks = 0; ks < 4; ++ks) choff[ks] = (((ks * 2 + (lane >> 5)) ^ fkey) << 4);
    f32x16 acc[4][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 fa[2][4], fb[2][3];
#define RD(buf, off) { \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const f16x8*>(base + choff[off] + i * 16384); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) fb[buf][j] = *reinterpret_cast<const f16x8*>(base + choff[off] + 65536 + j * 4096 * 4 / 4 * 1); }
#define MM(buf) { \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0); }
    RD(0, 0)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) RD(1, 1)
        if (MODE != 2) MM(0) else { for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[0][i])); for (int j = 0; j < 3; ++j) asm volatile("" :: "v"(fb[0][j])); }
        if (MODE != 1) RD(0, 2)
        if (MODE != 2) MM(1) else { for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[1][i])); for (int j = 0; j < 3; ++j) asm volatile("" :: "v"(fb[1][j])); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, const _Float16* src, float* out, unsigned long long* cyc) {
    const int iters = 2000, nb = 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pipe_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((pipe_kernel<MODE>), dim3(nb), dim3(256), 100 * 1024, 0, src, out, cyc, iters); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nb;
    printf("%-28s %7.1f clk per k16 step (12 MFMA + 7 ds_read_b128): %5.1f %% of the matrix peak\n", name, avg / (iters * 2.0), 100.0 * 12 * 32 / (avg / (iters * 2.0)));
}

int main() {
    _Float16* src; float* out; unsigned long long* cyc;
    CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 256 * 8));
    std::vector<_Float16> h(1 << 19); for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 64.0f);
    CK(hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice));
    run<0>("reads + MFMA (pipelined)", src, out, cyc);
    run<1>("MFMA only", src, out, cyc);
    run<2>("reads only", src, out, cyc);
    return 0;
}

// LDS-DMA (buffer_load ... lds) throughput probe: one 512-thread workgroup per CU streams an L2-resident buffer into
// LDS with different per-instruction access shapes; reports bytes/clk/CU.  No compute, no LDS reads.
//   shape 0: 1 KiB contiguous per instruction (64 lanes x 16 B)
//   shape 1: 8 rows x 128 B, row stride = ld bytes (the GEMM operand tile shape)
//   shape 2: as 1 with the 16-byte chunks of each row XOR-permuted (the swizzled source the GEMM uses)
//   shape 3: 16 rows x 64 B
//   shape 4: 4 rows x 256 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void* lds_vptr;

template <int SHAPE, int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, size_t span, int ld, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    int vo;
    if (SHAPE == 0) vo = lane * 16;
    else if (SHAPE == 1) vo = (lane >> 3) * ld + (lane & 7) * 16;
    else if (SHAPE == 2) vo = (lane >> 3) * ld + (((lane & 7) ^ ((lane >> 4) & 7)) << 4);
    else if (SHAPE == 3) vo = (lane >> 2) * ld + (lane & 3) * 16;
    else vo = (lane >> 4) * ld + (lane & 15) * 16;
    // each workgroup walks its own window of the buffer (span bytes, L2-resident across workgroups of an XCD)
    const unsigned base = (unsigned)((blockIdx.x % 8) * 1024 * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned off = wave * 8 * (SHAPE == 0 ? 128 : ld);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(smem + ((it * DEPTH + d) & 15) * 8192 + wave * 1024), 16, vo,
                                                     base + off, 0, 0);
            off += 64 * 8 * (SHAPE == 0 ? 128 : ld) / 8;   // advance 64 rows (or 8 KiB) per step
            if (off >= span) off -= span;
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// mixed: every wave alternates a burst of DMA pieces with a burst of ds_read_b128 (what the GEMM's read segments do)
template <int NREAD>
__global__ __launch_bounds__(512) void dma_mixed_kernel(const char* src, size_t span, int ld, int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int vo = (lane >> 3) * ld + (((lane & 7) ^ ((lane >> 4) & 7)) << 4);
    const unsigned base = (unsigned)((blockIdx.x % 8) * 1024 * 1024);
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned off = wave * 8 * ld;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(smem + ((it * 7 + d) & 7) * 8192 + wave * 1024), 16, vo, base + off, 0, 0);
            off += 64 * ld;
            if (off >= span) off -= span;
        }
#pragma unroll
        for (int r = 0; r < NREAD; ++r) {
            const f4 v = *reinterpret_cast<const f4*>(smem + 65536 + ((r * 4096 + (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4)) & 65535));
            acc += v;
        }
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[0] == 12345.f) sink[0] = acc[1];
}

template <int NREAD>
void run_mixed(const char* src, int ld, unsigned long long* dout, int ncu) {
    const int iters = 400;
    const size_t span = 768 * 1024;
    float* sink; CK(hipMalloc(&sink, 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_mixed_kernel<NREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((dma_mixed_kernel<NREAD>), dim3(ncu), dim3(512), 131072, 0, src, span, ld, iters, dout, sink);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(ncu);
    CK(hipMemcpy(h.data(), dout, ncu * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= ncu;
    const double bytes = (double)iters * 7 * 8 * 1024;
    printf("mixed: 7 DMA pieces + %2d ds_read_b128 per wave per step + barrier: %6.1f DMA B/clk/CU, %6.0f clk per step, LDS read %6.1f B/clk/CU\n",
           NREAD, bytes / avg, avg / iters, (double)iters * NREAD * 8 * 1024 / avg);
}

template <int SHAPE, int DEPTH>
void run(const char* name, const char* src, int ld, unsigned long long* dout, int ncu) {
    const int iters = 400;
    const size_t span = 768 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<SHAPE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((dma_kernel<SHAPE, DEPTH>), dim3(ncu), dim3(512), 131072, 0, src, span, ld, iters, dout);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(ncu);
    CK(hipMemcpy(h.data(), dout, ncu * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= ncu;
    const double bytes = (double)iters * DEPTH * 8 * 1024;
    printf("%-34s depth %d ld %5d: %7.1f B/clk/CU  (%.0f clk per 1 KiB piece per CU)\n", name, DEPTH, ld, bytes / avg, avg / (iters * DEPTH * 8.0));
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    char* src; CK(hipMalloc(&src, 64 << 20)); CK(hipMemset(src, 1, 64 << 20));
    unsigned long long* dout; CK(hipMalloc(&dout, ncu * 8));
    run<0, 4>("contiguous 1 KiB", src, 128, dout, ncu);
    run<0, 8>("contiguous 1 KiB", src, 128, dout, ncu);
    run<1, 4>("8 rows x 128 B", src, 1536, dout, ncu);
    run<1, 8>("8 rows x 128 B", src, 1536, dout, ncu);
    run<1, 8>("8 rows x 128 B", src, 6144, dout, ncu);
    run<2, 8>("8 rows x 128 B swizzled", src, 1536, dout, ncu);
    run<3, 8>("16 rows x 64 B", src, 1536, dout, ncu);
    run<4, 8>("4 rows x 256 B", src, 1536, dout, ncu);
    run<1, 8>("8 rows x 128 B, 64 CUs", src, 1536, dout, 64);
    run<0, 8>("contiguous 1 KiB, 64 CUs", src, 128, dout, 64);
    run_mixed<0>(src, 1536, dout, ncu);
    run_mixed<8>(src, 1536, dout, ncu);
    run_mixed<22>(src, 1536, dout, ncu);
    run_mixed<44>(src, 1536, dout, ncu);
    return 0;
}

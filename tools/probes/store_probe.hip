// What does one buffer_store_dwordx4 cost a CU as a function of the cache lines it touches?
// 256 workgroups x 4 waves; every wave issues NS stores of 1 KiB (64 lanes x 16 B) with a given lane -> address pattern:
//   0: 32 rows x 32 B  (the MFMA accumulator layout after v_permlane32_swap: what z192 / q192 store)
//   1: 16 rows x 64 B      2: 8 rows x 128 B (full aligned lines)      3: 5.33 rows x 192 B (1.5 lines per row)
//   4: 10.67 rows x 96 B   5: 2.67 rows x 384 B (3 full lines per row)
// Row stride = 6144 B (fc1's output: N = 3072 fp16).  Reports clk per store instruction per CU (all four waves storing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(char* out, unsigned long long* cyc, int ns, int ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    int row, col;       // byte column
    if (PAT == 0) { row = lane & 31; col = (lane >> 5) * 16; }
    else if (PAT == 1) { row = lane & 15; col = (lane >> 4) * 16; }
    else if (PAT == 2) { row = lane >> 3; col = (lane & 7) * 16; }
    else if (PAT == 3) { row = lane / 12; col = (lane % 12) * 16; }
    else if (PAT == 4) { row = lane / 6; col = (lane % 6) * 16; }
    else { row = lane / 24; col = (lane % 24) * 16; }
    // every workgroup owns a 256-row x 384-B tile region; each wave a 128-row x 192-B quadrant (like the GEMM)
    const int tile = blockIdx.x;
    const int base = ((tile >> 4) * 256 + (wave >> 1) * 128) * ld + (tile & 15) * 384 + (wave & 1) * 192;
    const int voff = base + row * ld + col;
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ns; ++i) {
        const int soff = (i & 3) * 32 * ld + ((i >> 2) & 1) * (PAT == 0 ? 32 : PAT == 1 ? 64 : PAT == 4 ? 96 : 0);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT> void run(const char* name, char* out, unsigned long long* cyc) {
    const int ns = 24, nb = 256, ld = 6144;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((store_kernel<PAT>), dim3(nb), dim3(256), 0, 0, out, cyc, ns, ld); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nb;
    printf("  %-52s %7.0f ticks for %d stores per wave = %6.1f ticks per store instruction per CU\n", name, avg, ns, avg / (ns * 4));
}

int main() {
    char* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)4096 * 6144 + (1 << 20))); CK(hipMalloc(&cyc, 256 * 8));
    run<0>("32 rows x 32 B (accumulator layout)", out, cyc);
    run<1>("16 rows x 64 B", out, cyc);
    run<2>("8 rows x 128 B (aligned full lines)", out, cyc);
    run<3>("5.33 rows x 192 B", out, cyc);
    run<4>("10.67 rows x 96 B", out, cyc);
    run<5>("2.67 rows x 384 B", out, cyc);
    return 0;
}

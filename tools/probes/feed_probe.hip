// What does feeding the matrix pipe cost on one CU?  One workgroup per CU runs the k-loop skeleton of the 256 x 192 GEMM
// (per 64-wide k-tile: 4 k16 steps of 7 ds_read_b128 + 12 v_mfma_f32_32x32x16_f16 per wave, one s_barrier) with each of the
// three ingredients switchable, and the LDS-DMA (buffer_load ... lds, 1 KiB per instruction, NP per k-tile and CU) issued
//   own-burst   : by the four compute waves, all of a k-tile's pieces right after the barrier (gemm_w192 / gemm_q192),
//   own-spread  : by the four compute waves, one piece every few MFMAs,
//   producer    : by a fifth wave that does nothing else (the compute waves then hold a 128 x 64 tile: 256 registers max).
// Reported: core clocks per k-tile (wall time x 2.4 GHz) against the 1536 (TJ = 3) / 1024 (TJ = 2) clocks the MFMAs need.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_vptr;

static int g_nreg = 4;         // distinct operand panels (688 KiB each): 4 stay L2-resident per XCD, 32 do not
constexpr int BUF = 57344;      // one k-tile: 24 KiB W + 32 KiB X
constexpr int LD = 1536;        // operand row stride in bytes (K = 768 fp16)

// RD: ds_reads on, MM: MFMAs on, DMA: 0 none, 1 own-burst, 2 own-spread, 3 producer wave; TJ: wave tile 128 x 32 TJ
template <int RD, int MM, int DMA, int TJ, int NP>
__global__ __launch_bounds__(DMA == 3 ? 320 : 256) __attribute__((amdgpu_waves_per_eu(1, DMA == 3 ? 2 : 1)))
void feed_kernel(const _Float16* src, float* out, int ktiles, unsigned long long* cyc, int nreg) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2 * BUF / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(src)[i & 4095];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int prow = lane >> 3, pc = lane & 7;
    // panels: 4 regions of 448 rows x 768 fp16, region chosen so that every XCD sees all four (L2-resident working set)
    const int region = (blockIdx.x >> 3) & (nreg - 1);
    const int vo = region * 448 * LD + prow * LD + ((pc ^ ((prow >> 1) & 7)) << 4);
    constexpr int PW = DMA == 3 ? NP : NP / 4;           // pieces per issuing wave and k-tile
#define PIECE(q, buf, kt) { const int pid_ = (DMA == 3 ? (q) : wave + 4 * (q)) % 56; \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(smem + (buf) * BUF + pid_ * 1024), 16, vo, pid_ * 8 * LD + ((kt) % 12) * 128, 0, 0); }

    if (DMA == 3 && wave == 4) {                        // the producer wave
        for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
            for (int q = 0; q < PW; ++q) PIECE(q, (kt & 1) ^ 1, kt)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    const int frow = lane & 31, fkey = (frow >> 1) & 7, fhalf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int choff[4];
    for (int ks = 0; ks < 4; ++ks) choff[ks] = (((ks * 2 + fhalf) ^ fkey) << 4);
    const char* const wbase = smem + (wn * 32 * TJ + frow) * 128;
    const char* const xbase = smem + 24576 + (wm * 128 + frow) * 128;
    f32x16 acc[TJ][4];
    for (int i = 0; i < TJ; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 fw[2][TJ], fx[2][4];
#define RDF(set, buf, ks) { if (RD) { \
    _Pragma("unroll") for (int i_ = 0; i_ < TJ; ++i_) fw[set][i_] = *reinterpret_cast<const f16x8*>(wbase + (buf) * BUF + i_ * 4096 + choff[ks]); \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) fx[set][j_] = *reinterpret_cast<const f16x8*>(xbase + (buf) * BUF + j_ * 4096 + choff[ks]); } }
    // NQ pieces q0 .. q0 + NQ - 1 woven into the TJ * 4 MFMAs of one k-step
#define MMD(set, q0, NQ, buf, kt) { \
    _Pragma("unroll") for (int i_ = 0; i_ < TJ; ++i_) \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { \
        if (MM) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fw[set][i_]), "v"(fx[set][j_])); \
        else if (RD) { asm volatile("" :: "v"(fw[set][i_]), "v"(fx[set][j_])); } \
        const int m_ = i_ * 4 + j_; \
        if ((NQ) > 0 && (m_ + 1) * (NQ) / (TJ * 4) > m_ * (NQ) / (TJ * 4)) { PIECE((q0) + m_ * (NQ) / (TJ * 4), buf, kt) } } }
    for (int i = 0; i < TJ; ++i) fw[0][i] = fw[1][i] = *reinterpret_cast<const f16x8*>(wbase + i * 4096);
    for (int j = 0; j < 4; ++j) fx[0][j] = fx[1][j] = *reinterpret_cast<const f16x8*>(xbase + j * 4096);
    constexpr int S3 = PW / 2, S0 = (PW - S3 + 1) / 2, S1 = PW - S3 - S0;     // spread: pieces per k-step 3 / 0 / 1
    for (int kt = 0; kt < ktiles; kt += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            RDF(1, b, 1)
            if (DMA == 2) MMD(0, S3, S0, b ^ 1, kt + b + 1) else MMD(0, 0, 0, b, kt)
            RDF(0, b, 2)
            if (DMA == 2) MMD(1, S3 + S0, S1, b ^ 1, kt + b + 1) else MMD(1, 0, 0, b, kt)
            RDF(1, b, 3)
            MMD(0, 0, 0, b, kt)
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            RDF(0, b ^ 1, 0)
            if (DMA == 1) {
#pragma unroll
                for (int q = 0; q < PW; ++q) PIECE(q, b, kt + b + 2)
            }
            if (DMA == 2) MMD(1, 0, S3, b, kt + b + 2) else MMD(1, 0, 0, b, kt)
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < TJ; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int RD, int MM, int DMA, int TJ, int NP>
void run(const char* name, const _Float16* src, float* out) {
    static unsigned long long* cyc = nullptr; if (!cyc) CK(hipMalloc(&cyc, 256 * 8));
    const int ktiles = 4000, nb = 256, threads = DMA == 3 ? 320 : 256;
    auto kern = feed_kernel<RD, MM, DMA, TJ, NP>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 2 * BUF, 0, src, out, ktiles, cyc, g_nreg);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double clk = best * 1e-3 * 2.4e9 / ktiles;
    printf("%-52s %7.0f clk / k-tile", name, clk);
    if (MM) printf("   %5.1f %% of the matrix peak", 100.0 * TJ * 16 * 32 / clk);
    if (DMA) printf("   DMA %5.1f B/clk/CU", NP * 1024.0 / clk);
    std::vector<unsigned long long> hc(nb); CK(hipMemcpy(hc.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : hc) avg += (double)v; avg /= nb;
    printf("   [s_memtime: %.0f ticks / k-tile = %.2f GHz]\n", avg / ktiles, avg / (best * 1e-3) * 1e-9);
}

int main() {
    _Float16* src; float* out;
    CK(hipMalloc(&src, 32 << 20)); CK(hipMalloc(&out, 1 << 22));
    std::vector<_Float16> h(16 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 64.0f);
    CK(hipMemcpy(src, h.data(), 32 << 20, hipMemcpyHostToDevice));
    printf("--- wave tile 128 x 96 (4 waves, 1 per SIMD), 56 pieces per k-tile\n");
    run<1, 1, 0, 3, 56>("reads + MFMA", src, out);
    run<0, 1, 0, 3, 56>("MFMA", src, out);
    run<1, 0, 0, 3, 56>("reads", src, out);
    run<0, 0, 1, 3, 56>("DMA burst", src, out);
    run<1, 0, 1, 3, 56>("reads + DMA burst", src, out);
    run<0, 1, 1, 3, 56>("MFMA + DMA burst", src, out);
    run<0, 1, 2, 3, 56>("MFMA + DMA spread", src, out);
    run<1, 1, 1, 3, 56>("reads + MFMA + DMA burst", src, out);
    run<1, 1, 2, 3, 56>("reads + MFMA + DMA spread", src, out);
    run<1, 1, 1, 3, 28>("reads + MFMA + DMA burst, half the pieces", src, out);
    printf("--- wave tile 128 x 64 (4 compute waves), 48 pieces per k-tile\n");
    run<1, 1, 0, 2, 48>("reads + MFMA", src, out);
    run<1, 1, 1, 2, 48>("reads + MFMA + DMA burst", src, out);
    run<1, 1, 2, 2, 48>("reads + MFMA + DMA spread", src, out);
    run<1, 1, 3, 2, 48>("reads + MFMA + DMA from a producer wave", src, out);
    run<0, 1, 3, 2, 48>("MFMA + DMA from a producer wave", src, out);
    run<0, 0, 3, 2, 48>("DMA from a producer wave alone", src, out);
    printf("--- 128 x 96 again with 32 distinct panels (22 MiB: misses the 4 MiB L2 of an XCD, served by MALL / HBM)\n");
    g_nreg = 32;
    run<0, 0, 1, 3, 56>("DMA burst", src, out);
    run<1, 1, 1, 3, 56>("reads + MFMA + DMA burst", src, out);
    run<1, 1, 2, 3, 56>("reads + MFMA + DMA spread", src, out);
    return 0;
}

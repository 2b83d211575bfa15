// What does the operand feed of the persistent 256 x (192 | 256) GEMM cost on the REAL access pattern?
// feed_probe.hip answered it for L2-resident panels (48 B/clk/CU burst, 1890 ticks per k-tile with the pieces woven between
// the MFMAs); the real kernels (q192 / w192) draw ~21 B/clk/CU.  This probe runs the one-wave-per-SIMD k-loop skeleton
// (4 waves, wave tile 128 x 32 TJ, v_mfma_f32_32x32x16_f16, fragments read one k-step ahead) over the real tile walk of a
// [M,K] x [N,K]^T layer on all 256 CUs, with
//   DMA   0 none | 1 burst after the k-tile barrier | 2 one piece every few MFMAs (woven)
//   D     lookahead of the LDS-DMA stream in k-tiles (1 = what a two-stage ring allows; 2, 3 = what a deeper ring would:
//         the probe lets the deeper stages alias the two real ones — nobody checks the products)
//   WARM  L2 warming: every CU touches ITS SHARE of the operand lines its XCD will stream WD k-tiles later (one 4-byte
//         LDS-DMA per line, 1/8 .. 1/4 of its own lines), so that the 1 KiB pieces of all sharers hit L2
// and reports the time of the whole layer (= what its k-loops alone would cost) as TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/feedx_probe.hip -o tools/probes/feedx_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_vptr;

struct FP {
    const f16* A; const f16* W; int M, N, K, lda, ldw, GR, WD;
    float* out; unsigned long long* cyc;
};

// virtual block -> tile (the order gemm_q192 uses): every XCD owns a contiguous run of the tile order, tiles in groups of GR
// tile rows with the column index outer.  Also returns this CU's rank among the CUs that share its X / W panel.
template <int TN>
__device__ __forceinline__ void tile_of(int vb, int ntiles, int tiles_m, int tiles_n, int GR, int& m0, int& n0, int& xs, int& ws) {
    const int xcd = vb & 7, loc = vb >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int group = t / (GR * tiles_n), within = t - group * GR * tiles_n;
    const int first_m = group * GR, gsz = min(GR, tiles_m - first_m);
    m0 = (first_m + within % gsz) * 256;
    n0 = (within / gsz) * TN;
    ws = within % gsz;
    xs = (within / gsz) % min(tiles_n, 32 / GR);
}

template <int TJ, int RD, int MM, int DMA, int D, int WARM, int SYNC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void feedx_kernel(FP p) {
    constexpr int TN = 64 * TJ, WB = TN * 128, XB = 32768, BUF = WB + XB;
    constexpr int PWW = 2 * TJ, PWX = 8, P = PWW + PWX;          // pieces per wave and k-tile
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2 * BUF / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(p.W)[i & 4095];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const int nk = p.K / 64, tiles_m = p.M / 256, tiles_n = p.N / TN, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int S_total = my_tiles * nk;
    const int prow = lane >> 3, pc = lane & 7;
    const int lr = wave * 8 + prow;
    const int vW = lr * p.ldw * 2 + ((pc ^ ((lr >> 1) & 7)) << 4), vX = lr * p.lda * 2 + ((pc ^ ((lr >> 1) & 7)) << 4);
    // DMA stream: next k-tile to issue
    int d_kt = 0, d_ti = 0, d_m0, d_n0, d_xs, d_ws, d_step = 0, sp_w = 0, sp_x = 0;
    tile_of<TN>(blockIdx.x, ntiles, tiles_m, tiles_n, p.GR, d_m0, d_n0, d_xs, d_ws);
#define D_DESCRIBE() { sp_w = d_n0 * p.ldw * 2 + d_kt * 128; sp_x = d_m0 * p.lda * 2 + d_kt * 128; ++d_step; \
    if (++d_kt == nk) { d_kt = 0; ++d_ti; if (d_step < S_total) tile_of<TN>(blockIdx.x + d_ti * G, ntiles, tiles_m, tiles_n, p.GR, d_m0, d_n0, d_xs, d_ws); } }
#define PIECE(q, buf) { \
    if ((q) < PWW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_vptr)(smem + (buf) * BUF + (wave + 4 * (q)) * 1024), 16, vW, sp_w + (q) * 64 * p.ldw, 0, 0); \
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(smem + (buf) * BUF + WB + (wave + 4 * ((q) - PWW)) * 1024), 16, vX, sp_x + ((q) - PWW) * 64 * p.lda, 0, 0); }
    // warm stream: WD k-tiles ahead of the DMA stream
    int w_kt = 0, w_ti = 0, w_m0, w_n0, w_xs, w_ws, w_step = 0;
    tile_of<TN>(blockIdx.x, ntiles, tiles_m, tiles_n, p.GR, w_m0, w_n0, w_xs, w_ws);
    const int nsx = min(tiles_n, 32 / p.GR), xrows = 256 / nsx, wrows = TN / p.GR;
#define WARM_ISSUE() { if (WARM && w_step < S_total) { \
    if (wave < 2) { const int l_ = (wave & 1) * 64 + lane; \
        if (l_ < xrows) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(smem + 2 * BUF + wave * 256), 4, (w_xs * xrows + l_) * p.lda * 2, w_m0 * p.lda * 2 + w_kt * 128, 0, 0); } \
    else { const int l_ = (wave & 1) * 64 + lane; \
        if (l_ < wrows) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_vptr)(smem + 2 * BUF + wave * 256), 4, (w_ws * wrows + l_) * p.ldw * 2, w_n0 * p.ldw * 2 + w_kt * 128, 0, 0); } \
    ++w_step; if (++w_kt == nk) { w_kt = 0; ++w_ti; if (w_step < S_total) tile_of<TN>(blockIdx.x + w_ti * G, ntiles, tiles_m, tiles_n, p.GR, w_m0, w_n0, w_xs, w_ws); } } }

    int* const sync_cnt = reinterpret_cast<int*>(smem + 2 * BUF + 1024);
    if (threadIdx.x == 0) *sync_cnt = 0;
    const int frow = lane & 31, fkey = (frow >> 1) & 7, fhalf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int choff[4];
    for (int ks = 0; ks < 4; ++ks) choff[ks] = (((ks * 2 + fhalf) ^ fkey) << 4);
    const char* const wbase = smem + (wn * 32 * TJ + frow) * 128;
    const char* const xbase = smem + WB + (wm * 128 + frow) * 128;
    f32x16 acc[TJ][4];
    for (int i = 0; i < TJ; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 fw[2][TJ], fx[2][4];
#define RDF(set, buf, ks) { if (RD) { \
    _Pragma("unroll") for (int i_ = 0; i_ < TJ; ++i_) fw[set][i_] = *reinterpret_cast<const f16x8*>(wbase + (buf) * BUF + i_ * 4096 + choff[ks]); \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) fx[set][j_] = *reinterpret_cast<const f16x8*>(xbase + (buf) * BUF + j_ * 4096 + choff[ks]); } }
    // one k-step: TJ * 4 MFMAs with pieces q0 .. q0 + NQ - 1 woven in
#define MMD(set, q0, NQ, buf) { \
    _Pragma("unroll") for (int i_ = 0; i_ < TJ; ++i_) \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { \
        if (MM) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fw[set][i_]), "v"(fx[set][j_])); \
        else if (RD) { asm volatile("" :: "v"(fw[set][i_]), "v"(fx[set][j_])); } \
        const int m_ = i_ * 4 + j_; \
        if ((NQ) > 0 && (m_ + 1) * (NQ) / (TJ * 4) > m_ * (NQ) / (TJ * 4)) { PIECE((q0) + m_ * (NQ) / (TJ * 4), buf) } } }
    for (int i = 0; i < TJ; ++i) fw[0][i] = fw[1][i] = *reinterpret_cast<const f16x8*>(wbase + i * 4096);
    for (int j = 0; j < 4; ++j) fx[0][j] = fx[1][j] = *reinterpret_cast<const f16x8*>(xbase + j * 4096);

    // prologue: the warm stream runs WD ahead of the DMA stream, the DMA stream D ahead of the MFMAs
    if (WARM) for (int i = 0; i < p.WD; ++i) WARM_ISSUE()
    if (DMA) for (int i = 0; i < D; ++i) { WARM_ISSUE() D_DESCRIBE()
#pragma unroll
        for (int q = 0; q < P; ++q) PIECE(q, i & 1) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int Q0 = (P + 3) / 4, Q1 = (P - Q0 + 2) / 3, Q2 = (P - Q0 - Q1 + 1) / 2, Q3 = P - Q0 - Q1 - Q2;   // pieces per k-step
    constexpr int NWAIT = (D - 1) * P + (WARM ? D - 1 + 0 : 0);     // younger VMEM ops than k-tile s+1's last piece
    for (int s = 0; s < S_total; s += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // k-tile s + b lives in stage b; the pieces issued during it belong to k-tile s + b + D
            const int tb = (b + D) & 1;
            if (DMA) { WARM_ISSUE() D_DESCRIBE() }
            if (DMA == 1) {
#pragma unroll
                for (int q = 0; q < P; ++q) PIECE(q, tb)
            }
            RDF(1, b, 1)
            if (DMA == 2) MMD(0, 0, Q0, tb) else MMD(0, 0, 0, tb)
            RDF(0, b, 2)
            if (DMA == 2) MMD(1, Q0, Q1, tb) else MMD(1, 0, 0, tb)
            RDF(1, b, 3)
            if (DMA == 2) MMD(0, Q0 + Q1, Q2, tb) else MMD(0, 0, 0, tb)
            // hand-over: k-tile s+b+1 has landed (this wave's pieces) -> barrier -> its first fragments
            if (DMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT) : "memory");
            if (SYNC == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else if (SYNC == 2) {
                // soft barrier: a monotonic arrival counter in LDS instead of s_barrier (which drains the matrix pipe of a
                // one-wave-per-SIMD kernel: nothing else is there to issue while the wave sits in it)
                if (lane == 0) __hip_atomic_fetch_add(sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int want = 4 * (s + b + 1);
                while (__builtin_amdgcn_readfirstlane(*(volatile int*)sync_cnt) < want) {}
            }
            RDF(0, b ^ 1, 0)
            if (DMA == 2) MMD(1, Q0 + Q1 + Q2, Q3, tb) else MMD(1, 0, 0, tb)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < TJ; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    p.out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) p.cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

static float* g_out; static unsigned long long* g_cyc;

template <int TJ, int RD, int MM, int DMA, int D, int WARM, int SYNC = 0>
void run(const char* name, FP p) {
    constexpr int TN = 64 * TJ, BUF = TN * 128 + 32768, LDS = 2 * BUF + 2048;
    p.out = g_out; p.cyc = g_cyc;
    auto kern = feedx_kernel<TJ, RD, MM, DMA, D, WARM, SYNC>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int nb = std::min(256, (p.M / 256) * (p.N / TN));      // persistent: one workgroup per CU, never more than tiles
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    constexpr int NL = 20;       // back-to-back launches per timing: the clock / power state of a continuous stream, not of one 60 us burst from idle
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), LDS, 0, p);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ts.push_back(ms / NL);
    }
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2] * 1e3, usmin = ts[0] * 1e3;
    const int ntiles = (p.M / 256) * (p.N / TN);
    fflush(stdout);
    const double steps = (double)((ntiles + nb - 1) / nb) * (p.K / 64);
    std::vector<unsigned long long> hc(nb); CK(hipMemcpy(hc.data(), g_cyc, nb * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : hc) avg += (double)v; avg /= nb;
    printf("  %-44s %7.1f us (min %6.1f)  %6.0f TFLOP/s-equiv  %5.0f ticks/k-tile", name, us, usmin, 2.0 * p.M * p.N * p.K / us * 1e-6, avg / steps);
    if (MM) printf("  %4.1f %% busy", 100.0 * TJ * 16 * 32 / (avg / steps));
    if (DMA) printf("  DMA %4.1f B/tick/CU", BUF * 1.0 / (avg / steps));
    printf("  [%.2f GHz]\n", avg / us * 1e-3);
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "fc1,fc2,qkv";
    CK(hipMalloc(&g_out, 1 << 22)); CK(hipMalloc(&g_cyc, 256 * 8));
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"fc1", 16384, 3072, 768}, {"fc2", 16384, 768, 3072}, {"qkv", 16384, 2304, 768}};
    for (const Shape& s : shapes) {
        if (!strstr(which, s.name)) continue;
        f16 *dA, *dW;
        const size_t nA = (size_t)s.M * s.K, nW = (size_t)s.N * s.K;
        std::vector<f16> hA(nA), hW(nW);
        unsigned x = 12345;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((int)(x >> 9) % 2001 - 1000) * 1e-3f; };
        for (auto& v : hA) v = (f16)(rnd() * 1.5f);
        for (auto& v : hW) v = (f16)(rnd() * 0.06f);
        CK(hipMalloc(&dA, nA * 2)); CK(hipMalloc(&dW, nW * 2));
        CK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
        FP p{dA, dW, s.M, s.N, s.K, s.K, s.K, 4, 2, nullptr, nullptr};
        printf("=== %s  M=%d N=%d K=%d   (256 x 192 tiles: wave tile 128 x 96, 56 KiB per k-tile)\n", s.name, s.M, s.N, s.K);
        run<3, 0, 1, 0, 1, 0>("MFMA only", p);
        run<3, 0, 1, 0, 1, 0, 1>("MFMA only, NO barrier", p);
        run<3, 0, 1, 0, 1, 0, 2>("MFMA only, LDS-counter barrier", p);
        run<3, 1, 1, 0, 1, 0, 1>("reads + MFMA, NO barrier", p);
        run<3, 1, 1, 0, 1, 0, 2>("reads + MFMA, LDS-counter barrier", p);
        run<3, 1, 1, 2, 2, 0, 1>("reads + MFMA + DMA woven, lookahead 2, NO barrier", p);
        run<3, 1, 1, 2, 2, 0, 2>("reads + MFMA + DMA woven, lookahead 2, LDS-counter barrier", p);
        run<3, 1, 1, 0, 1, 0>("reads + MFMA (no feed)", p);
        run<3, 0, 0, 1, 1, 0>("DMA only, burst, lookahead 1", p);
        run<3, 0, 0, 1, 2, 0>("DMA only, burst, lookahead 2", p);
        run<3, 0, 0, 1, 3, 0>("DMA only, burst, lookahead 3", p);
        run<3, 0, 0, 1, 2, 1>("DMA only, burst, lookahead 2, L2 warm +2", p);
        run<3, 1, 1, 1, 1, 0>("reads + MFMA + DMA burst,  lookahead 1", p);
        run<3, 1, 1, 2, 1, 0>("reads + MFMA + DMA woven,  lookahead 1", p);
        run<3, 1, 1, 1, 2, 0>("reads + MFMA + DMA burst,  lookahead 2", p);
        run<3, 1, 1, 2, 2, 0>("reads + MFMA + DMA woven,  lookahead 2", p);
        run<3, 1, 1, 2, 3, 0>("reads + MFMA + DMA woven,  lookahead 3", p);
        run<3, 1, 1, 2, 2, 1>("reads + MFMA + DMA woven,  lookahead 2, warm +2", p);
        { FP q = p; q.WD = 4; run<3, 1, 1, 2, 2, 1>("reads + MFMA + DMA woven,  lookahead 2, warm +4", q); }
        { FP q = p; q.GR = 8; run<3, 1, 1, 2, 2, 0>("reads + MFMA + DMA woven,  lookahead 2, GR 8", q); }
        { FP q = p; q.GR = 2; run<3, 1, 1, 2, 2, 0>("reads + MFMA + DMA woven,  lookahead 2, GR 2", q); }
        if (s.N % 256 == 0) {
            printf("--- 256 x 256 tiles: wave tile 128 x 128, 64 KiB per k-tile\n");
            run<4, 1, 1, 0, 1, 0>("reads + MFMA (no feed)", p);
            run<4, 0, 0, 1, 2, 0>("DMA only, burst, lookahead 2", p);
            run<4, 1, 1, 1, 1, 0>("reads + MFMA + DMA burst,  lookahead 1", p);
            run<4, 1, 1, 2, 1, 0>("reads + MFMA + DMA woven,  lookahead 1", p);
            run<4, 1, 1, 2, 2, 0>("reads + MFMA + DMA woven,  lookahead 2", p);
            run<4, 1, 1, 2, 2, 1>("reads + MFMA + DMA woven,  lookahead 2, warm +2", p);
        }
        CK(hipFree(dA)); CK(hipFree(dW));
    }
    return 0;
}

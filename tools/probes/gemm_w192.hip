// Persistent 256(M) x 192(N) x 64 f16 MFMA GEMM, ONE WAVE PER SIMD:  OUT16[M,N] = act(A[M,K] * W[N,K]^T + bias).
//
// EXPERIMENT (not on the default path; launch_gemm variants 60..65).  Result: its k-loop runs at the same ~1000-1100 TFLOP/s
// as gemm_q192's and gemm_pp256's although its MFMA + fragment-read stream alone sustains 99.8 % of the matrix peak —
// all three are held at ~20-24 operand bytes / clk / CU by the HBM/L2 -> LDS delivery of REAL operand data (a hot-data
// probe reaches 42-98).  Deeper lookahead (a four-stage 32-deep ring, measured then removed), L2 warming two and three
// k-tiles ahead (template PF), padded row strides and wave priorities all left that rate unchanged.  Kept because its
// structure (no intra-k-tile barriers, 2.4x fewer LDS bytes per FLOP, 130 spare registers) is the better base once the
// delivery rate is understood.
//
// Successor experiment to gemm_q192.hip.  There, two waves per SIMD alternate barrier-separated read / MFMA segments and
// the read segment (15 ds_read_b128 + LDS-DMA issue + bookkeeping) is longer than the 24-MFMA segment; with 96 + 48 + 56
// registers per wave there is no room to software-pipeline the fragment reads inside a wave.  A probe
// (tools/probes/pipe_probe.hip) showed that ONE wave per SIMD with a 128 x 96 wave tile (192 accumulator registers in the
// AGPR half of the 512-entry file), fragments double-buffered in registers and read one k-step ahead, sustains 99.8 % of
// the matrix peak under plain hipcc scheduling: 12 v_mfma_f32_32x32x16_f16 (384 clk) hide 7 ds_read_b128 per k-step, and
// the wave tile needs 2.4x fewer LDS bytes per FLOP than q192's 48 x 128.
//
// Structure: 4 waves = 2 (M halves) x 2 (N halves); per 64-deep k-tile 4 k-steps of 12 MFMAs; operand k-tiles HBM/L2 -> LDS
// by buffer_load ... lds into 2 x (W 24 KiB | X 32 KiB); ONE workgroup barrier per k-tile, placed in front of the last
// k-step's MFMAs: it publishes k-tile t+1 (every wave has waited for its own DMA pieces) and retires k-tile t's buffer,
// which the DMA of k-tile t+2 then overwrites while the last 12 MFMAs of k-tile t and the whole of k-tile t+1 run.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace srh {

typedef __attribute__((address_space(3))) void* lds_vptr;

constexpr int W_WBUF = 24576, W_XBUF = 32768, W_BUF = W_WBUF + W_XBUF;     // 56 KiB per k-tile
constexpr int W_LDS = 2 * W_BUF;

__device__ __forceinline__ void w_tile_of(int vb, int ntiles, int tiles_m, int tiles_n, int& m0, int& n0) {
    const int xcd = vb & 7, loc = vb >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int group = t / (4 * tiles_n), within = t - group * 4 * tiles_n;
    const int first_m = group * 4, gsz = min(4, tiles_m - first_m);
    m0 = (first_m + within % gsz) * 256;
    n0 = (within / gsz) * 192;
}

// PF: L2 warming.  Measured: without the MFMAs the loop still takes ~2650 clk per k-tile = 21 B/clk/CU, the same rate q192 draws,
// although the LDS-DMA engine sustains 42-98 B/clk/CU on L2-resident data (dma_probe): with one k-tile (56 KiB / CU) in
// flight the loop is bound by the LATENCY of operand lines that come from HBM / Infinity Cache.  LDS cannot hold a third
// k-tile, so the k-tile after the next one is pulled into L2 instead: one 4-byte load per 128-byte operand line (112 lines
// per wave per k-tile = two wave instructions, written to an LDS scratch so that no VGPR is involved), issued two k-tiles
// ahead of the DMA that will then hit L2.
// SPREAD: the 14 DMA pieces of a k-tile are not issued as one burst after the barrier but trickled between the MFMAs of the
// next four k-steps (7 + 4 + 3 + 0).  PMC showed the average L2 read latency is only ~350 clk and the TA busy 28 %, yet a
// wave that issues a burst of buffer_load...lds stalls until the TA has taken every piece of EVERY wave (~20 clk each, 56
// per k-tile): with a burst, all four waves — and with them the matrix pipes — sit idle for ~1000 clk per k-tile.
template <int ABL, int ACT, bool BIAS, bool PF, bool SPREAD>   // ABL: 0 normal, 1 no epilogue stores, 2 no MFMA
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_w192_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int nk = p.K / 64;
    const int tiles_m = p.M / 256, tiles_n = p.N / 192, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int S_total = my_tiles * nk;

    // LDS-DMA lane constants: a wave moves pieces {wave + 4 i} (8 rows x 128 B) of every slab; swizzle on the SOURCE address
    const int prow = lane >> 3, pc = lane & 7;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    int vW[6], vX[8];
#pragma unroll
    for (int i = 0; i < 6; ++i) { const int lr = (wave + 4 * i) * 8 + prow; vW[i] = lr * p.ldw * 2 + ((pc ^ ((lr >> 1) & 7)) << 4); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int lr = (wave + 4 * i) * 8 + prow; vX[i] = lr * p.lda * 2 + ((pc ^ ((lr >> 1) & 7)) << 4); }

    const int vPW = min(wave * 48 + lane, 191) * p.ldw * 2, vPX = (wave * 64 + lane) * p.lda * 2;   // prefetch: one lane per operand row
    // DMA stream state: next k-tile to issue
    int d_kt = 0, d_ti = 0, d_m0, d_n0, d_step = 0;
    w_tile_of(blockIdx.x, ntiles, tiles_m, tiles_n, d_m0, d_n0);
#define W_ISSUE(buf) { \
    const int sw_ = d_n0 * p.ldw * 2 + d_kt * 128, sx_ = d_m0 * p.lda * 2 + d_kt * 128; \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_vptr)(smem + (buf) * W_BUF + (wave + 4 * i_) * 1024), 16, vW[0], sw_ + i_ * 64 * p.ldw, 0, 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(smem + (buf) * W_BUF + W_WBUF + (wave + 4 * i_) * 1024), 16, vX[0], sx_ + i_ * 64 * p.lda, 0, 0); \
    ++d_step; \
    if (++d_kt == nk) { d_kt = 0; ++d_ti; if (d_step < S_total) w_tile_of(blockIdx.x + d_ti * G, ntiles, tiles_m, tiles_n, d_m0, d_n0); } }
    // warm L2 with the k-tile the DMA stream will issue NEXT (d_* already point at it)
#define W_PREFETCH() { \
    const int pk_ = d_kt + 2 < nk ? d_kt + 2 : d_kt;        /* three k-tiles ahead of its DMA (within the tile) */ \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_vptr)(smem + W_LDS + wave * 512), 4, vPW, d_n0 * p.ldw * 2 + pk_ * 128, 0, 0); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(smem + W_LDS + wave * 512 + 256), 4, vPX, d_m0 * p.lda * 2 + pk_ * 128, 0, 0); }

    // fragment addressing
    const int fkey = (frow >> 1) & 7;
    int choff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) choff[ks] = ((ks * 2 + fhalf) ^ fkey) << 4;
    const char* const wbase = smem + (wn * 96 + frow) * 128;
    const char* const xbase = smem + W_WBUF + (wm * 128 + frow) * 128;
    f16x8 fw[2][3], fx[2][4];
#define W_RD(set, buf, ks) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) fw[set][i_] = *reinterpret_cast<const f16x8*>(wbase + (buf) * W_BUF + i_ * 4096 + choff[ks]); \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) fx[set][j_] = *reinterpret_cast<const f16x8*>(xbase + (buf) * W_BUF + j_ * 4096 + choff[ks]); }
#define W_MM(set) { if (ABL != 2) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[i_][j_] = mfma32(fw[set][i_], fx[set][j_], acc[i_][j_]); } }
    // one DMA piece (0..5: W rows, 6..13: X rows) of the k-tile described by (sp_w, sp_x) into buffer sp_buf
    int sp_w = 0, sp_x = 0;
#define W_PIECE(q, buf) { { \
    /* piece i of a slab starts 32 rows below piece i-1 and the swizzle key has period 16 rows: one lane offset serves all */ \
    if ((q) < 6) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_vptr)(smem + (buf) * W_BUF + (wave + 4 * (q)) * 1024), 16, vW[0], sp_w + (q) * 64 * p.ldw, 0, 0); \
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(smem + (buf) * W_BUF + W_WBUF + (wave + 4 * ((q) - 6)) * 1024), 16, vX[0], sp_x + ((q) - 6) * 64 * p.lda, 0, 0); } }
    // 12 MFMAs of one k-step with DMA pieces q0 .. q0+n-1 woven in (one piece after every 12/n-th MFMA).  The MFMAs are
    // volatile asm here: the builtin is a pure node the scheduler slides across the (side-effecting) DMA issues, and pinning
    // its AGPR accumulators with empty asm makes the compiler bounce them through VGPRs.  Pieces are issued unconditionally
    // (past the end of the work list they re-fetch the last k-tile into a retired buffer) so that a k-step stays one block.
#define W_MM_DMA(set, q0, n, buf) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { \
        if (ABL != 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fw[set][i_]), "v"(fx[set][j_])); \
        const int m_ = i_ * 4 + j_; \
        if ((n) > 0 && (m_ + 1) * (n) / 12 > m_ * (n) / 12) { W_PIECE((q0) + m_ * (n) / 12, buf) } } }

    f32x16 acc[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: k-tiles 0 and 1
    W_ISSUE(0)
    if (SPREAD) {
        // k-tile 1: pieces 0..6 now, 7..13 ride on k-tile 0's MFMAs like in the steady state
        sp_w = d_n0 * p.ldw * 2 + d_kt * 128; sp_x = d_m0 * p.lda * 2 + d_kt * 128;
        if (S_total > 1) { ++d_step; if (++d_kt == nk) { d_kt = 0; ++d_ti; if (d_step < S_total) w_tile_of(blockIdx.x + d_ti * G, ntiles, tiles_m, tiles_n, d_m0, d_n0); } }
#pragma unroll
        for (int q = 0; q < 7; ++q) W_PIECE(q, 1)
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else if (S_total > 1) { W_ISSUE(1) if (PF && S_total > 2) { W_PREFETCH() asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    W_RD(0, 0, 0)

    int c_m0, c_n0;
    w_tile_of(blockIdx.x, ntiles, tiles_m, tiles_n, c_m0, c_n0);
    int s_left = S_total;
    for (int ti = 0; ti < my_tiles; ++ti) {
        for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
            for (int b = 0; b < 2; ++b, --s_left) {        // k-tile kt + b lives in buffer b (nk is even)
              if (!SPREAD) {
                W_RD(1, b, 1)
                W_MM(0)
                W_RD(0, b, 2)
                W_MM(1)
                W_RD(1, b, 3)
                W_MM(0)
                // hand-over: k-tile t+1 has landed (own pieces) -> barrier -> its first fragments; buffer b is free -> DMA k-tile t+2
                if (s_left > 1) {
                    if (PF) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // the two prefetch loads issued after k-tile t+1's pieces may stay in flight
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of buffer b have returned
                    __builtin_amdgcn_s_barrier();
                    W_RD(0, b ^ 1, 0)
                    if (s_left > 2) { W_ISSUE(b) if (PF) { if (s_left > 3) W_PREFETCH() else { W_PREFETCH() } } }
                }
                W_MM(1)
              } else {
                // pieces 7..13 of k-tile t+1 (target buffer b ^ 1) ride on k-step 0; k-steps 1 and 2 carry none, so the last piece
                // has two and a half k-steps (~1000 clk) to land before the hand-over waits for it.  (Spreading further — 4 + 3
                // pieces on k-steps 0 and 1 — leaves an L2 miss too little time: measured 25 % slower than the burst.)
                W_RD(1, b, 1)
                W_MM_DMA(0, 7, 7, b ^ 1)
                W_RD(0, b, 2)
                W_MM_DMA(1, 14, 0, b ^ 1)
                W_RD(1, b, 3)
                W_MM_DMA(0, 0, 0, b)
                if (s_left > 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    W_RD(0, b ^ 1, 0)
                    // start k-tile t+2 (into the buffer just retired): describe it, issue its first 7 pieces inside the last k-step
                    if (s_left > 2) {
                        sp_w = d_n0 * p.ldw * 2 + d_kt * 128; sp_x = d_m0 * p.lda * 2 + d_kt * 128;
                        ++d_step;
                        if (++d_kt == nk) { d_kt = 0; ++d_ti; if (d_step < S_total) w_tile_of(blockIdx.x + d_ti * G, ntiles, tiles_m, tiles_n, d_m0, d_n0); }
                    }
                    W_MM_DMA(1, 0, 7, b)
                } else {
                    W_MM_DMA(1, 0, 0, b)
                    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // asm MFMA results -> compiler-scheduled reads
                }
              }
            }
        }
        // tile done: epilogue straight from the accumulators (lane: row m = frow, 4 consecutive columns per register quad)
        if (ABL != 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = c_m0 + wm * 128 + j * 32 + frow;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = c_n0 + wn * 96 + i * 32 + 8 * q + 4 * fhalf;
                        float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        if (BIAS) { const float4 bb = *reinterpret_cast<const float4*>(p.bias + n); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
                        if (ACT == 1) { for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]); }
                        else if (ACT == 2) { for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f); }
                        const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h;
                    }
                }
        }
        if (ABL == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" :: "a"(acc[i][j]));       // keep the MFMAs alive
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (SPREAD) asm volatile("s_nop 7" ::: "memory");
        if (ti + 1 < my_tiles) w_tile_of(blockIdx.x + (ti + 1) * G, ntiles, tiles_m, tiles_n, c_m0, c_n0);
    }
}

template <int ACT, bool BIAS>
static void w192_launch(const GemmParams& p, hipStream_t stream, int grid, int ablation) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<0, ACT, BIAS, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<1, ACT, BIAS, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<2, ACT, BIAS, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<0, ACT, BIAS, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<1, ACT, BIAS, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<2, ACT, BIAS, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<0, ACT, BIAS, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<1, ACT, BIAS, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w192_kernel<2, ACT, BIAS, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS + 2048);
        attr_set = true;
    }
    const int L = W_LDS + 2048;
    if (ablation == 1) hipLaunchKernelGGL((gemm_w192_kernel<1, ACT, BIAS, false, false>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 2) hipLaunchKernelGGL((gemm_w192_kernel<2, ACT, BIAS, false, false>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 3) hipLaunchKernelGGL((gemm_w192_kernel<0, ACT, BIAS, true, false>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 4) hipLaunchKernelGGL((gemm_w192_kernel<1, ACT, BIAS, true, false>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 5) hipLaunchKernelGGL((gemm_w192_kernel<2, ACT, BIAS, true, false>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 6) hipLaunchKernelGGL((gemm_w192_kernel<0, ACT, BIAS, false, true>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 7) hipLaunchKernelGGL((gemm_w192_kernel<1, ACT, BIAS, false, true>), dim3(grid), dim3(256), L, stream, p);
    else if (ablation == 8) hipLaunchKernelGGL((gemm_w192_kernel<2, ACT, BIAS, false, true>), dim3(grid), dim3(256), L, stream, p);
    else hipLaunchKernelGGL((gemm_w192_kernel<0, ACT, BIAS, false, false>), dim3(grid), dim3(256), L, stream, p);
}

int launch_gemm_w192(const GemmParams& p, hipStream_t stream, int ablation) {
    if (!q192_supported(p) || (p.K / 64) % 2) return -2;

    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -3;
        n_cu = prop.multiProcessorCount;
    }
    const int ntiles = (p.M / 256) * (p.N / 192);
    const int grid = ntiles < n_cu ? ntiles : (n_cu / 8) * 8;
    if (p.bias) {
        if (p.act == 1) w192_launch<1, true>(p, stream, grid, ablation);
        else if (p.act == 2) w192_launch<2, true>(p, stream, grid, ablation);
        else w192_launch<0, true>(p, stream, grid, ablation);
    } else {
        if (p.act == 1) w192_launch<1, false>(p, stream, grid, ablation);
        else if (p.act == 2) w192_launch<2, false>(p, stream, grid, ablation);
        else w192_launch<0, false>(p, stream, grid, ablation);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

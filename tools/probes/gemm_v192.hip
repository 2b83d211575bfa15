// EXPERIMENT (round 2, not in libsamroad_hip.so): measured slower than gemm_q192 on every ViT-B layer (profiles/r02_gemm_v192_probe.txt:
// qkv 70.9 vs 64.6 us, fc1 98 vs 98, fc2 86 vs 76) — one wave per SIMD pays ~7 us per layer for the 16-byte deferred stores and ~18 us for the
// GELU fillers, and its GELU variant still has a layout bug (fc1 check fails).  Kept as the record of the attempt.
// Persistent 256(M) x 192(N) x 64 f16 MFMA GEMM, ONE WAVE PER SIMD, with a DEFERRED register-held epilogue:
//     OUT16[M,N] = act(A[M,K] * W[N,K]^T + bias)          (fp16 out, fp32 accumulate)
// for the big linear layers of the SAM ViT blocks (qkv, proj, fc1, fc2 — SURVEY.md §2.1 K4/K7/K8; reference model.py:245-258).
//
// Why (measured on MI355X, profiles/r01b_gemm_w192_prefetch_probe.txt, r01c_gemm_w192_spread.txt): with one 128 x 96 wave tile
// per SIMD (192 accumulators in the AGPR half of the 512-entry register file, fragments double-buffered in VGPRs and read one
// k16 step ahead) the k-loop ALONE runs fc1 in 69 us against 83 us for the two-waves-per-SIMD ping-pong kernel
// (gemm_q192.hip), but an epilogue executed after the k-loop doubled the kernel: fc1 writes 100 MB of fp16 through 256
// store-issue-bound CUs.  gemm_q192's answer — convert the finished tile to packed fp16 registers and finish / store it in
// small steps riding on the next tile's k-loop — costs that kernel 0.25-0.5 us per k-tile because its two waves per SIMD
// have no register room to pipeline and every epilogue step lengthens a barrier-paired segment.  Here the same idea lives in
// a wave that owns its SIMD: a k16 step is 12 v_mfma_f32_32x32x16_f16 (384 clk) whose issue gaps hide ~5 single-issue
// instructions each (MI355X_MICROARCH.md constants table), so 7 fragment ds_read_b128, one epilogue group (4 outputs: GELU
// math + packing) and every second step one 16-byte-per-lane store disappear between the MFMAs.
//
//   * bias is the C OPERAND of a tile's first twelve MFMAs (three 16-register AGPR tuples in accumulator layout, read from
//     the tile's LDS bias slot when the previous tile is packed): no accumulator initialisation code, the deferred data is
//     rounded to fp16 once, after the bias — and layers without activation defer nothing but stores;
//   * packing: v_cvt_pk + v_permlane32_swap (guide T21): the 32x32 MFMA leaves a row's 8-column groups split across the
//     half-waves; after the swap every lane holds 8 consecutive columns = one 16-byte store, 24 stores per wave and tile;
//   * operands HBM/L2 -> LDS by buffer_load ... lds (1 KiB pieces, XOR swizzle on the SOURCE address), 2 x 56 KiB stages,
//     ONE workgroup barrier per k-tile placed in front of the last k16 step's MFMAs (it publishes k-tile t+1 — every wave
//     has waited for its own pieces — and retires k-tile t's buffer, which the DMA of k-tile t+2 then overwrites);
//   * vmcnt counts loads and stores in issue order on gfx9-family parts (see gemm_q192.hip header): the hand-over wait is
//     vmcnt(1) while an epilogue is riding along (one store of this k-tile is younger than the awaited pieces; from the
//     second k-tile on the previous hand-over's tail store is too, but it is a whole k-tile old by then).
//
// Reference semantics: nn.Linear (+ exact-erf nn.GELU for mlp.lin1) of the fork's ImageEncoderViT blocks, SURVEY App. B.2-3.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace srh {

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int V_WBUF = 24576, V_XBUF = 32768, V_BUF = V_WBUF + V_XBUF;     // 56 KiB per k-tile
constexpr int V_BIAS = 2 * V_BUF;                                          // 2 slots x 4 waves x 1 KiB
constexpr int V_LDS = V_BIAS + 2 * 4096;                                   // 120 KiB

// virtual block vb (runs on XCD vb % 8; speed only) -> tile origin: every XCD owns a contiguous run of the tile order, tiles
// ordered in groups of 4 tile rows with the column index outer (shared A / W panels stay in its L2)
__device__ __forceinline__ void v_tile_of(int vb, int ntiles, int tiles_m, int tiles_n, int& m0, int& n0) {
    const int xcd = vb & 7, loc = vb >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int group = t / (4 * tiles_n), within = t - group * 4 * tiles_n;
    const int first_m = group * 4, gsz = min(4, tiles_m - first_m);
    m0 = (first_m + within % gsz) * 256;
    n0 = (within / gsz) * 192;
}

template <int ACT>
__device__ __forceinline__ float v_act(float x) {          // ACT 1: exact-erf GELU (common.hpp gelu_fast), 2: ReLU
    return ACT == 1 ? gelu_fast(x) : (ACT == 2 ? fmaxf(x, 0.f) : x);
}

template <int ABL, int ACT, bool BIAS>   // ABL ablation aid: 0 normal, 1 no epilogue stores, 2 no epilogue at all (k-loop only)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_v192_kernel(GemmParams p) {
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

    // ---- LDS-DMA lane constants: a wave moves pieces {wave + 4 i} (8 rows x 128 B) of every slab; piece i of a slab starts 32
    // rows below piece i-1 and the swizzle key has period 16 rows, so one lane offset per operand serves all pieces
    const int prow = lane >> 3, pc = lane & 7;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)p.out_f16, 0, 0x7fffffff, 0x00020000);
    const int lr0 = wave * 8 + prow;
    const int vW = lr0 * p.ldw * 2 + ((pc ^ ((lr0 >> 1) & 7)) << 4);
    const int vX = lr0 * p.lda * 2 + ((pc ^ ((lr0 >> 1) & 7)) << 4);
    const int vB = min(lane, 47) * 16;                            // 192 bias floats = 48 lanes x 16 B (per-wave copy)

    // DMA stream state: the next k-tile to issue
    int d_kt = 0, d_ti = 0, d_m0, d_n0, d_step = 0, d_slot = 0;
    v_tile_of(blockIdx.x, ntiles, tiles_m, tiles_n, d_m0, d_n0);
#define V_ISSUE(buf) { \
    if (BIAS && d_kt == 0) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr)(smem + V_BIAS + d_slot * 4096 + wave * 1024), 16, vB, d_n0 * 4, 0, 0); \
    const int sw_ = d_n0 * p.ldw * 2 + d_kt * 128, sx_ = d_m0 * p.lda * 2 + d_kt * 128; \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_vptr)(smem + (buf) * V_BUF + (wave + 4 * i_) * 1024), 16, vW, sw_ + i_ * 64 * p.ldw, 0, 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(smem + (buf) * V_BUF + V_WBUF + (wave + 4 * i_) * 1024), 16, vX, sx_ + i_ * 64 * p.lda, 0, 0); \
    ++d_step; \
    if (++d_kt == nk) { d_kt = 0; ++d_ti; d_slot ^= 1; \
        if (d_step < S_total) v_tile_of(blockIdx.x + d_ti * G, ntiles, tiles_m, tiles_n, d_m0, d_n0); } }

    // ---- fragment addressing (32x32x16 operand: row = lane & 31, k = (lane >> 5) * 8 + j; 16-byte chunk XOR swizzle)
    const int fkey = (frow >> 1) & 7;
    int choff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) choff[ks] = ((ks * 2 + fhalf) ^ fkey) << 4;
    const char* const wbase = smem + (wn * 96 + frow) * 128;
    const char* const xbase = smem + V_WBUF + (wm * 128 + frow) * 128;
    f16x8 fw[2][3], fx[2][4];
#define V_RD(set, buf, ks) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) fw[set][i_] = *reinterpret_cast<const f16x8*>(wbase + (buf) * V_BUF + i_ * 4096 + choff[ks]); \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) fx[set][j_] = *reinterpret_cast<const f16x8*>(xbase + (buf) * V_BUF + j_ * 4096 + choff[ks]); }
    // MFMA m = 4 i + j of a k16 step as volatile asm: the accumulators stay IN PLACE in the AGPR half of the register file (the
    // builtin lets the allocator rotate D != C through all 256 AGPRs) and, volatile statements keeping their order, the epilogue
    // VALU work pinned between them with empty asm operands (§5.7 item 3 of the HIP guide) really issues in the MFMA gaps —
    // left to itself hipcc emits [all VALU][12 MFMAs] and the matrix pipe idles during the VALU block (measured: fc1 +53 us).
    // The FIRST k16 step of a tile takes the tile's bias (one 16-register tuple per 32-column block i, identical for the four
    // row blocks j) as the C operand and defines the accumulators afresh: no accumulator initialisation code at all.
#define V_MFMA(set, m, FIRST) { \
    if (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=a"(acc[(m) >> 2][(m) & 3]) : "v"(fw[set][(m) >> 2]), "v"(fx[set][(m) & 3]), "a"(biasT[(m) >> 2])); \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[(m) >> 2][(m) & 3]) : "v"(fw[set][(m) >> 2]), "v"(fx[set][(m) & 3])); }
#define V_PIN(x) asm volatile("" : "+v"(x));
    // one k16 step: 12 MFMAs; when TR, epilogue group G of the previous tile is finished between them (value k after MFMA
    // 3k+1), packed after MFMA 10, and every second group's 16-byte store issued after MFMA 11
#define V_STEP(set, G, TR, FIRST) { \
    const int PR_ = ((TR) ? (G) : 0) >> 1, C0_ = (((TR) ? (G) : 0) & 1) * 2; \
    int ea_ = o4[PR_][C0_], eb_ = o4[PR_][C0_ + 1]; float e0_ = 0.f, e1_ = 0.f, e2_ = 0.f, e3_ = 0.f; \
    V_MFMA(set, 0, FIRST) V_MFMA(set, 1, FIRST) \
    if ((TR) && ACT != 0 && ABL != 2) { V_PIN(ea_) e0_ = v_act<ACT>((float)__builtin_bit_cast(f16x2, ea_)[0]); V_PIN(e0_) } \
    V_MFMA(set, 2, FIRST) V_MFMA(set, 3, FIRST) V_MFMA(set, 4, FIRST) \
    if ((TR) && ACT != 0 && ABL != 2) { V_PIN(ea_) e1_ = v_act<ACT>((float)__builtin_bit_cast(f16x2, ea_)[1]); V_PIN(e1_) } \
    V_MFMA(set, 5, FIRST) V_MFMA(set, 6, FIRST) V_MFMA(set, 7, FIRST) \
    if ((TR) && ACT != 0 && ABL != 2) { V_PIN(eb_) e2_ = v_act<ACT>((float)__builtin_bit_cast(f16x2, eb_)[0]); V_PIN(e2_) } \
    V_MFMA(set, 8, FIRST) V_MFMA(set, 9, FIRST) V_MFMA(set, 10, FIRST) \
    if ((TR) && ACT != 0 && ABL != 2) { V_PIN(eb_) e3_ = v_act<ACT>((float)__builtin_bit_cast(f16x2, eb_)[1]); \
        const f16x2 lo_ = {(f16)e0_, (f16)e1_}, hi_ = {(f16)e2_, (f16)e3_}; \
        o4[PR_][C0_] = __builtin_bit_cast(int, lo_); o4[PR_][C0_ + 1] = __builtin_bit_cast(int, hi_); } \
    V_MFMA(set, 11, FIRST) \
    if ((TR) && ((G) & 1) && ABL == 0) { \
        V_PIN(o4[PR_]) \
        int sb_ = p_soff; \
        asm volatile("" : "+s"(sb_)); \
        __builtin_amdgcn_raw_buffer_store_b128(o4[PR_], rsO, vO, sb_ + (((G) >> 2) & 3) * ld32 + (((G) >> 4) * 32 + (((G) & 3) >> 1) * 16) * 2, 0); } }

    // accumulators: acc[i][j][4 q + e] = output (row m = wm*128 + j*32 + frow, column n = wn*96 + i*32 + 8 q + 4 fhalf + e)
    f32x16 acc[3][4];
    // bias of the tile whose k-loop starts next (its LDS slot was staged with the tile's first k-tile), in accumulator layout
    f32x16 biasT[3];
#define V_BIAS_LOAD(slot) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { \
        f32x4 b4_ = {0.f, 0.f, 0.f, 0.f}; \
        if (BIAS) b4_ = *reinterpret_cast<const f32x4*>((const char*)__builtin_assume_aligned( \
            smem + V_BIAS + (slot) * 4096 + wave * 1024 + (wn * 96 + i_ * 32 + q_ * 8 + fhalf * 4) * 4, 16)); \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) biasT[i_][q_ * 4 + e_] = b4_[e_]; } }

    // ---- deferred epilogue state: group g = (i*4 + j)*4 + q holds 4 outputs as two packed registers; after the
    // v_permlane32_swap of V_PACK a PAIR of groups (q even, q+1) is 16 contiguous bytes of one output row:
    // lanes 0-31 columns i*32 + 8q .. +7, lanes 32-63 columns i*32 + 8q + 8 .. +15
    v4i o4[24];                                                   // pair pr = g >> 1: components 2 (g & 1), 2 (g & 1) + 1 = group g
#pragma unroll
    for (int pr = 0; pr < 24; ++pr) o4[pr] = v4i{0, 0, 0, 0};
    const int vO = ((wm * 128 + frow) * p.ldc16 + wn * 96 + fhalf * 8) * 2;
    const int ld32 = p.ldc16 * 64;                                // bytes per 32 output rows
    int p_soff = 0;                                               // finished tile: byte offset of its origin in OUT16
#define V_PACK() { \
    _Pragma("unroll") for (int ij_ = 0; ij_ < 12; ++ij_) { \
        const f32x16& a_ = acc[ij_ >> 2][ij_ & 3]; \
        int h_[8]; \
        _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) { \
            const f16x2 t_ = {(f16)a_[c_ * 2], (f16)a_[c_ * 2 + 1]}; \
            h_[c_] = __builtin_bit_cast(int, t_); } \
        _Pragma("unroll") for (int qp_ = 0; qp_ < 2; ++qp_) { \
            /* groups q = 2 qp (registers h[4qp], h[4qp+1]) and q + 1 (h[4qp+2], h[4qp+3]) */ \
            const auto r0_ = __builtin_amdgcn_permlane32_swap(h_[4 * qp_], h_[4 * qp_ + 2], false, false); \
            const auto r1_ = __builtin_amdgcn_permlane32_swap(h_[4 * qp_ + 1], h_[4 * qp_ + 3], false, false); \
            o4[ij_ * 2 + qp_] = v4i{(int)r0_[0], (int)r1_[0], (int)r0_[1], (int)r1_[1]}; } \
        __builtin_amdgcn_sched_barrier(0); } }

    // ---- prologue: k-tiles 0 and 1 of the first tile
    V_ISSUE(0)
    if (S_total > 1) { V_ISSUE(1) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    V_RD(0, 0, 0)
    V_BIAS_LOAD(0)

    int c_m0, c_n0, c_slot = 0;
    v_tile_of(blockIdx.x, ntiles, tiles_m, tiles_n, c_m0, c_n0);
    int s_left = S_total;
    bool have_prev = false;
    // one k-tile (buffer b, epilogue groups E0 .. E0+3 of the previous tile riding along when TR)
#define V_KTILE(b, E0, TR, FIRST) { \
    V_RD(1, b, 1) \
    V_STEP(0, (E0), TR, FIRST) \
    V_RD(0, b, 2) \
    V_STEP(1, (E0) + 1, TR, false) \
    V_RD(1, b, 3) \
    V_STEP(0, (E0) + 2, TR, false) \
    if (s_left > 1) { \
        /* hand-over: k-tile t+1 has landed (own pieces; this k-tile's epilogue store of group E0+1 is younger: the one of */ \
        /* group E0+3 is issued after the wait) -> barrier -> its first fragments; buffer b is free -> DMA k-tile t+2 */ \
        if ((TR) && ABL == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
        V_RD(0, (b) ^ 1, 0) \
        if (s_left > 2) V_ISSUE(b) \
    } \
    V_STEP(1, (E0) + 3, TR, false) \
    --s_left; }

    for (int ti = 0; ti < my_tiles; ++ti) {
        for (int kb = 0; kb < nk; kb += 12) {
            if (kb == 0 && have_prev) {
#pragma unroll
                for (int kk = 0; kk < 12; ++kk) V_KTILE(kk & 1, 4 * kk, true, kk == 0)
            } else if (kb == 0) {
#pragma unroll
                for (int kk = 0; kk < 12; ++kk) V_KTILE(kk & 1, 0, false, kk == 0)
            } else {
#pragma unroll
                for (int kk = 0; kk < 12; ++kk) V_KTILE(kk & 1, 0, false, false)
            }
        }
        // tile done: hand the accumulators to the deferred epilogue (fp16, bias included, store lane layout) and go on.
        // The asm MFMAs are invisible to hipcc's hazard recognizer: 16 wait states before its v_accvgpr_read of their results
        asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
        V_PACK()
        have_prev = true;
        p_soff = (c_m0 * p.ldc16 + c_n0) * 2;
        c_slot ^= 1;
        if (ti + 1 < my_tiles) {
            v_tile_of(blockIdx.x + (ti + 1) * G, ntiles, tiles_m, tiles_n, c_m0, c_n0);
            V_BIAS_LOAD(c_slot)
        }
    }
    // last tile: nothing left to hide its epilogue behind
    if (ABL != 2) {
#pragma unroll
        for (int pr = 0; pr < 24; ++pr) {
            if (ACT != 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x2 a_ = __builtin_bit_cast(f16x2, o4[pr][c]);
                    const f16x2 r_ = {(f16)v_act<ACT>((float)a_[0]), (f16)v_act<ACT>((float)a_[1])};
                    o4[pr][c] = __builtin_bit_cast(int, r_);
                }
            }
            const int g = 2 * pr + 1;
            if (ABL == 0)
                __builtin_amdgcn_raw_buffer_store_b128(o4[pr], rsO, vO, p_soff + ((g >> 2) & 3) * ld32 + ((g >> 4) * 32 + ((g & 3) >> 1) * 16) * 2, 0);
        }
    }
    if (ABL != 0) {
#pragma unroll
        for (int pr = 0; pr < 24; ++pr) asm volatile("" :: "v"(o4[pr]));
    }
}

bool v192_supported(const GemmParams& p) {
    return q192_supported(p);                    // same shapes: M % 256, N % 192, K % 768, fp16 output only
}

template <int ACT, bool BIAS>
static void v192_launch(const GemmParams& p, hipStream_t stream, int grid, int ablation) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v192_kernel<0, ACT, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, V_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v192_kernel<1, ACT, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, V_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v192_kernel<2, ACT, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, V_LDS);
        attr_set = true;
    }
    if (ablation == 1) hipLaunchKernelGGL((gemm_v192_kernel<1, ACT, BIAS>), dim3(grid), dim3(256), V_LDS, stream, p);
    else if (ablation == 2) hipLaunchKernelGGL((gemm_v192_kernel<2, ACT, BIAS>), dim3(grid), dim3(256), V_LDS, stream, p);
    else hipLaunchKernelGGL((gemm_v192_kernel<0, ACT, BIAS>), dim3(grid), dim3(256), V_LDS, stream, p);
}

int launch_gemm_v192(const GemmParams& p, hipStream_t stream, int ablation) {
    if (!v192_supported(p)) return -2;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -3;
        n_cu = prop.multiProcessorCount;
    }
    const int ntiles = (p.M / 256) * (p.N / 192);
    const int grid = ntiles < n_cu ? ntiles : (n_cu / 8) * 8;    // persistent: one workgroup per CU
    if (p.bias) {
        if (p.act == 1) v192_launch<1, true>(p, stream, grid, ablation);
        else if (p.act == 2) v192_launch<2, true>(p, stream, grid, ablation);
        else v192_launch<0, true>(p, stream, grid, ablation);
    } else {
        if (p.act == 1) v192_launch<1, false>(p, stream, grid, ablation);
        else if (p.act == 2) v192_launch<2, false>(p, stream, grid, ablation);
        else v192_launch<0, false>(p, stream, grid, ablation);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

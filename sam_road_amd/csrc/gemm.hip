// f16 x f16 -> f32 MFMA GEMMs with fused epilogues for the linear layers z192 (gemm_z192.hip) does not take:
//     OUT[M,N] = act(A[M,K] * W[N,K]^T + bias) (+ residual | + pos-embed), fp32 and / or fp16 out.
//
// Replaces the ATen call sites K2 / K4 / K7 / K8 / K9 / K13 of SURVEY.md §2.1 (nn.Linear, the neck's 1x1 and 3x3 Conv2d, the TopoNet
// projections; reference model.py:245-258 through the SAM fork, :283-285, :88-117) where no generated body applies: the patch embedding
// of ViT-L / H, every block GEMM of the small-M models (ViT-L / ViT-H at 256 px), the neck, feature_proj, the SAM-decoder branch.
//
// Kernels (all LDS-DMA fed unless noted; launch_gemm at the end of the file is the whole dispatch):
//   gemm_glds256_kernel   256 x 256 tiles, 8 waves                      big layers without a z192 body, half a round of the chip or more
//   gemm_glds_kernel      128 x 128 tiles, two stages, 2 workgroups/CU   everything small; deterministic split-K (+ splitk_reduce_kernel)
//   gemm_ring_kernel      128 x 128 tiles, three-stage ring              <= 256 tiles with K >= 768 (ViT-H proj, the neck's 1x1); <3, 1>: the
//                                                                        implicit 3 x 3 conv with tap-following A pieces (the neck)
//   gemm_pp_kernel        128 x 256 tiles, 8 waves, ping-pong k-loop     small-M layers in one round (ViT-H qkv) and fc2's split-K slices
//   gemm_r320_kernel      128 x 320 tiles, 8 waves                       ViT-H fc1
//   gemm_glds160_kernel   128 x 160 tiles                                small-M layers whose 128 x 128 tile count misses one round
// The MFMA is issued "transposed" (A operand = weight rows, B operand = activation rows) so that every lane ends up with 4 CONSECUTIVE
// output columns of one output row: the epilogue does 16-byte bias / residual loads and 8 / 16-byte stores.  LDS images carry a
// 16-byte-chunk XOR swizzle (applied on the SOURCE side of an LDS-DMA piece) so the ds_read_b128 fragment reads are conflict-free.
// Probe builds (-DSRH_TUNING -Itools/probes) add the probe-only kernels (the register-staged round-1 kernels among them) and the
// by-number dispatcher of tools/probes/gemm_tuning.inc.
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include <utility>

namespace srh {

constexpr int BK = 64;


// Epilogue shared by the GEMM kernels: lane holds, per (i,j,q), 4 consecutive columns n..n+3 of row m.
// fp16 output without an f32 output: the register quads q = 2 k and 2 k + 1 of a row are exchanged between the two half-waves with one
// v_permlane32_swap per packed register pair (half 0 ends up with columns 16 k .. 16 k + 7 of its row, half 1 with 16 k + 8 .. 16 k + 15),
// so a lane stores 16 bytes instead of twice 8: a store instruction touches 32 rows either way and costs the CU's address path the same
// ~87 ticks (profiles/r04_store_probe.txt) — half as many of them.
template <int TN, int TM>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x16 (&acc)[TN][TM], int mw, int nw, int lane) {
    const int frow = lane & 31, fhalf = lane >> 5;
    const bool f16_only = p.out_f16 && !p.out_f32;          // kernel-uniform
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = mw + j * 32 + frow;
        const int mr = min(m, p.M - 1);                      // rows past M compute on a valid row (the exchange below needs every lane) and store nothing
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            float v[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nw + i * 32 + 8 * q + 4 * fhalf;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q][e] = acc[i][j][4 * q + e];
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                    v[q][0] += b.x; v[q][1] += b.y; v[q][2] += b.z; v[q][3] += b.w;
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = gelu_fast(v[q][e]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = fmaxf(v[q][e], 0.f);
                }
                if (p.pos) {
                    const float4 b = *reinterpret_cast<const float4*>(p.pos + (size_t)(mr % p.pos_rows) * p.N + n);
                    v[q][0] += b.x; v[q][1] += b.y; v[q][2] += b.z; v[q][3] += b.w;
                }
                if (p.resid) {
                    const float4 b = *reinterpret_cast<const float4*>(p.resid + (size_t)mr * p.ldr + n);
                    v[q][0] += b.x; v[q][1] += b.y; v[q][2] += b.z; v[q][3] += b.w;
                }
                if (p.out_f32 && m < p.M)
                    *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc + n) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
                if (p.out_f16 && !f16_only && m < p.M) {
                    f16x4 h = {(f16)v[q][0], (f16)v[q][1], (f16)v[q][2], (f16)v[q][3]};
                    *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h;
                }
            }
            if (f16_only) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f16x2 a0h = {(f16)v[2 * k][0], (f16)v[2 * k][1]}, a1h = {(f16)v[2 * k][2], (f16)v[2 * k][3]};
                    const f16x2 b0h = {(f16)v[2 * k + 1][0], (f16)v[2 * k + 1][1]}, b1h = {(f16)v[2 * k + 1][2], (f16)v[2 * k + 1][3]};
                    // swap(x, y): lanes 32..63 of x <-> lanes 0..31 of y; x = quad 2 k, y = quad 2 k + 1
                    const auto r0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0h), __builtin_bit_cast(unsigned, b0h), false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a1h), __builtin_bit_cast(unsigned, b1h), false, false);
                    // half 0: (own quad 2 k | half 1's quad 2 k) = columns 16 k + 0..7; half 1: (half 0's quad 2 k + 1 | own quad 2 k + 1) = 16 k + 8..15
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ w = {r0[0], r1[0], r0[1], r1[1]};
                    if (m < p.M)
                        *reinterpret_cast<u32x4_*>(p.out_f16 + (size_t)m * p.ldc16 + nw + i * 32 + 16 * k + 8 * fhalf) = w;
                }
            }
        }
    }
}

// LDS-staged epilogue for a 64x64 wave tile (acc[2][2]): the accumulators (lane = one row, 4 consecutive
// columns) are transposed through a wave-private 16 KiB f32 region [64 rows][16 chunks of 16 B], chunk index
// XOR-swizzled with the row so both the b128 writes (8-lane groups = 8 rows, same column) and the b128
// reads (16 lanes = one row) are conflict-free.  On read-back 16 consecutive lanes own one 256-byte row
// segment, so bias / pos / residual loads and the f32 / f16 stores are full-line coalesced (the direct
// form wrote 32 different rows per instruction, 16 B each).
template <int TM, int J0>
__device__ __forceinline__ void epilogue_staged(const GemmParams& p, f32x16 (&acc)[2][TM], char* wbuf,
                                                int mw, int nw, int lane) {
    const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 32 + frow;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = i * 8 + 2 * q + fhalf;          // 16-byte chunk (4 columns) inside the 64-col row
                const f32x4 v = {acc[i][J0 + j][4 * q], acc[i][J0 + j][4 * q + 1], acc[i][J0 + j][4 * q + 2], acc[i][J0 + j][4 * q + 3]};
                *reinterpret_cast<f32x4*>(wbuf + r * 256 + ((c ^ (r & 15)) << 4)) = v;
            }
    }
    __builtin_amdgcn_wave_barrier();
    // read-back: 8 lanes own one 64-column row segment (2 chunks = 8 columns each), 8 rows per pass
    const int c2 = (lane & 7) * 2, rsub = lane >> 3;
    const int n = nw + c2 * 4;
    float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
    if (p.bias) { bias0 = *reinterpret_cast<const float4*>(p.bias + n); bias1 = *reinterpret_cast<const float4*>(p.bias + n + 4); }
#pragma unroll 4
    for (int pass = 0; pass < 8; ++pass) {
        const int r = pass * 8 + rsub;
        const int m = mw + r;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(wbuf + r * 256 + ((c2 ^ (r & 15)) << 4));
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(wbuf + r * 256 + (((c2 + 1) ^ (r & 15)) << 4));
        if (m >= p.M) continue;
        float v[8] = {t0[0] + bias0.x, t0[1] + bias0.y, t0[2] + bias0.z, t0[3] + bias0.w,
                      t1[0] + bias1.x, t1[1] + bias1.y, t1[2] + bias1.z, t1[3] + bias1.w};
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
        } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.pos) {
            const float* pp = p.pos + (size_t)(m % p.pos_rows) * p.N + n;
            const float4 b0 = *reinterpret_cast<const float4*>(pp), b1 = *reinterpret_cast<const float4*>(pp + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.resid) {
            const float* pr = p.resid + (size_t)m * p.ldr + n;
            const float4 b0 = *reinterpret_cast<const float4*>(pr), b1 = *reinterpret_cast<const float4*>(pr + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.out_f32) {
            float* po = p.out_f32 + (size_t)m * p.ldc + n;
            *reinterpret_cast<float4*>(po) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(po + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (p.out_f16) {
            f16x8 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3], (f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
            *reinterpret_cast<f16x8*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h;
        }
    }
}

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed dispatch, speed only): give each XCD a
// contiguous run of tiles, ordered in groups of GM tile rows with the column index outer, so the
// workgroups resident on one XCD share a few A panels and W panels that fit its 4 MiB L2.
template <int GM>
__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    const int nb = gridDim.x, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int q = nb >> 3, r = nb & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int group = t / (GM * tiles_n), within = t - group * GM * tiles_n;
    const int first_m = group * GM, gsz = min(GM, tiles_m - first_m);
    tile_m = first_m + within % gsz;
    tile_n = within / gsz;
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant (plain row-major A): 128x128x64 tile, 4 waves, operands go HBM/L2 -> LDS directly with
// global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass).  The DMA destination is
// wave-uniform base + lane*16, i.e. linear; the XOR swizzle the fragment reads need is applied to the
// per-lane SOURCE address instead (lane (row, physical chunk pc) fetches logical chunk pc ^ key(row)).
// Double-buffered: the DMA of k-tile t+1 is in flight while the MFMAs of k-tile t run; one barrier per
// k-tile (whose wait also retires this wave's DMA).
// ---------------------------------------------------------------------------------------------------
template <int ABL, int NST>   // ABL ablation aid: 0 normal, 1 no DMA inside the k-loop, 2 no fragment reads / MFMA
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NST == 1 ? 4 : 2, NST == 1 ? 4 : 2)))
void gemm_glds_kernel(GemmParams p) {
    constexpr int BM = 128, BN = 128, TILE = BM * BK * 2, STAGE = 2 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int tile_m, tile_n;
    tile_of_block<8>((p.M + BM - 1) / BM, p.N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // split-K: workgroup z = blockIdx.y of splitk handles k-tiles [kt0, kt1)
    const int nsplit = p.splitk > 1 ? p.splitk : 1, zsplit = blockIdx.y;
    const int kt0 = zsplit * (p.K / BK) / nsplit, nk = (zsplit + 1) * (p.K / BK) / nsplit;

    // DMA piece i (0..3) of a wave covers rows i*32 + wave*8 .. +8 of the tile, 8 rows x 128 B = 1 KiB
    const int prow = lane >> 3, pc = lane & 7;
    const char* asrc[4];
    const char* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 32 + wave * 8 + prow;
        const int c = pc ^ ((r >> 1) & 7);
        asrc[i] = reinterpret_cast<const char*>(p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + c * 8);
        wsrc[i] = reinterpret_cast<const char*>(p.W + (size_t)(n0 + r) * p.ldw + c * 8);
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
#define SRH_DMA_TILE(kt, stage) { \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
        char* d_ = smem + (stage) * STAGE + (i * 32 + wave * 8) * 128; \
        __builtin_amdgcn_global_load_lds((glb_ptr)(asrc[i] + (size_t)(kt) * (BK * 2)), (lds_ptr)d_, 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((glb_ptr)(wsrc[i] + (size_t)(kt) * (BK * 2)), (lds_ptr)(d_ + TILE), 16, 0, 0); } }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((ks * 2 + fhalf) ^ fkey) << 4);
    const int w_row0 = (wn * 64) * 128, x_row0 = (wm * 64) * 128;

    if (NST == 2) SRH_DMA_TILE(kt0, 0)
    for (int kt = kt0; kt < nk; ++kt) {
        const int stage = NST == 2 ? ((kt - kt0) & 1) : 0;
        if (NST == 1) {
            if (kt > kt0) __syncthreads();           // everyone finished reading the single stage
            SRH_DMA_TILE(kt, 0)
            __syncthreads();                       // vmcnt(0) + barrier: the tile has landed
        } else {
            __syncthreads();   // retires this wave's DMA (vmcnt(0)) and orders everyone's; frees stage^1
            if (ABL != 1 && kt + 1 < nk) SRH_DMA_TILE(kt + 1, stage ^ 1)
        }
        if (ABL == 2) continue;
        const char* sa = smem + stage * STAGE + x_row0;
        const char* sw = smem + stage * STAGE + TILE + w_row0;
        f16x8 fwA[2], fxA[2], fwB[2], fxB[2];
#define SRH_FRAG2(fw, fx, ks) { fw[0] = *reinterpret_cast<const f16x8*>(sw + foff[ks]); \
        fw[1] = *reinterpret_cast<const f16x8*>(sw + foff[ks] + 4096); \
        fx[0] = *reinterpret_cast<const f16x8*>(sa + foff[ks]); \
        fx[1] = *reinterpret_cast<const f16x8*>(sa + foff[ks] + 4096); }
#define SRH_MMA2(fw, fx) { acc[0][0] = mfma32(fw[0], fx[0], acc[0][0]); acc[0][1] = mfma32(fw[0], fx[1], acc[0][1]); \
        acc[1][0] = mfma32(fw[1], fx[0], acc[1][0]); acc[1][1] = mfma32(fw[1], fx[1], acc[1][1]); }
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG2(fwA, fxA, 0)
        SRH_FRAG2(fwB, fxB, 1)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA2(fwA, fxA)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG2(fwA, fxA, 2)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA2(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG2(fwB, fxB, 3)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA2(fwA, fxA)
        SRH_MMA2(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NST == 1) { epilogue<2, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane); return; }
    __syncthreads();   // every wave is done reading operand tiles: LDS becomes epilogue staging space
    if (nsplit > 1) {  // raw f32 partial sums; bias / residual / activation are applied by splitk_reduce_kernel
        GemmParams q = p;
        q.bias = nullptr; q.resid = nullptr; q.pos = nullptr; q.act = 0; q.out_f16 = nullptr;
        q.out_f32 = p.split_ws + (size_t)zsplit * p.M * p.N; q.ldc = p.N;
        epilogue_staged<2, 0>(q, acc, smem + wave * 16384, m0 + wm * 64, n0 + wn * 64, lane);
        return;
    }
    epilogue_staged<2, 0>(p, acc, smem + wave * 16384, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------
// Small-M layers (ViT-H / ViT-L at 256 px: M = 2048 rows, 56 - 480 tiles of 128 x 128): the two-stage kernel above spends 1.4 us per
// k-tile against 0.5 us of MFMA, because its one-barrier-per-k-tile structure (__syncthreads = vmcnt(0)) keeps exactly ONE k-tile of
// LDS-DMA in flight and nothing else is resident to cover the L2 / HBM latency.  Same tile, same fragment / MFMA body, same epilogue —
// but an NST-stage ring with the DMA NST - 1 k-tiles ahead and COUNTED waits.  That is expressible in HIP because the compiler tracks
// in-flight LDS-DMA per LDS object and models __builtin_amdgcn_s_waitcnt (see attn_global_kernel in attention.hip): every ring stage
// is its own static __shared__ array, the DMA is issued in straight-line code (the tail re-fetches the last k-tile instead of
// branching), the wait is the builtin.  buffer_load ... lds with the k-tile's scalar offset: no address arithmetic in the loop.
// One workgroup per CU (NST x 32 KiB of LDS): made for layers whose tile count is below ~2 per CU anyway.
// ---------------------------------------------------------------------------------------------------
template <int NST, int AMODE = 0>      // AMODE 1: implicit 3 x 3 conv (a_src): the A pieces' per-lane offsets follow the k-tile's tap
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmParams p) {
    static_assert(NST == 3 || NST == 4, "ring depth");
    constexpr int BM = 128, BN = 128, TILE = BM * BK * 2, STAGE = 2 * TILE;
    __shared__ __attribute__((aligned(16))) char ring0[STAGE];
    __shared__ __attribute__((aligned(16))) char ring1[STAGE];
    __shared__ __attribute__((aligned(16))) char ring2[STAGE];
    __shared__ __attribute__((aligned(16))) char ring3[NST == 4 ? STAGE : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int tile_m, tile_n;
    tile_of_block<8>((p.M + BM - 1) / BM, p.N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nsplit = p.splitk > 1 ? p.splitk : 1, zsplit = blockIdx.y;
    const int kt0 = zsplit * (p.K / BK) / nsplit, nk = (zsplit + 1) * (p.K / BK) / nsplit;

    // DMA piece i (0..3) of a wave covers rows i*32 + wave*8 .. +8 of the tile, 8 rows x 128 B = 1 KiB; swizzle on the source side
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // conv: the A resource ends with the activation grid, and a tap that leaves the image is an offset beyond it — the buffer load's
    // range check returns the zero padding
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, AMODE == 1 ? p.M * p.lda * 2 : 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    const int prow = lane >> 3, pc = lane & 7;
    int aoff[4], woff[4], apx[4], apy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 32 + wave * 8 + prow;
        const int c = pc ^ ((r >> 1) & 7);
        const int m = min(m0 + r, p.M - 1);
        aoff[i] = (m * p.lda + c * 8) * 2;
        woff[i] = ((n0 + r) * p.ldw + c * 8) * 2;
        apx[i] = AMODE == 1 ? m % p.conv_S : 0;
        apy[i] = AMODE == 1 ? (m / p.conv_S) % p.conv_S : 0;
    }
#define SRH_RING_DMA(kt, ring) { const int so_ = (kt) * (BK * 2); \
    int tap_ = 0, dy_ = 0, dx_ = 0, ashift_ = 0; \
    if (AMODE == 1) { tap_ = (kt) * BK / p.conv_C; dy_ = tap_ / 3 - 1; dx_ = tap_ - 3 * (tap_ / 3) - 1; \
                      ashift_ = ((dy_ * p.conv_S + dx_) * p.lda + (kt) * BK - tap_ * p.conv_C) * 2; } \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
        char* d_ = (ring) + (i * 32 + wave * 8) * 128; \
        if (AMODE == 1) { \
            const bool in_ = (unsigned)(apy[i] + dy_) < (unsigned)p.conv_S && (unsigned)(apx[i] + dx_) < (unsigned)p.conv_S; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)d_, 16, in_ ? aoff[i] + ashift_ : 0x7ffffff0, 0, 0, 0); \
        } else { \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)d_, 16, aoff[i], so_, 0, 0); \
        } \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(d_ + TILE), 16, woff[i], so_, 0, 0); } }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((ks * 2 + fhalf) ^ fkey) << 4);
    const int w_row0 = (wn * 64) * 128, x_row0 = (wm * 64) * 128;
    const int klast = nk - 1;

    // prologue: k-tiles kt0 .. kt0 + NST - 2 into ring stages 0 .. NST - 2
    SRH_RING_DMA(min(kt0, klast), ring0)
    SRH_RING_DMA(min(kt0 + 1, klast), ring1)
    if (NST == 4) SRH_RING_DMA(min(kt0 + 2, klast), ring2)
    // one step = k-tile kt from ring stage `cur`; the DMA of k-tile kt + NST - 1 goes into stage `nxt` (k-tile kt - 1's: every wave is
    // done with it once it has passed this step's barrier).  8 DMA per k-tile and wave: NST - 2 k-tiles may stay in flight at the wait.
#define SRH_RING_STEP(kt, cur, nxt) { \
        __builtin_amdgcn_s_waitcnt(NST == 4 ? 0x4F70 : 0x0F78);          /* vmcnt(16) / vmcnt(8) */ \
        __builtin_amdgcn_s_barrier(); \
        asm volatile("" ::: "memory"); \
        SRH_RING_DMA(min((kt) + NST - 1, klast), nxt) \
        if ((kt) < nk) { \
            const char* sa = (cur) + x_row0; \
            const char* sw = (cur) + TILE + w_row0; \
            f16x8 fwA[2], fxA[2], fwB[2], fxB[2]; \
            __builtin_amdgcn_sched_barrier(0); \
            SRH_FRAG2(fwA, fxA, 0) \
            SRH_FRAG2(fwB, fxB, 1) \
            __builtin_amdgcn_sched_barrier(0); \
            SRH_MMA2(fwA, fxA) \
            __builtin_amdgcn_sched_barrier(0); \
            SRH_FRAG2(fwA, fxA, 2) \
            __builtin_amdgcn_sched_barrier(0); \
            SRH_MMA2(fwB, fxB) \
            __builtin_amdgcn_sched_barrier(0); \
            SRH_FRAG2(fwB, fxB, 3) \
            __builtin_amdgcn_sched_barrier(0); \
            SRH_MMA2(fwA, fxA) \
            SRH_MMA2(fwB, fxB) \
            __builtin_amdgcn_sched_barrier(0); \
        } }
    for (int kt = kt0; kt < nk; kt += NST) {
        if (NST == 4) {
            SRH_RING_STEP(kt, ring0, ring3)
            SRH_RING_STEP(kt + 1, ring1, ring0)
            SRH_RING_STEP(kt + 2, ring2, ring1)
            SRH_RING_STEP(kt + 3, ring3, ring2)
        } else {
            SRH_RING_STEP(kt, ring0, ring2)
            SRH_RING_STEP(kt + 1, ring1, ring0)
            SRH_RING_STEP(kt + 2, ring2, ring1)
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // the tail's redundant fetches have landed
    __syncthreads();                         // every wave is done reading operand tiles: the ring becomes epilogue staging space
    char* stage = wave == 0 ? ring0 : wave == 1 ? ring1 : wave == 2 ? ring2 : (NST == 4 ? ring3 : ring0 + 16384);
    if (nsplit > 1) {  // raw f32 partial sums; bias / residual / activation are applied by splitk_reduce_kernel
        GemmParams q = p;
        q.bias = nullptr; q.resid = nullptr; q.pos = nullptr; q.act = 0; q.out_f16 = nullptr;
        q.out_f32 = p.split_ws + (size_t)zsplit * p.M * p.N; q.ldc = p.N;
        epilogue_staged<2, 0>(q, acc, stage, m0 + wm * 64, n0 + wn * 64, lane);
        return;
    }
    epilogue_staged<2, 0>(p, acc, stage, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------
// gemm_r8_kernel (round 5; a PROBE kernel since gemm_pp_kernel below took its layers) — 128(M) x 256(N) x 64 tiles, EIGHT waves (2 x 4 wave
// tiles of 64 x 64: two waves per SIMD), a three-stage ring of 48 KiB stages (one workgroup per CU), LDS-DMA two k-tiles ahead, ONE barrier
// per k-tile.  Its ablations (probe variants 47-49, 39; profiles/r05_vith_gemm_pp.txt) are what led to the ping-pong loop: 1.04 us per
// k-tile, of which the DMA stream alone needs 0.57, fragment reads + MFMA alone 0.76 and the MFMAs 0.5-0.66 — the parts add up instead of
// overlapping, because the barrier keeps the two waves of every SIMD in lockstep.
// The LDS-DMA is issued from INLINE ASM and waited for with an inline-asm counted s_waitcnt (the form found for the persistent attention
// experiment, attention.hip): the compiler does not know that LDS-DMA is in flight, so the ring needs neither one static LDS object per
// stage nor an unrolled stage sequence — the stage is a runtime offset into one dynamic LDS block — and no conservative vmcnt(0) appears
// before LDS reads.  What that leaves to this code (exactly 6 LDS-DMA and no other VMEM operation per k-tile and wave; tests/test_isa_contract.py
// checks that on the emitted ISA of gemm_pp_kernel and gemm_r320_kernel): RAW — k-tile kt is read behind "s_waitcnt vmcnt(6); s_barrier" (6 = the pieces of k-tile
// kt + 1, the only younger operations); WAR — the DMA of k-tile kt + 2 goes into the stage of k-tile kt - 1 and is issued behind that
// barrier, which every wave passes only after its last fragment read of k-tile kt - 1 has been consumed by an MFMA.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_wave_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_wave_uniform) : "memory");
}



// ---------------------------------------------------------------------------------------------------
// gemm_pp_kernel (round 5) — gemm_r8_kernel's tile, ring and DMA with a PING-PONG k-loop: the two waves of a SIMD (waves w and w + 4)
// are half a k-tile apart.  Every k-tile has a MEMORY slot (issue this wave's six pieces of k-tile kt + 2, read ALL sixteen fragments of
// k-tile kt into registers) and a COMPUTE slot (sixteen MFMAs out of registers, raised priority), one s_barrier between slots; group 0
// (waves 0-3) is in its memory slot while group 1 (waves 4-7) computes and the other way round, so the matrix pipe of every SIMD always
// has one wave with nothing but MFMAs to issue and the LDS / address paths always serve the other one.  Why: r8's ablations
// (profiles/r05_vith_gemm_r8.txt) — per 128 x 256 x 64 k-tile 0.51 us of MFMA, 0.57 us for the DMA + barriers alone, 0.76 us for fragment
// reads + MFMA without DMA, 1.04 us all together: with one barrier per k-tile the two waves of a SIMD run in lockstep (both read, both
// multiply), and nothing overlaps.
//   slot:     2k                2k+1              2k+2
//   group 0:  MEM(k)            CMP(k)            MEM(k+1)
//   group 1:  CMP(k-1)          MEM(k)            CMP(k)
// RAW: k-tile k is read from slot 2k on; every wave waits for its own pieces of k-tile k (vmcnt(6): only k-tile k + 1's are younger)
// before the barrier that opens slot 2k — group 0 at the end of CMP(k-1), group 1 at the end of MEM(k-1).  WAR: MEM(k) sends k-tile
// k + 2 into the stage of k-tile k - 1, last read in slot 2k-1 (group 1's MEM(k-1), lgkmcnt(0) before its barrier).  Compute slots never
// touch LDS.  Both groups pass the same number of barriers (group 1 one before the loop, group 0 one after its last compute slot).
// ---------------------------------------------------------------------------------------------------
// Two geometries (PpCfg): 128 x 256 with 2 x 4 wave tiles of 64 x 64 (qkv-like layers, and fc2-like ones with split-K), and 256 x 160
// with eight stacked 32 x 160 wave tiles (fc1-like layers whose 128 x 256 tiling would spill into a second round: ViT-H fc1 at M = 2048
// is exactly 256 such tiles).  The second has 52 pieces per k-tile for 8 waves: group 0's waves send 7, group 1's send 6 — the counted
// waits are per group anyway.
template <int CFG> struct PpCfg;
template <> struct PpCfg<0> { static constexpr int BM = 128, BN = 256, WAVES_M = 2, TXF = 2, TWF = 2; };
template <> struct PpCfg<1> { static constexpr int BM = 256, BN = 160, WAVES_M = 8, TXF = 1, TWF = 5; };
template <int CFG> constexpr int pp_lds_bytes() { return 3 * (PpCfg<CFG>::BM + PpCfg<CFG>::BN) * BK * 2; }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int CFG>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmParams p) {
    using C = PpCfg<CFG>;
    constexpr int BM = C::BM, BN = C::BN, TXF = C::TXF, TWF = C::TWF, XT = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int XP = BM / 64, WP = BN / 64, WX = (BN / 8) % 8;       // pieces per wave: X, W (every wave), and WX waves send one more W piece
    static_assert(WX == 0 || WX == 4, "the odd W pieces go to exactly group 0's four waves");
    constexpr int NP0 = XP + WP + (WX ? 1 : 0), NP1 = XP + WP;          // pieces per k-tile of a group-0 / group-1 wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD: they are the ping and the pong
    const int wm = C::WAVES_M == 8 ? wave : wave >> 2, wn = C::WAVES_M == 8 ? 0 : wave & 3;
    int tile_m, tile_n;
    tile_of_block<8>((p.M + BM - 1) / BM, p.N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nsplit = p.splitk > 1 ? p.splitk : 1, zsplit = blockIdx.y;
    const int kt0 = zsplit * (p.K / BK) / nsplit, nk = (zsplit + 1) * (p.K / BK) / nsplit;

    // DMA pieces of a k-tile: 8 rows x 128 B each (1 KiB), swizzle on the source side; wave w sends pieces w, w + 8, ...
    const int prow = lane >> 3, pc = lane & 7;
    unsigned xoff[XP], woff[WP + 1];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int r = (i * 8 + wave) * 8 + prow;
        xoff[i] = (unsigned)((min(m0 + r, p.M - 1) * p.lda + ((pc ^ ((r >> 1) & 7)) * 8)) * 2);
    }
#pragma unroll
    for (int i = 0; i < WP + 1; ++i) {
        const int r = min((i * 8 + wave) * 8 + prow, BN - 1);
        woff[i] = (unsigned)(((n0 + r) * p.ldw + ((pc ^ ((r >> 1) & 7)) * 8)) * 2);
    }
    const char* const abase = reinterpret_cast<const char*>(p.A);
    const char* const wbase = reinterpret_cast<const char*>(p.W);
    const int klast = nk - 1;
    auto dma_ktile = [&](int kt, int stage) {
        const char* a = abase + (size_t)kt * (BK * 2);
        const char* w = wbase + (size_t)kt * (BK * 2);
        const unsigned d = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < XP; ++i) dma16_saddr(xoff[i], a, d + i * 8192);
#pragma unroll
        for (int i = 0; i < WP; ++i) dma16_saddr(woff[i], w, d + XT + i * 8192);
        if (WX && !grp) dma16_saddr(woff[WP], w, d + XT + WP * 8192);
    };
    auto wait_next = [&](bool last) {                       // this wave's pieces of the NEXT k-tile have landed (only the one after may be in flight)
        if (last) wait_vmcnt<0>();
        else if (grp) wait_vmcnt<NP1>();
        else wait_vmcnt<NP0>();
    };

    f32x16 acc[TWF][TXF];
#pragma unroll
    for (int i = 0; i < TWF; ++i)
#pragma unroll
        for (int j = 0; j < TXF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((ks * 2 + fhalf) ^ fkey) << 4);
    const int w_row0 = (wn * TWF * 32) * 128, x_row0 = (wm * TXF * 32) * 128;

    dma_ktile(min(kt0, klast), 0);
    dma_ktile(min(kt0 + 1, klast), 1);
    wait_next(false);
    __syncthreads();                                         // k-tile kt0 has landed, everyone's pieces
    if (grp) __syncthreads();                                // group 1 sits out slot 0
    int stage = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        // ---- memory slot: all fragments of k-tile kt into registers, then this wave's pieces of k-tile kt + 2 into the stage of kt - 1
        const int nxt = stage == 0 ? 2 : stage - 1;
        const char* sa = smem + stage * STAGE + x_row0;
        const char* sw = smem + stage * STAGE + XT + w_row0;
        f16x8 fw[4][TWF], fx[4][TXF];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < TWF; ++i) fw[ks][i] = *reinterpret_cast<const f16x8*>(sw + foff[ks] + i * 4096);
#pragma unroll
            for (int j = 0; j < TXF; ++j) fx[ks][j] = *reinterpret_cast<const f16x8*>(sa + foff[ks] + j * 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_ktile(min(kt + 2, klast), nxt);                  // the tail re-fetches the last k-tile: the counts stay exact
        if (grp) wait_next(kt == klast);                     // group 1: k-tile kt + 1 before the barrier that opens slot 2 (kt + 1)
        __syncthreads();                                     // (lgkmcnt(0): the fragments are in registers, the stage may be overwritten)
        // ---- compute slot
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < TWF; ++i)
#pragma unroll
                for (int j = 0; j < TXF; ++j) acc[i][j] = mfma32(fw[ks][i], fx[ks][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!grp) {
            wait_next(kt == klast);
            __syncthreads();
        } else if (kt != klast) {
            __syncthreads();
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    // every piece has landed and every fragment read has completed before the last barrier anyone passed: the ring is staging space
    GemmParams q = p;
    if (nsplit > 1) {  // raw f32 partial sums; bias / residual / activation are applied by splitk_reduce_kernel
        q.bias = nullptr; q.resid = nullptr; q.pos = nullptr; q.act = 0; q.out_f16 = nullptr;
        q.out_f32 = p.split_ws + (size_t)zsplit * p.M * p.N; q.ldc = p.N;
    }
    if constexpr (CFG == 0) epilogue_staged<2, 0>(q, acc, smem + wave * 16384, m0 + wm * 64, n0 + wn * 64, lane);
    else epilogue<TWF, TXF>(q, acc, m0 + wm * TXF * 32, n0 + wn * TWF * 32, lane);
}


// ---------------------------------------------------------------------------------------------------
// 128(M) x 160(N) x 64 LDS-DMA variant for the SMALL-M layers whose 128x128 tile count misses the chip's 512 workgroup slots
// (two workgroups per CU): ViT-H at 256 px, B = 8 has M = 2048 rows, and fc1 (N = 5120) is 640 tiles of 128x128 = two rounds
// with the second a quarter full, but exactly 512 tiles of 128x160 = ONE round.  Four waves stacked along M: wave tile 32(M) x 160(N) = 1 x 5 v_mfma_f32_32x32x16_f16 tiles (80
// accumulators), 2 x (16 + 20) KiB LDS stages = 72 KiB, same DMA / swizzle / one-barrier-per-k-tile structure as the
// 128x128 kernel above.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_glds160_kernel(GemmParams p) {
    constexpr int BM = 128, BN = 160, ATILE = BM * BK * 2, WTILE = BN * BK * 2, STAGE = ATILE + WTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n;
    tile_of_block<8>((p.M + BM - 1) / BM, p.N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nsplit = p.splitk > 1 ? p.splitk : 1, zsplit = blockIdx.y;
    const int kt0 = zsplit * (p.K / BK) / nsplit, nk = (zsplit + 1) * (p.K / BK) / nsplit;

    // DMA piece i of a wave covers rows i*32 + wave*8 .. +8 of a tile (8 rows x 128 B = 1 KiB): 4 pieces of A, 5 of W
    const int prow = lane >> 3, pc = lane & 7;
    const char* asrc[4];
    const char* wsrc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int r = i * 32 + wave * 8 + prow;
        const int c = pc ^ ((r >> 1) & 7);
        if (i < 4) asrc[i] = reinterpret_cast<const char*>(p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + c * 8);
        wsrc[i] = reinterpret_cast<const char*>(p.W + (size_t)(n0 + r) * p.ldw + c * 8);
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
#define SRH_DMA_TILE160(kt, stage) { \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) { \
        char* d_ = smem + (stage) * STAGE + (i * 32 + wave * 8) * 128; \
        if (i < 4) __builtin_amdgcn_global_load_lds((glb_ptr)(asrc[i] + (size_t)(kt) * (BK * 2)), (lds_ptr)d_, 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((glb_ptr)(wsrc[i] + (size_t)(kt) * (BK * 2)), (lds_ptr)(d_ + ATILE), 16, 0, 0); } }

    f32x16 acc[5][1];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((ks * 2 + fhalf) ^ fkey) << 4);
    const int x_row0 = (wave * 32) * 128;

    SRH_DMA_TILE160(kt0, 0)
    for (int kt = kt0; kt < nk; ++kt) {
        const int stage = (kt - kt0) & 1;
        __syncthreads();   // retires this wave's DMA (vmcnt(0)) and orders everyone's; frees stage^1
        if (kt + 1 < nk) SRH_DMA_TILE160(kt + 1, stage ^ 1)
        const char* sa = smem + stage * STAGE + x_row0;
        const char* sw = smem + stage * STAGE + ATILE;
        f16x8 fwA[5], fwB[5], fxA, fxB;
#define SRH_FRAG160(fw, fx, ks) { fx = *reinterpret_cast<const f16x8*>(sa + foff[ks]); \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) fw[i] = *reinterpret_cast<const f16x8*>(sw + foff[ks] + i * 4096); }
#define SRH_MMA160(fw, fx) { _Pragma("unroll") for (int i = 0; i < 5; ++i) acc[i][0] = mfma32(fw[i], fx, acc[i][0]); }
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG160(fwA, fxA, 0)
        SRH_FRAG160(fwB, fxB, 1)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA160(fwA, fxA)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG160(fwA, fxA, 2)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA160(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG160(fwB, fxB, 3)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA160(fwA, fxA)
        SRH_MMA160(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (nsplit > 1) {  // raw f32 partial sums; bias / residual / activation are applied by splitk_reduce_kernel
        GemmParams q = p;
        q.bias = nullptr; q.resid = nullptr; q.pos = nullptr; q.act = 0; q.out_f16 = nullptr;
        q.out_f32 = p.split_ws + (size_t)zsplit * p.M * p.N; q.ldc = p.N;
        epilogue<5, 1>(q, acc, m0 + wave * 32, n0, lane);
        return;
    }
    epilogue<5, 1>(p, acc, m0 + wave * 32, n0, lane);
}

// ---------------------------------------------------------------------------------------------------
// gemm_r320_kernel (round 5): 128(M) x 320(N) x 64 tiles, EIGHT waves (4 along M x 2 along N, the 32 x 160 wave tile of
// gemm_glds160_kernel), two 56-KiB stages, one workgroup per CU, LDS-DMA from inline asm (see gemm_r8_kernel).  For the small-M layers
// that are one round of such tiles — ViT-H fc1 at M = 2048: 16 x 16 = 256 tiles.  Why the wider tile: these layers are bound by the
// rate at which operand bytes reach LDS, and a 128 x 320 tile moves M N K 2 (1/128 + 1/320) bytes against (1/128 + 1/160) for
// 128 x 160 — 293 instead of 377 MB for fc1 (profiles/r05_vith_gemm_r8.txt).  Why two stages suffice here and not in gemm_glds_kernel:
// there the compiler puts a vmcnt(0) in front of the fragment reads (it cannot tell the stages of one dynamic LDS block apart), so the
// DMA of k-tile kt + 1 had to LAND before k-tile kt was computed and only a second resident workgroup covered it; with the DMA issued
// from inline asm the compiler knows of nothing to wait for, and the DMA of k-tile kt + 1 lands under the MFMAs of k-tile kt.
// tests/test_isa_contract.py checks the loop on the emitted ISA: 7 LDS-DMA per k-tile and wave, no other VMEM operation.
// ---------------------------------------------------------------------------------------------------
constexpr int R320_BM = 128, R320_BN = 320, R320_XT = R320_BM * BK * 2, R320_WT = R320_BN * BK * 2, R320_STAGE = R320_XT + R320_WT;
constexpr int R320_LDS = 2 * R320_STAGE;           // 114 688 B

__global__ __launch_bounds__(512, 1) void gemm_r320_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    int tile_m, tile_n;
    tile_of_block<8>((p.M + R320_BM - 1) / R320_BM, p.N / R320_BN, tile_m, tile_n);
    const int m0 = tile_m * R320_BM, n0 = tile_n * R320_BN;
    const int nk = p.K / BK;

    // DMA pieces of a k-tile (8 rows x 128 B, swizzle on the source side): X has 16 (wave w: w, w + 8), W has 40 (w, w + 8, .., w + 32)
    const int prow = lane >> 3, pc = lane & 7;
    unsigned xoff[2], woff[5];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (i * 8 + wave) * 8 + prow;
        xoff[i] = (unsigned)((min(m0 + r, p.M - 1) * p.lda + ((pc ^ ((r >> 1) & 7)) * 8)) * 2);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int r = (i * 8 + wave) * 8 + prow;
        woff[i] = (unsigned)(((n0 + r) * p.ldw + ((pc ^ ((r >> 1) & 7)) * 8)) * 2);
    }
    const char* const abase = reinterpret_cast<const char*>(p.A);
    const char* const wbase = reinterpret_cast<const char*>(p.W);
    auto dma_ktile = [&](int kt, int stage) {
        const char* a = abase + (size_t)kt * (BK * 2);
        const char* w = wbase + (size_t)kt * (BK * 2);
        const unsigned d = lds0 + stage * R320_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16_saddr(xoff[i], a, d + i * 8192);
#pragma unroll
        for (int i = 0; i < 5; ++i) dma16_saddr(woff[i], w, d + R320_XT + i * 8192);
    };

    f32x16 acc[5][1];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((ks * 2 + fhalf) ^ fkey) << 4);
    const int x_row0 = (wm * 32) * 128, w_row0 = (wn * 160) * 128;

    dma_ktile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // k-tile kt has landed (this wave's pieces: nothing else is in flight)
        __syncthreads();                                     // ... every wave's; and every wave is done with the other stage (k-tile kt - 1)
        dma_ktile(min(kt + 1, nk - 1), stage ^ 1);           // the tail re-fetches the last k-tile (no branch around the asm)
        const char* sa = smem + stage * R320_STAGE + x_row0;
        const char* sw = smem + stage * R320_STAGE + R320_XT + w_row0;
        f16x8 fwA[5], fwB[5], fxA, fxB;
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG160(fwA, fxA, 0)
        SRH_FRAG160(fwB, fxB, 1)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA160(fwA, fxA)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG160(fwA, fxA, 2)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA160(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG160(fwB, fxB, 3)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA160(fwA, fxA)
        SRH_MMA160(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tail's redundant fetch must not outlive the workgroup's LDS
    epilogue<5, 1>(p, acc, m0 + wm * 32, n0 + wn * 160, lane);
}

// out = act(sum_z partial[z] + bias) (+ resid): the partial sums are added in ascending z (fixed order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
    const size_t quad = (size_t)blockIdx.x * 256 + threadIdx.x;          // 4 consecutive columns
    const size_t total = (size_t)p.M * p.N / 4;
    if (quad >= total) return;
    const int m = (int)(quad / (p.N / 4)), n = (int)(quad % (p.N / 4)) * 4;
    const size_t MN = (size_t)p.M * p.N;
    float4 a = *reinterpret_cast<const float4*>(p.split_ws + (size_t)m * p.N + n);
    for (int z = 1; z < p.splitk; ++z) {
        const float4 b = *reinterpret_cast<const float4*>(p.split_ws + z * MN + (size_t)m * p.N + n);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
    if (p.act == 1) { for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]); }
    else if (p.act == 2) { for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f); }
    if (p.resid) { const float4 r = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
    if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    if (p.out_f16) {
        const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h;
    }
}

// ---------------------------------------------------------------------------------------------------
// 256x256x64 LDS-DMA variant: 512 threads = 8 waves as 4(N) x 2(M), wave tile 64(N) x 128(M) = 2x4 MFMA
// tiles (128 accumulator registers), 2 x 64 KiB LDS stages, one workgroup per CU.  Half the L2->LDS bytes
// per FLOP of the 128x128 tile and 0.75 ds_read_b128 per MFMA instead of 1.
// ---------------------------------------------------------------------------------------------------
template <int ABL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_glds256_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, TILE = BM * BK * 2, STAGE = 2 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int tile_m, tile_n;
    tile_of_block<4>((p.M + BM - 1) / BM, p.N / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / BK;

    // DMA piece i (0..3) of a wave covers rows i*64 + wave*8 .. +8 of the 256-row tile
    const int prow = lane >> 3, pc = lane & 7;
    const char* asrc[4];
    const char* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 64 + wave * 8 + prow;
        const int c = pc ^ ((r >> 1) & 7);
        asrc[i] = reinterpret_cast<const char*>(p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + c * 8);
        wsrc[i] = reinterpret_cast<const char*>(p.W + (size_t)(n0 + r) * p.ldw + c * 8);
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
#define SRH_DMA_TILE256(kt, stage) { \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
        char* d_ = smem + (stage) * STAGE + (i * 64 + wave * 8) * 128; \
        __builtin_amdgcn_global_load_lds((glb_ptr)(asrc[i] + (size_t)(kt) * (BK * 2)), (lds_ptr)d_, 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((glb_ptr)(wsrc[i] + (size_t)(kt) * (BK * 2)), (lds_ptr)(d_ + TILE), 16, 0, 0); } }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int fkey = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = frow * 128 + (((ks * 2 + fhalf) ^ fkey) << 4);
    const int w_row0 = (wn * 64) * 128, x_row0 = (wm * 128) * 128;

    SRH_DMA_TILE256(0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        __syncthreads();
        if (ABL != 1 && kt + 1 < nk) SRH_DMA_TILE256(kt + 1, stage ^ 1)
        if (ABL == 2) continue;
        const char* sa = smem + stage * STAGE + x_row0;
        const char* sw = smem + stage * STAGE + TILE + w_row0;
        f16x8 fwA[2], fxA[4], fwB[2], fxB[4];
#define SRH_FRAG4(fw, fx, ks) { fw[0] = *reinterpret_cast<const f16x8*>(sw + foff[ks]); \
        fw[1] = *reinterpret_cast<const f16x8*>(sw + foff[ks] + 4096); \
        fx[0] = *reinterpret_cast<const f16x8*>(sa + foff[ks]); fx[1] = *reinterpret_cast<const f16x8*>(sa + foff[ks] + 4096); \
        fx[2] = *reinterpret_cast<const f16x8*>(sa + foff[ks] + 8192); fx[3] = *reinterpret_cast<const f16x8*>(sa + foff[ks] + 12288); }
#define SRH_MMA4(fw, fx) { \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]); }
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG4(fwA, fxA, 0)
        SRH_FRAG4(fwB, fxB, 1)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA4(fwA, fxA)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG4(fwA, fxA, 2)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA4(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
        SRH_FRAG4(fwB, fxB, 3)
        __builtin_amdgcn_sched_barrier(0);
        SRH_MMA4(fwA, fxA)
        SRH_MMA4(fwB, fxB)
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    epilogue_staged<4, 0>(p, acc, smem + wave * 16384, m0 + wm * 128, n0 + wn * 64, lane);
    __builtin_amdgcn_wave_barrier();
    epilogue_staged<4, 2>(p, acc, smem + wave * 16384, m0 + wm * 128 + 64, n0 + wn * 64, lane);
}

// Small-M layers: take 128x160 tiles (gemm_glds160_kernel) when more than one round of 128x128 tiles is one round of 128x160 tiles
// (ViT-H fc1 at M = 2048: 640 -> 512 workgroups on 512 slots; measured 46.0 -> 41.3 us).  Not for the few-tile layers: proj / fc2 of
// ViT-H as 128 tiles x split-K 4 measured no better than 128x128 tiles without / with split-K 3 (profiles/r03_vith_gemm_t160.txt).
static bool use_tile160(const GemmParams& p) {
    if (p.conv_S > 0 || p.pos || p.N % 160 != 0 || p.N % 128 != 0 || p.K % BK != 0 || p.M >= 4096) return false;
    const long t128 = (long)((p.M + 127) / 128) * (p.N / 128), t160 = (long)((p.M + 127) / 128) * (p.N / 160);
    return t128 > 512 && t160 <= 512;
}

static bool r320_applies(const GemmParams& p) {     // ViT-H fc1-like: one round of 128 x 320 tiles where 128 x 256 tiles would need two
    if (p.conv_S > 0 || p.pos || p.variant != 0 || p.M >= 4096 || p.M < 128 || p.N % R320_BN != 0 || p.K % BK != 0 || p.K / BK < 6 || z192_preferred(p)) return false;
    const long t320 = (long)((p.M + R320_BM - 1) / R320_BM) * (p.N / R320_BN);
    const long t256 = p.N % 256 == 0 ? (long)((p.M + 127) / 128) * (p.N / 256) : 1 << 30;
    return t320 >= 160 && t320 <= 256 && t256 > 256;
}

// gemm_pp_kernel<0>'s share of the small-M layers (profiles/r05_vith_gemm_pp.txt):
//   * those that are ONE round of 128 x 256 tiles with at least ~5/8 of the CUs busy — ViT-H qkv: 240 tiles, 30.4 (gemm_r8_kernel, round 5's
//     first answer, now a probe kernel) -> 25.5 us; ViT-L qkv / fc1;
//   * the few-tile deep-K ones with split-K — ViT-H fc2: 80 tiles x 3 slices, 43.6 us with the reduce pass against 45.9 on 128 x 128 tiles.
// ViT-H proj (80 tiles, K = 1280: slices of 6 k-tiles) stays on the 128 x 128 ring kernel (18.4 against 22.9 / 25.7 us).
static bool pp_shape_ok(const GemmParams& p) {
    return !(p.conv_S > 0 || p.pos || p.variant != 0 || p.M >= 4096 || p.M < 128 || p.N % 256 != 0 || p.K % BK != 0 || p.K / BK < 6 || z192_preferred(p));
}
static bool pp_applies(const GemmParams& p) {
    if (!pp_shape_ok(p)) return false;
    const long tiles = (long)((p.M + 127) / 128) * (p.N / 256);
    return tiles >= 160 && tiles <= 256;
}
static int pp_split_factor(const GemmParams& p) {           // > 1: gemm_pp_kernel<0> with that many K slices
    if (!pp_shape_ok(p) || p.M % 128 != 0) return 1;
    const long tiles = (long)(p.M / 128) * (p.N / 256);
    const int nk = p.K / BK;
    if (tiles > 128 || nk < 48) return 1;
    int s = (int)(256 / tiles);
    if (s > 4) s = 4;
    while (s > 1 && nk / s < 24) --s;
    if (tiles * s < 192) return 1;             // too few workgroups for one per CU (ViT-L fc2: 64 tiles x 2: 37.1 against 36.1 us)
    return s;
}

// Split-K only pays when the tiles cannot fill the chip's workgroup slots and K is deep enough to share out.
int gemm_splitk_factor(const GemmParams& p) {
    if (pp_applies(p) || r320_applies(p)) return 1;
    if (p.conv_S > 0 || p.pos || p.variant != 0 || p.M % 128 != 0 || p.N % 128 != 0 || p.K % BK != 0) return 1;
    if (p.M >= 4096 || z192_preferred(p)) return 1;
    if (use_tile160(p)) return 1;
    if (const int s = pp_split_factor(p); s > 1) return s;
    const long tiles = (long)(p.M / 128) * (p.N / 128);
    const int nk = p.K / BK;
    if (tiles >= 224 || nk < 16) return 1;
    int s = (int)(512 / tiles);
    if (s > 4) s = 4;
    while (s > 1 && nk / s < 12) --s;          // a slice shorter than 12 k-tiles does not pay for the reduce pass (ViT-H proj: K = 1280)
    return s;
}

static int launched() { return hipGetLastError() == hipSuccess ? 0 : -3; }

static int launch_splitk_reduce(const GemmParams& p, hipStream_t stream) {
    const size_t quads = (size_t)p.M * p.N / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, p);
    return launched();
}

#ifdef SRH_TUNING      // probe builds only (tools/probes/build_probes.sh, -Itools/probes): the probe kernels, their ablation switches and the by-number dispatcher
#include "gemm_tuning.inc"
#endif

// Kernel choice by shape — the whole product dispatch:
//   3x3 conv (neck)                                     -> the three-stage LDS-DMA ring kernel with tap-following A pieces (gemm_ring_kernel<3, 1>)
//   big fp16-output layers with a bias (ViT-B blocks)   -> gemm_z192 (generated body, persistent 256 x 192 tiles, deferred epilogue)
//   M >= 4096, N % 256 == 0, tiles fill the chip        -> 256 x 256 LDS-DMA tiles
//   small M (ViT-L / ViT-H at 256 px)                   -> 128 x 256 tiles on the 8-wave ping-pong kernel (+ split-K, partials reduced in order by a
//                                                          reduce pass or by the caller's next LayerNorm) | 128 x 320 | 128 x 160 | 128 x 128 split-K | 3-stage ring
//   everything else                                     -> 128 x 128 LDS-DMA tiles, two workgroups per CU
int launch_gemm(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0) return 0;
    if (p.N % 128 != 0 || p.K % BK != 0) return -2;
    if (p.conv_S > 0) {                           // implicit 3 x 3 conv (the neck): the ring kernel with tap-following A pieces; the zero padding is
        if (p.conv_C % BK != 0 || p.K != 9 * p.conv_C || p.splitk > 1 || (long)p.M * p.lda * 2 >= 0x7ffffff0L) return -2;      // the buffer load's range check
        hipLaunchKernelGGL((gemm_ring_kernel<3, 1>), dim3(((p.M + 127) / 128) * (p.N / 128)), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
#ifdef SRH_TUNING      // probe builds: kernel / ablation selection by number (tools/probes/gemm_tuning.inc)
    { int rc; if (gemm_tuning_dispatch(p, stream, &rc)) return rc; }
#endif
    const bool have_tables = p.ztab != nullptr;  // gemm_z192's tile tables live in the caller's context
    if (p.a_blocked16 || p.out_blocked16)        // only gemm_z192 understands the blocked-16 layout
        return have_tables && z192_preferred(p) ? launch_gemm_z192(p, stream) : -2;
    if (have_tables && z192_preferred(p)) return launch_gemm_z192(p, stream);

    static OncePerDevice opt_in;                 // dynamic-LDS opt-in of this file's LDS-DMA kernels, once per device
    if (!opt_in.run([] {
            const std::pair<const void*, int> k[] = {
                {reinterpret_cast<const void*>(gemm_glds_kernel<0, 2>), 65536}, {reinterpret_cast<const void*>(gemm_glds160_kernel), 73728},
                {reinterpret_cast<const void*>(gemm_glds256_kernel<0>), 131072},
                {reinterpret_cast<const void*>(gemm_pp_kernel<0>), pp_lds_bytes<0>()}, {reinterpret_cast<const void*>(gemm_r320_kernel), R320_LDS},
            };
            for (const auto& f : k)
                if (hipFuncSetAttribute(f.first, hipFuncAttributeMaxDynamicSharedMemorySize, f.second) != hipSuccess) return false;
            return true;
        }))
        return -3;

    // 256x256 tiles halve the L2->LDS traffic per FLOP but there are only 256 CUs: use them when the tile count fills the chip evenly
    // (half to one round, or >= 80 % occupancy of the last round).  Not for short-K layers (decoder ConvT: K = 128 / 256): those are all
    // prologue + epilogue, and two 128x128 workgroups per CU overlap each other's store tail.  Not for narrow layers either: the neck's
    // 1x1 conv (N = 256, M = 16384) is 64 such tiles — a quarter of the CUs, 31 us — and one round of 128 x 128 tiles on the ring kernel.
    const long t256 = (long)((p.M + 255) / 256) * (p.N / 256);
    const bool fits256 = (t256 >= 128 && t256 <= 256) || (t256 > 256 && (double)t256 / (double)(((t256 + 255) / 256) * 256) >= 0.8);
    if (p.N % 256 == 0 && p.M >= 4096 && p.K > 256 && fits256) {
        hipLaunchKernelGGL(gemm_glds256_kernel<0>, dim3((unsigned)t256), dim3(512), 131072, stream, p);
        return launched();
    }
    if (r320_applies(p)) {                        // ... or in one round of 128 x 320 tiles (ViT-H fc1)
        GemmParams q = p;
        q.splitk = 1;
        hipLaunchKernelGGL(gemm_r320_kernel, dim3(((p.M + R320_BM - 1) / R320_BM) * (p.N / R320_BN), 1), dim3(512), R320_LDS, stream, q);
        return launched();
    }
    if (pp_applies(p)) {                          // small-M layers in one round of 128 x 256 tiles on the ping-pong kernel
        GemmParams q = p;
        q.splitk = 1;
        hipLaunchKernelGGL(gemm_pp_kernel<0>, dim3(((p.M + 127) / 128) * (p.N / 256), 1), dim3(512), pp_lds_bytes<0>(), stream, q);
        return launched();
    }
    const int grid = ((p.M + 127) / 128) * (p.N / 128);
    if (use_tile160(p)) {                         // one round of 160-wide tiles (gemm_splitk_factor is 1 for these layers)
        GemmParams q = p;
        q.splitk = 1;
        hipLaunchKernelGGL(gemm_glds160_kernel, dim3(((p.M + 127) / 128) * (p.N / 160), 1), dim3(256), 73728, stream, q);
        return launched();
    }
    if (p.split_ws && p.splitk > 1) {             // deterministic split-K: f32 partials + a reduce pass in ascending slice order
        if (p.splitk != gemm_splitk_factor(p) || p.splitk > (p.K / BK)) return -2;
        if (p.splitk == pp_split_factor(p))       // few 128 x 256 tiles, deep K: the ping-pong kernel's slices
            hipLaunchKernelGGL(gemm_pp_kernel<0>, dim3((p.M / 128) * (p.N / 256), p.splitk), dim3(512), pp_lds_bytes<0>(), stream, p);
        else
            hipLaunchKernelGGL((gemm_glds_kernel<0, 2>), dim3(grid, p.splitk), dim3(256), 65536, stream, p);
        return p.defer_reduce ? launched() : launch_splitk_reduce(p, stream);
    }
    // at most one 128 x 128 tile per CU and a k-loop long enough to fill a ring: the three-stage ring kernel (one workgroup per CU, the
    // L2 / HBM latency hidden by depth instead of by a second resident workgroup): ViT-H proj 21.3 -> 18.3 us (profiles/r04_gemm_ring_vith.txt)
    if (grid <= 256 && p.K / BK >= 12) {
        hipLaunchKernelGGL((gemm_ring_kernel<3>), dim3(grid), dim3(256), 0, stream, p);
        return launched();
    }
    hipLaunchKernelGGL((gemm_glds_kernel<0, 2>), dim3(grid), dim3(256), 65536, stream, p);
    return launched();
}

}  // namespace srh

// The naive map decoder (one fused kernel) and the scene-level mask fusion.
//
// Reference: model.py:286-295 (ConvTranspose2d k2 s2 x4, LayerNorm2d, exact GELU), :445-446/:490-494
// (sigmoid, NCHW -> NHWC) and inferencer.py:79-110 (scatter-add of the two masks + coverage
// counter, divide, x255, truncate to u8).
//
// A stride-2 kernel-2 transposed conv does not overlap: every input pixel owns a 2x2 output
// block, so each layer is a per-pixel GEMM to 4*Cout columns and the successive activations of a
// token form a quad tree (pixel, sub1, sub2, sub3) that one wave can walk in registers.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

// ---- the whole map_decoder in one kernel ----------------------------------------------------------------------------
// ConvT(256->128, k2 s2) + LayerNorm2d(128) + GELU + ConvT(128->64) + GELU + ConvT(64->32) + GELU + ConvT(32->2) + sigmoid + the
// quad-tree -> NHWC scatter (model.py:286-295, :445-446).  Every stage is per-source-token independent (a stride-2 kernel-2
// transposed conv never overlaps), so nothing but the 8 MB of neck output comes in and nothing but the masks goes out; the
// layer-by-layer path (two GEMM launches, a LayerNorm2d pass and the two-layer tail kernel) moved 33 MB f32 + 17 MB + 33 MB of
// intermediates through HBM for 0.84 GFLOP per tile.
// Organisation: the transposed MFMA chain of topo_fused.hip.  Y^T[feature, token] = W . X^T with v_mfma_f32_16x16x32_f16, A = a
// packed 16 x 32 weight fragment, B = activations; a C tile pair of one layer IS the B operand of the next if the next layer's
// weights are packed with the k permutation 8 g + j -> 16 (j >> 2) + 4 g + (j & 3) (api.hip pack_decoder_fused).  A wave job is
// 16 * NCG tokens (NCG column groups that share every A fragment read) and ONE first-level sub-pixel sub1 of them: its slice of layer 0 is 128 of the
// 512 output columns, LayerNorm2d normalises exactly those 128 channels (lane-local + two cross-lane adds), and the three later
// layers expand it depth-first — per second-level sub-pixel 64 -> 4 x 32 channels -> 4 x (2 x 2 pixels x 2 classes) — so at most 64
// accumulator registers per column group are live.  A workgroup (NW waves) has sub1 = blockIdx & 3 fixed: its LDS holds that 64 KiB slice of
// layer 0 plus all of layers 3 (64 KiB) and 5 (16 KiB), staged once by LDS-DMA; the last layer's 8 x 32 f32 weights sit in registers as an fp16 hi + lo
// pair (two MFMAs keep their f32 value to 2^-22).  Per lane the last tile holds one 2-pixel x 2-class float4 of one output row.
// GELU = gelu_fast (degree-5 exponent polynomial, common.hpp; the degree-3 one of the z192 bodies saved 2.5 us and cost 0.5e-4 of the 5e-4
// mask-score bound in the randomised sweep: not taken).  Bound: VALU (58.7 M GELUs + 8.4 M sigmoids per 16 tiles) behind a serial
// per-wave chain; HBM floor 8 MB in + 33.5 MB out.  Measured (profiles/r06_decoder_probe.txt): 39 us alone at B = 16, ~45 in the model.
constexpr int DF_W0 = 65536, DF_W3 = 65536, DF_W5 = 16384, DF_PRM_BYTES = 3072, DF_LDS = DF_W0 + DF_W3 + DF_W5 + DF_PRM_BYTES;   // parameters: 738 floats, packed into 3 KiB (api.hip)
// NCG column groups of 16 tokens per wave job x NW waves per workgroup: <2, 8> shares every fragment read between two column groups
// (half the LDS traffic, 170 VGPRs: two waves per SIMD; measured 52 us in the model); <1, 16> — the one launched — has four waves per
// SIMD (104 VGPRs) to hide the serial MFMA -> bias -> GELU -> pack -> MFMA chain of a job behind other waves' phases (47 -> ~42 us).

__device__ __forceinline__ f16x8 df_pack8(const f32x4& a, const f32x4& b) {
    f16x8 r;
    r[0] = (f16)a[0]; r[1] = (f16)a[1]; r[2] = (f16)a[2]; r[3] = (f16)a[3];
    r[4] = (f16)b[0]; r[5] = (f16)b[1]; r[6] = (f16)b[2]; r[7] = (f16)b[3];
    return r;
}
__device__ __forceinline__ float df_sum4(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }

template <int NCG, int DF_WAVES>
__global__ __launch_bounds__(DF_WAVES * 64) void decode_fused_kernel(DecodeFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const w0 = smem;
    char* const w3 = smem + DF_W0;
    char* const w5 = smem + DF_W0 + DF_W3;
    float* const prm = reinterpret_cast<float*>(smem + DF_W0 + DF_W3 + DF_W5);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int sub1 = blockIdx.x & 3;
    {   // this workgroup's weights -> LDS, once, by LDS-DMA (1 KiB pieces, lane-linear): layer 0's slice for sub1 (64 pieces), layers 3 and 5
        // (80, contiguous in the packed stream) and the parameters (3).  (A copy loop through registers took its L2 round trips one
        // after the other: 9 per thread.)
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        const char* s0 = p.frags + (size_t)sub1 * DF_W0 + lane * 16;
        const char* s1 = p.frags + (size_t)4 * DF_W0 + lane * 16;
        for (int pc = wave; pc < DF_W0 / 1024; pc += DF_WAVES)
            __builtin_amdgcn_global_load_lds((glb_ptr)(s0 + pc * 1024), (lds_ptr)(w0 + pc * 1024), 16, 0, 0);
        for (int pc = wave; pc < (DF_W3 + DF_W5) / 1024; pc += DF_WAVES)
            __builtin_amdgcn_global_load_lds((glb_ptr)(s1 + pc * 1024), (lds_ptr)(w3 + pc * 1024), 16, 0, 0);
        if (wave < DF_PRM_BYTES / 1024)
            __builtin_amdgcn_global_load_lds((glb_ptr)(reinterpret_cast<const char*>(p.prm) + wave * 1024 + lane * 16),
                                             (lds_ptr)(reinterpret_cast<char*>(prm) + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the last layer's A fragment: rows n < 8 = (ky, kx, class), k = the 32 channels in the permuted order
    f16x8 a7h, a7l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = 16 * (j >> 2) + 4 * g + (j & 3);
        const float v = n < 8 ? prm[480 + n * 32 + ch] : 0.f;
        a7h[j] = (f16)v;
        a7l[j] = (f16)(v - (float)a7h[j]);
    }
    const float b7a = prm[736], b7b = prm[737];
#define DF_FR(base, fi) (*reinterpret_cast<const f16x8*>((base) + (fi) * 1024 + lane * 16))
    const int S = p.S, P = S * 16;
    constexpr int JT = 16 * NCG;                                     // tokens per wave job
    const int ngroups = (p.B * S * S) / JT;                          // (S * S is a multiple of 256)
    const int wg_stride = (gridDim.x >> 2) * DF_WAVES;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int tg = (blockIdx.x >> 2) * DF_WAVES + wave; tg < ngroups; tg += wg_stride) {
        // ---- layer 0 slice: U^T[co, token] for this sub1, K = 256 straight from memory
        f16x8 xin[NCG][8];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const f16* xr = p.emb16 + ((size_t)tg * JT + cg * 16 + n) * 256 + 8 * g;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) xin[cg][kb] = *reinterpret_cast<const f16x8*>(xr + 32 * kb);
        }
        f32x4 a0[8][NCG];
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) a0[rt][cg] = zero;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const f16x8 a = DF_FR(w0, kb * 8 + rt);
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) a0[rt][cg] = mfma16(a, xin[cg][kb], a0[rt][cg]);
                if (rt == 7) __builtin_amdgcn_sched_barrier(0);      // hipcc would hoist all 64 fragment reads (256 registers) to the top
            }
        // ---- bias + LayerNorm2d over the 128 channels of (token, sub1) + GELU -> B operands of layer 3
        f16x8 x1[NCG][4];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            float sum = 0.f;
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(prm + 16 * rt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) { a0[rt][cg][r] += b[r]; sum += a0[rt][cg][r]; }
            }
            const float mean = df_sum4(sum) * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int rt = 0; rt < 8; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { a0[rt][cg][r] -= mean; q = fmaf(a0[rt][cg][r], a0[rt][cg][r], q); }
            const float var = df_sum4(q) * (1.0f / 128.0f);
            if (p.nf && !(var < INFINITY) && g == 0) p.nf[p.nf_tag] = 1u;
            const float rstd = rsqrtf(var + 1e-6f);
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(prm + 128 + 16 * rt + 4 * g);
                const f32x4 be = *reinterpret_cast<const f32x4*>(prm + 256 + 16 * rt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[rt][cg][r] = gelu_fast(a0[rt][cg][r] * rstd * ga[r] + be[r]);
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) x1[cg][kb] = df_pack8(a0[2 * kb][cg], a0[2 * kb + 1][cg]);
        }
        // the lanes' tokens and their level-1 pixel
        int ob[NCG], oy[NCG], ox[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            int tk = tg * JT + cg * 16 + n;
            const int px = tk % S; tk /= S;
            const int py = tk % S; ob[cg] = tk / S;
            oy[cg] = py * 2 + (sub1 >> 1); ox[cg] = px * 2 + (sub1 & 1);
        }
#pragma unroll 1
        for (int s2 = 0; s2 < 4; ++s2) {
            // ---- layer 3, the 64 channels of sub2: K = 128
            f32x4 a3[4][NCG];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) a3[rt][cg] = zero;
            const char* w3s = w3 + s2 * 16384;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f16x8 a = DF_FR(w3s, kb * 4 + rt);
#pragma unroll
                    for (int cg = 0; cg < NCG; ++cg) a3[rt][cg] = mfma16(a, x1[cg][kb], a3[rt][cg]);
                    if (rt == 3 && (kb & 1)) __builtin_amdgcn_sched_barrier(0);
                }
            f16x8 x2[NCG][2];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(prm + 384 + 16 * rt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a3[rt][cg][r] = gelu_fast(a3[rt][cg][r] + b[r]);
                }
                x2[cg][0] = df_pack8(a3[0][cg], a3[1][cg]);
                x2[cg][1] = df_pack8(a3[2][cg], a3[3][cg]);
            }
            // ---- layer 5: 64 -> 4 sub3 x 32 channels, K = 64
            f32x4 a5[8][NCG];
#pragma unroll
            for (int rt = 0; rt < 8; ++rt)
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) a5[rt][cg] = zero;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int rt = 0; rt < 8; ++rt) {
                    const f16x8 a = DF_FR(w5, kb * 8 + rt);
#pragma unroll
                    for (int cg = 0; cg < NCG; ++cg) a5[rt][cg] = mfma16(a, x2[cg][kb], a5[rt][cg]);
                    if (rt == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(prm + 448 + 16 * (rt & 1) + 4 * g);
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a5[rt][cg][r] = gelu_fast(a5[rt][cg][r] + b[r]);
            }
            // ---- layer 7 (32 -> 2 x 2 pixels x 2 classes) + sigmoid + scatter; rows 4 g + r: g = ky, r = kx * 2 + class
#pragma unroll
            for (int s3 = 0; s3 < 4; ++s3)
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) {
                    const f16x8 ub = df_pack8(a5[2 * s3][cg], a5[2 * s3 + 1][cg]);
                    f32x4 o = mfma16(a7h, ub, zero);
                    o = mfma16(a7l, ub, o);
                    if (g < 2) {
                        const int y = (((oy[cg] * 2 + (s2 >> 1)) * 2 + (s3 >> 1)) * 2) + g;
                        const int xx = ((ox[cg] * 2 + (s2 & 1)) * 2 + (s3 & 1)) * 2;
                        const size_t off = (((size_t)ob[cg] * P + y) * P + xx) * 2;
                        const float4 lg = make_float4(o[0] + b7a, o[1] + b7b, o[2] + b7a, o[3] + b7b);
                        if (p.logits) *reinterpret_cast<float4*>(p.logits + off) = lg;
                        if (p.scores) {
                            // sigmoid as rcp(1 + exp2(-x log2 e)): v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~25-instruction
                            // expf + IEEE division sequence
                            auto sg = [](float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); };
                            *reinterpret_cast<float4*>(p.scores + off) = make_float4(sg(lg.x), sg(lg.y), sg(lg.z), sg(lg.w));
                        }
                    }
                }
        }
    }
#undef DF_FR
}

int launch_decode_fused(const DecodeFusedParams& p, hipStream_t s) {
    const long T = (long)p.B * p.S * p.S;
    if (T <= 0) return 0;
    if (T % 32 || (p.S != 16 && p.S != 32 && p.S != 64)) return -2;
    constexpr int NCG = 1, NW = 16;
    static OncePerDevice opt_in;
    if (!opt_in.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<NCG, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, DF_LDS) == hipSuccess; }))
        return -3;
    const long ngroups = T / (16 * NCG);
    const long per_sub = (ngroups + NW - 1) / NW;                         // workgroups per sub1 when every wave gets one job
    const unsigned grid = 4u * (unsigned)(per_sub < 64 ? per_sub : 64);   // one workgroup per CU at most (150 KiB of LDS each)
    hipLaunchKernelGGL((decode_fused_kernel<NCG, NW>), dim3(grid), dim3(NW * 64), DF_LDS, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- scene fusion -----------------------------------------------------------------------------
// One thread per canvas pixel walks the batch's tiles IN ORDER, so the f32 summation order is the
// reference's sequential `canvas[y0:y1, x0:x1] += patch` order (deterministic; no atomics).
__global__ __launch_bounds__(256) void scene_add_kernel(const float* scores, int B, int P, const int* tile_xy,
                                                        float* kp, float* road, int S) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)S * S) return;
    const int x = gid % S, y = gid / S;
    float a = kp[gid], r = road[gid];
    bool touched = false;
    for (int t = 0; t < B; ++t) {
        const int x0 = tile_xy[2 * t], y0 = tile_xy[2 * t + 1];
        const int lx = x - x0, ly = y - y0;
        if (lx >= 0 && lx < P && ly >= 0 && ly < P) {
            const float2 v = *reinterpret_cast<const float2*>(scores + (((size_t)t * P + ly) * P + lx) * 2);
            a += v.x; r += v.y; touched = true;
        }
    }
    if (touched) { kp[gid] = a; road[gid] = r; }
}

__global__ __launch_bounds__(256) void scene_count_kernel(float* counter, int S, const int* tile_xy, int n, int P) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)S * S) return;
    const int x = gid % S, y = gid / S;
    float c = 0.f;
    for (int t = 0; t < n; ++t) {
        const int lx = x - tile_xy[2 * t], ly = y - tile_xy[2 * t + 1];
        if (lx >= 0 && lx < P && ly >= 0 && ly < P) c += 1.f;
    }
    counter[gid] = c;
}

__global__ __launch_bounds__(256) void scene_norm_kernel(SceneNormParams p) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= p.n) return;
    const float c = p.counter[gid];
    // (canvas / counter) * 255 -> uint8 truncation; 0/0 = NaN casts to 0 in the reference (App. D.2)
    const float a = (p.canvas_kp[gid] / c) * 255.f, r = (p.canvas_road[gid] / c) * 255.f;
    p.kp_u8[gid] = (c > 0.f) ? (uint8_t)a : (uint8_t)0;
    p.road_u8[gid] = (c > 0.f) ? (uint8_t)r : (uint8_t)0;
}

int launch_scene_add(const float* scores, int B, int P, const int* tile_xy, float* kp, float* road, int S, hipStream_t s) {
    const long n = (long)S * S;
    hipLaunchKernelGGL(scene_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scores, B, P, tile_xy, kp, road, S);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_scene_count(float* counter, int S, const int* tile_xy, int n_tiles, int P, hipStream_t s) {
    const long n = (long)S * S;
    hipLaunchKernelGGL(scene_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, counter, S, tile_xy, n_tiles, P);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_scene_normalise(const SceneNormParams& p, hipStream_t s) {
    hipLaunchKernelGGL(scene_norm_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

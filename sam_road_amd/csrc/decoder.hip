// Tail of the naive map decoder and the scene-level mask fusion.
//
// Reference: model.py:286-295 (ConvTranspose2d k2 s2 x4, LayerNorm2d, exact GELU), :445-446/:490-494
// (sigmoid, NCHW -> NHWC) and inferencer.py:79-110 (scatter-add of the two masks + coverage
// counter, divide, x255, truncate to u8).
//
// A stride-2 kernel-2 transposed conv does not overlap: every input pixel owns a 2x2 output
// block, so each layer is a per-pixel GEMM to 4*Cout columns (done by gemm.hip) and the rows of
// the successive activations are in quad-tree order (pixel, sub1, sub2, sub3).  This file holds the
// last layer (32 -> 2 channels, N = 8: too thin for MFMA), fused with the sigmoid and with the
// quad-tree -> row-major NHWC scatter; it is write-bound (16 B/px logits+scores).
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

// ---- the last TWO layers in one kernel ---------------------------------------------------------------------
// ConvT(64->32, k2 s2) + GELU + ConvT(32->2, k2 s2) + sigmoid + quad-tree -> NHWC scatter (model.py:292-295, :445-446).
// The layer-by-layer path wrote the 64 -> 4 x 32 activations (67 MB per 16 tiles) for decode_out_kernel to read back; here
// one wave keeps 16 level-2 rows in registers through both layers with the transposed MFMA chain of topo_fused.hip:
// U^T[sub3*32 + co, row] = W5 . X^T (v_mfma_f32_16x16x32_f16, A = W5 fragments held in registers for the wave's whole
// life, B = the rows straight from memory), bias + GELU in the C layout, and a pair of C tiles IS the B operand of the
// last layer if its weights are packed with the k permutation 8 g + j -> 16 (j >> 2) + 4 g + (j & 3).  The 8 x 32 f32 weights
// of the last layer enter as an fp16 hi + lo pair (two MFMAs), which keeps their f32 value to 2^-22.
// Per lane the output tile holds one 2-pixel x 2-class float4 of one output row: exactly decode_out_kernel's stores.
__global__ __launch_bounds__(256) void decode_tail_kernel(DecodeTailParams p) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const long ctiles = (long)p.B * p.S * p.S;                       // 16 level-2 rows each = one token's 4 x 4 block
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    f16x8 a5[8][2];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) a5[t][kb] = *reinterpret_cast<const f16x8*>(p.w5 + (size_t)(16 * t + n) * 64 + 32 * kb + 8 * g);
    f32x4 b5[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) b5[t] = *reinterpret_cast<const f32x4*>(p.b5 + 16 * t + 4 * g);
    f16x8 a7h, a7l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = 16 * (j >> 2) + 4 * g + (j & 3);
        const float v = n < 8 ? p.w7[n * 32 + ch] : 0.f;
        a7h[j] = (f16)v;
        a7l[j] = (f16)(v - (float)a7h[j]);
    }
    const float b7a = p.b7[0], b7b = p.b7[1];
    const int P = p.S * 16;
    // the next column tile's rows are fetched before the current tile's GELUs (two waves per SIMD do not hide a global load)
    f16x8 nx0 = f16x8{0, 0, 0, 0, 0, 0, 0, 0}, nx1 = nx0;
    if (wave0 < ctiles) {
        const f16* xr = p.x + ((size_t)wave0 * 16 + n) * 64 + 8 * g;
        nx0 = *reinterpret_cast<const f16x8*>(xr); nx1 = *reinterpret_cast<const f16x8*>(xr + 32);
    }
    for (long ct = wave0; ct < ctiles; ct += nwaves) {
        const f16x8 x0 = nx0, x1 = nx1;
        if (ct + nwaves < ctiles) {
            const f16* xr = p.x + ((size_t)(ct + nwaves) * 16 + n) * 64 + 8 * g;
            nx0 = *reinterpret_cast<const f16x8*>(xr); nx1 = *reinterpret_cast<const f16x8*>(xr + 32);
        }
        f32x4 u[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 a = mfma16(a5[t][0], x0, f32x4{0.f, 0.f, 0.f, 0.f});
            a = mfma16(a5[t][1], x1, a);
#pragma unroll
            for (int r = 0; r < 4; ++r) u[t][r] = gelu_fast(a[r] + b5[t][r]);
        }
        // the lane's level-2 row: (tile b, token py px, sub1, sub2) = ct * 16 + n
        long rr = ct;
        const int s2 = n & 3, s1 = n >> 2;
        const int px = (int)(rr % p.S); rr /= p.S;
        const int py = (int)(rr % p.S); rr /= p.S;
        const int b = (int)rr;
#pragma unroll
        for (int s3 = 0; s3 < 4; ++s3) {
            f16x8 ub;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ub[r] = (f16)u[2 * s3][r]; ub[4 + r] = (f16)u[2 * s3 + 1][r]; }
            f32x4 o = mfma16(a7h, ub, f32x4{0.f, 0.f, 0.f, 0.f});
            o = mfma16(a7l, ub, o);                                   // rows 4 g + r: g = ky, r = kx * 2 + class
            if (g < 2) {
                const int y = (((py * 2 + (s1 >> 1)) * 2 + (s2 >> 1)) * 2 + (s3 >> 1)) * 2 + g;
                const int xx = (((px * 2 + (s1 & 1)) * 2 + (s2 & 1)) * 2 + (s3 & 1)) * 2;
                const size_t off = (((size_t)b * P + y) * P + xx) * 2;
                const float4 lg = make_float4(o[0] + b7a, o[1] + b7b, o[2] + b7a, o[3] + b7b);
                if (p.logits) *reinterpret_cast<float4*>(p.logits + off) = lg;
                if (p.scores) {
                    // sigmoid as rcp(1 + exp2(-x log2 e)): v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~25-instruction
                    // expf + IEEE division sequence — the kernel is VALU-bound (32 GELUs + 16 sigmoids per lane and tile)
                    auto sg = [](float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); };
                    *reinterpret_cast<float4*>(p.scores + off) = make_float4(sg(lg.x), sg(lg.y), sg(lg.z), sg(lg.w));
                }
            }
        }
    }
}

int launch_decode_tail(const DecodeTailParams& p, hipStream_t s) {
    const long ctiles = (long)p.B * p.S * p.S;
    if (ctiles <= 0) return 0;
    const long want = (ctiles + 15) / 16;                                // >= 4 column tiles per wave when there is enough work
    const unsigned grid = (unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want);
    hipLaunchKernelGGL(decode_tail_kernel, dim3(grid), dim3(256), 0, s, p);
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

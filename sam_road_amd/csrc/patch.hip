// Tile batcher + pixel normalise + 16x16 patch im2col (SURVEY a2/a4, K1 + the gather half of K2).
//
// Reference: inferencer.py:52-58 (python crop + f32 stack + H2D), model.py:465-467
// ((x - mean) / std, NHWC -> NCHW) and the fork's PatchEmbed Conv2d(3, D, 16, 16).
// Here the crop, the normalisation and the im2col are one pass: each thread owns one patch row
// (16 px x 3 ch = 48 contiguous values of the channels-last source) and writes 48 contiguous
// fp16 of the GEMM A matrix, whose K axis is ordered (ky, kx, c) — the conv weight is permuted
// to match at pack time.  HBM-bound: 48 B (u8 scene) or 192 B (f32 tiles) in, 96 B out per thread.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

template <bool U8>
__global__ __launch_bounds__(256) void patch_im2col_kernel(PatchParams p) {
    const int S = p.P / 16;
    const long total = (long)p.B * S * S * 16;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ky = gid & 15;
    const long m = gid >> 4;
    const int px = m % S, py = (m / S) % S, b = m / ((long)S * S);
    long row_stride, base;
    if (p.scene_S > 0) {
        const int x0 = p.tile_xy[2 * b], y0 = p.tile_xy[2 * b + 1];
        row_stride = (long)p.scene_S * 3;
        base = (long)(y0 + py * 16 + ky) * row_stride + (long)(x0 + px * 16) * 3;
    } else {
        row_stride = (long)p.P * 3;
        base = (long)b * p.P * row_stride + (long)(py * 16 + ky) * row_stride + (long)px * 48;
    }
    const float mean[3] = {123.675f, 116.28f, 103.53f};
    const float stdv[3] = {58.395f, 57.12f, 57.375f};
    f16* out = p.out + m * 768 + ky * 48;
    float v[48];
    if (U8) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(p.src) + base;
#pragma unroll
        for (int i = 0; i < 48; ++i) v[i] = (float)src[i];
    } else {
        const float* src = reinterpret_cast<const float*>(p.src) + base;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(src + 4 * i);
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = i * 8 + e, c = k % 3;
            h[e] = (f16)((v[k] - mean[c]) / stdv[c]);
        }
        *reinterpret_cast<f16x8*>(out + i * 8) = h;
    }
}

int launch_patch_im2col(const PatchParams& p, hipStream_t s) {
    const int S = p.P / 16;
    const long total = (long)p.B * S * S * 16;
    if (total <= 0) return 0;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (p.src_is_u8) hipLaunchKernelGGL(patch_im2col_kernel<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(patch_im2col_kernel<false>, grid, block, 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

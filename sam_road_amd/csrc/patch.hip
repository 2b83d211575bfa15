// Tile batcher + pixel normalise + 16x16 patch im2col (SURVEY a2/a4, K1 + the gather half of K2).
//
// Reference: inferencer.py:52-58 (python crop + f32 stack + H2D), model.py:465-467
// ((x - mean) / std, NHWC -> NCHW) and the fork's PatchEmbed Conv2d(3, D, 16, 16).
// Here the crop, the normalisation and the im2col are one pass over the channels-last source into the
// GEMM A matrix, whose K axis is ordered (ky, kx, c) — the conv weight is permuted to match at pack
// time.  HBM-bound: 1 B (u8 scene) or 4 B (f32 tiles) in, 2 B out per value.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

// One thread = 4 consecutive values of one image row (the row of a tile is P*3 contiguous values: consecutive lanes read
// consecutive 16-byte (f32) / 4-byte (u8) pieces — fully coalesced; the first version gave each thread a whole 48-value
// patch row, i.e. 64 lanes read 64 different image rows per load instruction).  48 = 16 px * 3 is a multiple of 4, so a
// thread's 4 values never straddle two patches and land as one 8-byte store in the im2col row (k = ky*48 + kx*3 + c).
template <bool U8>
__global__ __launch_bounds__(256) void patch_im2col_kernel(PatchParams p) {
    const int S = p.P / 16, Q = p.P * 3 / 4;               // 4-value pieces per tile row
    const long total = (long)p.B * p.P * Q;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int q = (int)(gid % Q);
    const int y = (int)((gid / Q) % p.P);
    const int b = (int)(gid / ((long)Q * p.P));
    long src_off;
    if (p.scene_S > 0) {
        const int x0 = p.tile_xy[2 * b], y0 = p.tile_xy[2 * b + 1];
        src_off = ((long)(y0 + y) * p.scene_S + x0) * 3 + 4 * q;
    } else {
        src_off = ((long)b * p.P + y) * p.P * 3 + 4 * q;
    }
    float v[4];
    if (U8) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(p.src) + src_off;   // scene rows need not be 4-byte aligned
        v[0] = (float)src[0]; v[1] = (float)src[1]; v[2] = (float)src[2]; v[3] = (float)src[3];
    } else {
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.src) + src_off);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    // (x - mean) / std (model.py:465-467) as a multiply by the f32 reciprocal: within 1 ulp of the division before the
    // result is rounded to fp16 anyway
    const float mean[3] = {123.675f, 116.28f, 103.53f};
    const float rstd[3] = {1.0f / 58.395f, 1.0f / 57.12f, 1.0f / 57.375f};
    const int e0 = 4 * q;                                   // element index in the tile row: px*48 + kx*3 + c
    const int px = e0 / 48, within = e0 % 48;
    f16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = (within + e) % 3; h[e] = (f16)((v[e] - mean[c]) * rstd[c]); }
    const long m = ((long)b * S + (y >> 4)) * S + px;
    *reinterpret_cast<f16x4*>(p.out + m * 768 + (y & 15) * 48 + within) = h;
}

int launch_patch_im2col(const PatchParams& p, hipStream_t s) {
    const long total = (long)p.B * p.P * (p.P * 3 / 4);
    if (total <= 0) return 0;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (p.src_is_u8) hipLaunchKernelGGL(patch_im2col_kernel<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(patch_im2col_kernel<false>, grid, block, 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

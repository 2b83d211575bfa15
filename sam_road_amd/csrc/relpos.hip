// Decomposed relative-position bias terms of SAM attention (K5/K6 bias):
//   rel_h[q, kh] = q . Rh[qh, kh, :],  rel_w[q, kw] = q . Rw[qw, kw, :],  R(t)[i, j] = t[i - j + win - 1]
// computed from the UNSCALED q.  Output is pre-multiplied by 1/scale (8 for head dim 64) so the
// attention kernel can feed it to the MFMA as the accumulator's initial value and apply `scale`
// once in the softmax exponent:  scale * (q.k + rel/scale) = scale*q.k + rel.
//
// One wave per (image, window, head, direction, line): the 14/16/32 tokens of a window row
// (direction h) or column (direction w) share the same table slice, so the product is a
// [win x hd] x [hd x win] MFMA (v_mfma_f32_16x16x32_f16, transposed issue: lanes end up with 4
// consecutive kh of one token -> 16-byte f32 stores).  Pad positions are skipped (pad queries are
// never evaluated); lines entirely outside the image exit immediately.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

__global__ __launch_bounds__(256) void relpos_kernel(RelPosParams p) {
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int win = p.win, S = p.S;
    const int nw = (S + win - 1) / win;
    const int Wp = (win + 15) & ~15;
    const int total = p.B * nw * nw * p.heads * 2 * win;
    if (unit >= total) return;
    int u = unit;
    const int line = u % win; u /= win;
    const int dir = u & 1; u >>= 1;
    const int head = u % p.heads; u /= p.heads;
    const int wx = u % nw; u /= nw;
    const int wy = u % nw; u /= nw;
    const int b = u;
    // fixed coordinate of the line inside the image
    const int fixed = (dir == 0 ? wy : wx) * win + line;
    if (fixed >= S) return;
    const f16* table = dir == 0 ? p.table_h : p.table_w;
    const int g = lane >> 4, r16 = lane & 15;
    const int nks = (p.hd + 31) / 32;
    const int ntile = Wp / 16;
    for (int tt = 0; tt < ntile; ++tt) {           // token tile along the line
        const int j = tt * 16 + r16;               // position along the line
        const int run = (dir == 0 ? wx : wy) * win + j;
        const bool real = j < win && run < S;
        const int y = dir == 0 ? fixed : run, x = dir == 0 ? run : fixed;
        const size_t tok = ((size_t)b * S + (real ? y : 0)) * S + (real ? x : 0);
        f16x8 qf[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int c = ks * 32 + g * 8;
            if (real && c < p.hd) qf[ks] = *reinterpret_cast<const f16x8*>(p.qkv + tok * p.ld + head * p.hd + c);
            else for (int e = 0; e < 8; ++e) qf[ks][e] = (f16)0.f;
        }
        for (int kt = 0; kt < ntile; ++kt) {       // key-coordinate tile
            const int kh = kt * 16 + r16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                if (ks >= nks) break;
                const int c = ks * 32 + g * 8;
                f16x8 tf;
                if (kh < win && c < p.hd)
                    tf = *reinterpret_cast<const f16x8*>(table + (size_t)(line - kh + win - 1) * p.hd + c);
                else for (int e = 0; e < 8; ++e) tf[e] = (f16)0.f;
                acc = mfma16(tf, qf[ks], acc);
            }
            // D[kh_local = 4g + reg][token = r16]
            if (real) {
                float4 o;
                const int k0 = kt * 16 + 4 * g;
                o.x = (k0 + 0 < win) ? acc[0] * p.inv_scale : 0.f;
                o.y = (k0 + 1 < win) ? acc[1] * p.inv_scale : 0.f;
                o.z = (k0 + 2 < win) ? acc[2] * p.inv_scale : 0.f;
                o.w = (k0 + 3 < win) ? acc[3] * p.inv_scale : 0.f;
                *reinterpret_cast<float4*>(p.rel + (tok * p.heads + head) * (2 * Wp) + dir * Wp + k0) = o;
            }
        }
    }
}

int launch_relpos(const RelPosParams& p, hipStream_t s) {
    if (p.hd > 96 || p.hd % 8) return -2;
    const int nw = (p.S + p.win - 1) / p.win;
    const int total = p.B * nw * nw * p.heads * 2 * p.win;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(relpos_kernel, dim3((total + 3) / 4), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

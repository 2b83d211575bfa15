// Row LayerNorm / cast kernels (K3, LayerNorm2d of the neck and map_decoder, TopoNet post-LN).
// Channels-last activations make LayerNorm2d (per-pixel LN over channels, reference
// model.py:288 and the SAM neck) a plain row LN.  One wave per row, the row lives in registers,
// two-pass mean / biased variance in f32 (matches torch.nn.LayerNorm numerics).  HBM-bound:
// algorithmic bytes = 4*D read + 2*D (or 4*D) written per row.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

template <int EPL, int VW>  // elements per lane, vector width (floats)
__global__ __launch_bounds__(256) void layernorm_kernel(NormParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    constexpr int NV = EPL / VW;
    const float* x = p.x + (size_t)row * p.D;
    float v[EPL];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int off = (c * 64 + lane) * VW;
        if (VW == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + off);
            v[c * 4 + 0] = t.x; v[c * 4 + 1] = t.y; v[c * 4 + 2] = t.z; v[c * 4 + 3] = t.w;
        } else {
            const float2 t = *reinterpret_cast<const float2*>(x + off);
            v[c * 2 + 0] = t.x; v[c * 2 + 1] = t.y;
        }
    }
    if (p.gamma) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[e];
        const float mean = wave_sum(s) / (float)p.D;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { v[e] -= mean; q += v[e] * v[e]; }
        const float rstd = rsqrtf(wave_sum(q) / (float)p.D + p.eps);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int off = (c * 64 + lane) * VW;
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                float y = v[c * VW + e] * rstd * p.gamma[off + e] + p.beta[off + e];
                if (p.act == 1) y = gelu_erf(y);
                v[c * VW + e] = y;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int off = (c * 64 + lane) * VW;
        if (p.out_f16) {
            f16* o = p.out_f16 + (size_t)row * p.D + off;
            if (VW == 4) {
                f16x4 h = {(f16)v[c * 4], (f16)v[c * 4 + 1], (f16)v[c * 4 + 2], (f16)v[c * 4 + 3]};
                *reinterpret_cast<f16x4*>(o) = h;
            } else {
                f16x2 h = {(f16)v[c * 2], (f16)v[c * 2 + 1]};
                *reinterpret_cast<f16x2*>(o) = h;
            }
        }
        if (p.out_f32) {
            float* o = p.out_f32 + (size_t)row * p.D + off;
            if (VW == 4) *reinterpret_cast<float4*>(o) = make_float4(v[c * 4], v[c * 4 + 1], v[c * 4 + 2], v[c * 4 + 3]);
            else *reinterpret_cast<float2*>(o) = make_float2(v[c * 2], v[c * 2 + 1]);
        }
    }
}

int launch_layernorm(const NormParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    const dim3 grid((p.M + 3) / 4), block(256);
    switch (p.D) {
        case 128:  hipLaunchKernelGGL((layernorm_kernel<2, 2>), grid, block, 0, s, p); break;
        case 256:  hipLaunchKernelGGL((layernorm_kernel<4, 4>), grid, block, 0, s, p); break;
        case 768:  hipLaunchKernelGGL((layernorm_kernel<12, 4>), grid, block, 0, s, p); break;
        case 1024: hipLaunchKernelGGL((layernorm_kernel<16, 4>), grid, block, 0, s, p); break;
        case 1280: hipLaunchKernelGGL((layernorm_kernel<20, 4>), grid, block, 0, s, p); break;
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

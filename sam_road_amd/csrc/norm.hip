// Row LayerNorm / cast kernels (K3, LayerNorm2d of the neck and map_decoder, TopoNet post-LN).
// Channels-last activations make LayerNorm2d (per-pixel LN over channels, reference
// model.py:288 and the SAM neck) a plain row LN.  One wave per row, the row lives in registers,
// two-pass mean / biased variance in f32 (matches torch.nn.LayerNorm numerics).  HBM-bound:
// algorithmic bytes = 4*D read + 2*D (or 4*D) written per row.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

// R rows per wave per iteration (all R rows' loads are issued before any is consumed: the kernel is
// latency-bound otherwise — PMC showed >90 % of wave cycles parked in s_waitcnt with one row in flight),
// grid-stride over row groups.
template <int EPL, int VW, int R, int NT = 0, bool SL = false>  // elements per lane, vector width (floats), rows per wave per iteration, nontemporal x loads (2) / x_out stores (1), split-K slices folded
__global__ __launch_bounds__(256) void layernorm_kernel(NormParams p) {
    const int lane = threadIdx.x & 63;
    constexpr int NV = EPL / VW;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    // per-lane affine parameters are row-invariant: load once
    float g[EPL], be[EPL];
    if (p.gamma) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {   // vector loads: 24 scalar dword loads per row made the kernel VMEM-issue bound
            const int off = (c * 64 + lane) * VW;
            if (VW == 4) {
                const float4 tg = *reinterpret_cast<const float4*>(p.gamma + off), tb = *reinterpret_cast<const float4*>(p.beta + off);
                g[c * 4] = tg.x; g[c * 4 + 1] = tg.y; g[c * 4 + 2] = tg.z; g[c * 4 + 3] = tg.w;
                be[c * 4] = tb.x; be[c * 4 + 1] = tb.y; be[c * 4 + 2] = tb.z; be[c * 4 + 3] = tb.w;
            } else {
                const float2 tg = *reinterpret_cast<const float2*>(p.gamma + off), tb = *reinterpret_cast<const float2*>(p.beta + off);
                g[c * 2] = tg.x; g[c * 2 + 1] = tg.y; be[c * 2] = tb.x; be[c * 2 + 1] = tb.y;
            }
        }
    }
    const float inv_d = 1.0f / (float)p.D;   // once per wave: an IEEE division per row statistic costs ~20 VALU instructions
    for (int row0 = wave_global * R; row0 < p.M; row0 += n_waves * R) {
        float v[R][EPL];
        f16 dl[R][EPL];                                   // optional fp16 residual-branch output to fold in (dead when unused)
        f16 dl2[R][EPL];                                  // optional second one (delta16b)
        float sl[SL ? R : 1][SL ? EPL : 1];               // optional split-K partials of the preceding GEMM, summed in slice order
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = min(row0 + r, p.M - 1);
            const float* x = p.x + (size_t)(p.x_period ? row % p.x_period : row) * p.D;
            if (p.delta16) {
                const f16* d = p.delta16 + (size_t)row * p.D;
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    const int off = (c * 64 + lane) * VW;
                    if (VW == 4) {
                        const f16x4 t = *reinterpret_cast<const f16x4*>(d + off);
                        dl[r][c * 4 + 0] = t[0]; dl[r][c * 4 + 1] = t[1]; dl[r][c * 4 + 2] = t[2]; dl[r][c * 4 + 3] = t[3];
                    } else {
                        const f16x2 t = *reinterpret_cast<const f16x2*>(d + off);
                        dl[r][c * 2 + 0] = t[0]; dl[r][c * 2 + 1] = t[1];
                    }
                }
                if (p.delta16b) {
                    const f16* d2 = p.delta16b + (size_t)row * p.D;
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const int off = (c * 64 + lane) * VW;
                        if (VW == 4) {
                            const f16x4 t = *reinterpret_cast<const f16x4*>(d2 + off);
                            dl2[r][c * 4 + 0] = t[0]; dl2[r][c * 4 + 1] = t[1]; dl2[r][c * 4 + 2] = t[2]; dl2[r][c * 4 + 3] = t[3];
                        } else {
                            const f16x2 t = *reinterpret_cast<const f16x2*>(d2 + off);
                            dl2[r][c * 2 + 0] = t[0]; dl2[r][c * 2 + 1] = t[1];
                        }
                    }
                }
            }
            if constexpr (SL && VW == 4) {
                {
                    const float* sp = p.slices + (size_t)row * p.D;
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const float4 t = *reinterpret_cast<const float4*>(sp + (c * 64 + lane) * 4);
                        sl[r][c * 4] = t.x; sl[r][c * 4 + 1] = t.y; sl[r][c * 4 + 2] = t.z; sl[r][c * 4 + 3] = t.w;
                    }
                    for (int z = 1; z < p.nslices; ++z) {
                        sp += p.slice_stride;
#pragma unroll
                        for (int c = 0; c < NV; ++c) {
                            const float4 t = *reinterpret_cast<const float4*>(sp + (c * 64 + lane) * 4);
                            sl[r][c * 4] += t.x; sl[r][c * 4 + 1] += t.y; sl[r][c * 4 + 2] += t.z; sl[r][c * 4 + 3] += t.w;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const float4 t = *reinterpret_cast<const float4*>(p.slice_bias + (c * 64 + lane) * 4);
                        sl[r][c * 4] += t.x; sl[r][c * 4 + 1] += t.y; sl[r][c * 4 + 2] += t.z; sl[r][c * 4 + 3] += t.w;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int off = (c * 64 + lane) * VW;
                if (VW == 4) {
                    float4 t;
                    if (NT & 2) { const f32x4 n4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + off)); t = make_float4(n4[0], n4[1], n4[2], n4[3]); }
                    else t = *reinterpret_cast<const float4*>(x + off);
                    v[r][c * 4 + 0] = t.x; v[r][c * 4 + 1] = t.y; v[r][c * 4 + 2] = t.z; v[r][c * 4 + 3] = t.w;
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(x + off);
                    v[r][c * 2 + 0] = t.x; v[r][c * 2 + 1] = t.y;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (SL || p.delta16) {                        // x <- x + delta (the residual add the GEMM epilogue no longer does)
                if constexpr (SL) {                       // (sum of the K slices + bias) + x: splitk_reduce_kernel's order
#pragma unroll
                    for (int e = 0; e < EPL; ++e) v[r][e] = sl[r][e] + v[r][e];
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) v[r][e] += (float)dl[r][e];
                }
                if (p.delta16b) {                         // (x + first) + second: the order in which the reference adds its two branches
#pragma unroll
                    for (int e = 0; e < EPL; ++e) v[r][e] += (float)dl2[r][e];
                }
                if (p.x_out && row < p.M) {
                    float* xo = p.x_out + (size_t)row * p.D;
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const int off = (c * 64 + lane) * VW;
                        if (VW == 4) { const float4 t4 = make_float4(v[r][c * 4], v[r][c * 4 + 1], v[r][c * 4 + 2], v[r][c * 4 + 3]); if (NT & 1) { const f32x4 n4 = {t4.x, t4.y, t4.z, t4.w}; __builtin_nontemporal_store(n4, reinterpret_cast<f32x4*>(xo + off)); } else *reinterpret_cast<float4*>(xo + off) = t4; }
                        else *reinterpret_cast<float2*>(xo + off) = make_float2(v[r][c * 2], v[r][c * 2 + 1]);
                    }
                }
            }
            if (p.gamma) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) s += v[r][e];
                const float mean = wave_sum(s) * inv_d;
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) { v[r][e] -= mean; q += v[r][e] * v[r][e]; }
                const float var = wave_sum(q) * inv_d;
                if (p.nf && !(var < INFINITY) && lane == 0) p.nf[p.nf_tag] = 1u;     // Inf / NaN among the row's inputs (NormParams::nf)
                const float rstd = rsqrtf(var + p.eps);
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    float y = v[r][e] * rstd * g[e] + be[e];
                    if (p.act == 1) y = gelu_fast(y);     // exact-erf GELU to 1.6e-6 (common.hpp); erff made the LayerNorm2d + GELU pass VALU-bound
                    v[r][e] = y;
                }
            }
            if (row >= p.M) continue;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int off = (c * 64 + lane) * VW;
                if (p.out_f16) {
                    f16* o = p.out_f16 + (size_t)row * p.D + off;
                    if (VW == 4) {
                        f16x4 h = {(f16)v[r][c * 4], (f16)v[r][c * 4 + 1], (f16)v[r][c * 4 + 2], (f16)v[r][c * 4 + 3]};
                        *reinterpret_cast<f16x4*>(o) = h;
                    } else {
                        f16x2 h = {(f16)v[r][c * 2], (f16)v[r][c * 2 + 1]};
                        *reinterpret_cast<f16x2*>(o) = h;
                    }
                }
                if (p.out_f32) {
                    float* o = p.out_f32 + (size_t)row * p.D + off;
                    if (VW == 4) *reinterpret_cast<float4*>(o) = make_float4(v[r][c * 4], v[r][c * 4 + 1], v[r][c * 4 + 2], v[r][c * 4 + 3]);
                    else *reinterpret_cast<float2*>(o) = make_float2(v[r][c * 2], v[r][c * 2 + 1]);
                }
            }
        }
    }
}

template <int EPL, int VW, int R, int NT = 0, bool SL = false>
static void launch_ln(const NormParams& p, hipStream_t s) {
    const int groups = (p.M + R - 1) / R;                 // wave-iterations needed
    const int blocks = min((groups + 3) / 4, 256 * 8);    // <= 8 blocks per CU, grid-stride beyond
    hipLaunchKernelGGL((layernorm_kernel<EPL, VW, R, NT, SL>), dim3(blocks), dim3(256), 0, s, p);
}

int launch_layernorm(const NormParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (p.slices) {      // the small-M models' pass after a split-K fc2 (api.hip): its own instantiations, the others stay as they were
        if (p.delta16 || p.delta16b || p.nslices < 1 || !p.slice_bias) return -2;
        switch (p.D) {
            case 1024: launch_ln<16, 4, 1, 0, true>(p, s); break;
            case 1280: launch_ln<20, 4, 1, 0, true>(p, s); break;
            default: return -2;
        }
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    switch (p.D) {
        case 128:  launch_ln<2, 2, 4>(p, s); break;
        case 256:  launch_ln<4, 4, 4>(p, s); break;
        // ViT-B block LayerNorms: one row per wave and iteration, the f32 residual stream read and written with nontemporal hints (it is
        // touched once per pass; keeping it out of L2 leaves qkv / K / V there for the attention kernels): +0.7 % tiles/s over <12, 4, 2>
        // with cached accesses, same box, alternating (profiles/r04_layernorm_nt.txt).  Bit-identical output.
        case 768:  launch_ln<12, 4, 1, 3>(p, s); break;
        case 1024: launch_ln<16, 4, 1>(p, s); break;
        case 1280: launch_ln<20, 4, 1>(p, s); break;
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

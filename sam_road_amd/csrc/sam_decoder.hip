// SAM MaskDecoder branch of SAMRoad (`USE_SAM_DECODER: True`; reference model.py:260-282 construction, :426-443 / :471-488 use):
// PromptEncoder no-prompt path -> TwoWayTransformer(depth 2, 256 wide, 8 heads, mlp 2048) over 4 output tokens (IoU token +
// 3 mask tokens) and the S x S image tokens -> 2 x ConvTranspose upscaling -> hyper-network dot products -> masks 1, 2 of the 3
// -> bilinear x4 upsample to the tile -> sigmoid.  The fork's source is absent; semantics follow upstream
// segment-anything as restated and cross-checked against transformers in oracle/sam_decoder.py.
//
// Split of the work (per batch of B tiles; HW = S*S image tokens per tile, T = B*HW):
//   * image side — every [T,256] x [256 -> 128 / 256] projection is an MFMA GEMM (gemm.hip) on the fp16 copy of the keys; the
//     positional term is linear, (keys + pe) W = keys W + (pe W): pe W is computed once at pack time and enters as the GEMM's
//     per-row `pos` addend, so no "keys + pe" pass exists; residual + LayerNorm reuse the encoder's kernels (norm.hip);
//   * token side — 4 tokens per tile: f32 VALU kernels below (a few MFLOP per tile: latency, not throughput);
//   * the two attentions are tiny per head (4 x HW scores): one workgroup per (tile, head) with the scores in LDS
//     (token -> image), one thread per (image token, head) over the 4 token keys (image -> token).
// Archived configs only: built for parity, not tuned.
#include <atomic>

#include "common.hpp"
#include "kernels.hpp"

namespace srh {

// keys = emb + no_mask_embed (dense prompt embedding, model.py:427-429 / fork mask_decoder.py `src = src + dense`)
__global__ __launch_bounds__(256) void sd_add_channel_kernel(const float* emb, const float* vec, float* out_f32, f16* out_f16, size_t n4) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= n4) return;
    const float4 e = reinterpret_cast<const float4*>(emb)[i];
    const float4 v = reinterpret_cast<const float4*>(vec)[i & 63];            // 256 channels = 64 float4
    const float4 r = make_float4(e.x + v.x, e.y + v.y, e.z + v.z, e.w + v.w);
    reinterpret_cast<float4*>(out_f32)[i] = r;
    const f16x4 h = {(f16)r.x, (f16)r.y, (f16)r.z, (f16)r.w};
    reinterpret_cast<f16x4*>(out_f16)[i] = h;
}

// y[r, n] = act( (x[r,:] + xadd[r % add_rows,:]) . W[n,:] + b[n] ) ; one wave per output, lanes split K
__global__ __launch_bounds__(256) void sd_tok_linear_kernel(SdLinearParams p) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6), r = blockIdx.y;
    if (n >= p.N) return;
    const float* x = p.x + (size_t)r * p.ldx;
    const float* xa = p.xadd ? p.xadd + (size_t)(r % p.add_rows) * p.K : nullptr;
    const float* w = p.W + (size_t)n * p.K;
    float s = 0.f;
    for (int k = lane; k < p.K; k += 64) s += (x[k] + (xa ? xa[k] : 0.f)) * w[k];
    s = wave_sum(s);
    if (lane == 0) {
        s += p.b ? p.b[n] : 0.f;
        if (p.act == 2) s = fmaxf(s, 0.f);
        p.y[(size_t)r * p.ldy + n] = s;
    }
}

// y[r,:] = LayerNorm(x[r,:] (+ resid[r,:])) over 256 channels, eps 1e-5 (nn.LayerNorm default of the fork's transformer)
__global__ __launch_bounds__(256) void sd_tok_ln_kernel(const float* x, const float* resid, const float* g, const float* b, float* y, int rows) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x[(size_t)r * 256 + e * 64 + lane] + (resid ? resid[(size_t)r * 256 + e * 64 + lane] : 0.f);
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] -= mean; q += v[e] * v[e]; }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / 256.f) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[(size_t)r * 256 + e * 64 + lane] = v[e] * rstd * g[e * 64 + lane] + b[e * 64 + lane];
}

// self attention among the 4 tokens of a tile: q, k, v [B*4, 256] (8 heads x 32) -> out [B*4, 256]
__global__ __launch_bounds__(256) void sd_tok_selfattn_kernel(const float* q, const float* k, const float* v, float* out) {
    __shared__ float sq[4][256], sk[4][256], sv[4][256];
    const int b = blockIdx.x, tid = threadIdx.x;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        sq[t][tid] = q[(size_t)(b * 4 + t) * 256 + tid];
        sk[t][tid] = k[(size_t)(b * 4 + t) * 256 + tid];
        sv[t][tid] = v[(size_t)(b * 4 + t) * 256 + tid];
    }
    __syncthreads();
    const int h0 = (tid >> 5) * 32;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
            for (int d = 0; d < 32; ++d) a += sq[t][h0 + d] * sk[j][h0 + d];
            s[j] = a * 0.17677669529663687f;                 // 1 / sqrt(32)
        }
        const float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        float e[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = __expf(s[j] - m); sum += e[j]; }
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o += e[j] * sv[j][tid];
        out[(size_t)(b * 4 + t) * 256 + tid] = o / sum;
    }
}

// token -> image attention: q [B*4, 128] f32 (8 heads x 16), K / V [B*HW, 128] fp16 -> out [B*4, 128] f32.
// One workgroup per (tile, head); scores of the 4 tokens against all HW keys live in LDS.
__global__ __launch_bounds__(256) void sd_t2i_attn_kernel(const float* q, const f16* K, const f16* V, float* out, int HW) {
    extern __shared__ float sc[];                             // [4][HW] + reduction scratch [4][4] + [4][16][4]
    float* red = sc + 4 * HW;
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float qv[4][16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < 16; ++d) qv[t][d] = q[(size_t)(b * 4 + t) * 128 + h * 16 + d] * 0.25f;      // 1 / sqrt(16)
    float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int key = tid; key < HW; key += 256) {
        const f16x8* kp = reinterpret_cast<const f16x8*>(K + ((size_t)b * HW + key) * 128 + h * 16);
        const f16x8 k0 = kp[0], k1 = kp[1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) a += qv[t][d] * (float)k0[d] + qv[t][8 + d] * (float)k1[d];
            sc[t * HW + key] = a;
            mx[t] = fmaxf(mx[t], a);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], o, 64));
        if (lane == 0) red[t * 4 + wave] = mx[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) mx[t] = fmaxf(fmaxf(red[t * 4], red[t * 4 + 1]), fmaxf(red[t * 4 + 2], red[t * 4 + 3]));
    // out[t][d] = sum_key exp(s - max) v[key][d] / sum_key exp(s - max): every thread accumulates its keys
    float acc[4][16], den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[t][d] = 0.f;
    for (int key = tid; key < HW; key += 256) {
        const f16x8* vp = reinterpret_cast<const f16x8*>(V + ((size_t)b * HW + key) * 128 + h * 16);
        const f16x8 v0 = vp[0], v1 = vp[1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float e = __expf(sc[t * HW + key] - mx[t]);
            den[t] += e;
#pragma unroll
            for (int d = 0; d < 8; ++d) { acc[t][d] += e * (float)v0[d]; acc[t][8 + d] += e * (float)v1[d]; }
        }
    }
    __syncthreads();                                         // everyone is done with red[] (max phase)
    float* racc = red + 16;                                  // [4 waves][4 tokens][17]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        den[t] = wave_sum(den[t]);
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[t][d] = wave_sum(acc[t][d]);
        if (lane == 0) {
            racc[(wave * 4 + t) * 17 + 16] = den[t];
#pragma unroll
            for (int d = 0; d < 16; ++d) racc[(wave * 4 + t) * 17 + d] = acc[t][d];
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int t = tid >> 4, d = tid & 15;
        float a = 0.f, dn = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += racc[(w * 4 + t) * 17 + d]; dn += racc[(w * 4 + t) * 17 + 16]; }
        out[(size_t)(b * 4 + t) * 128 + h * 16 + d] = a / dn;
    }
}

// image -> token attention: Q [B*HW, 128] fp16 (8 heads x 16), token k, v [B*4, 128] f32 -> out [B*HW, 128] fp16
__global__ __launch_bounds__(256) void sd_i2t_attn_kernel(const f16* Q, const float* k, const float* v, f16* out, int HW) {
    __shared__ float sk[4][128], sv[4][128];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 512; i += 256) { sk[i >> 7][i & 127] = k[(size_t)b * 512 + i]; sv[i >> 7][i & 127] = v[(size_t)b * 512 + i]; }
    __syncthreads();
    const int item = blockIdx.x * 256 + tid;                 // (image token, head)
    if (item >= HW * 8) return;
    const int tok = item >> 3, h = item & 7;
    const f16x8* qp = reinterpret_cast<const f16x8*>(Q + ((size_t)b * HW + tok) * 128 + h * 16);
    const f16x8 q0 = qp[0], q1 = qp[1];
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) a += (float)q0[d] * sk[j][h * 16 + d] + (float)q1[d] * sk[j][h * 16 + 8 + d];
        s[j] = a * 0.25f;
    }
    const float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    float e[4], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { e[j] = __expf(s[j] - m); sum += e[j]; }
    const float inv = 1.f / sum;
    f16x8 o0, o1;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0 += e[j] * sv[j][h * 16 + d]; a1 += e[j] * sv[j][h * 16 + 8 + d]; }
        o0[d] = (f16)(a0 * inv); o1[d] = (f16)(a1 * inv);
    }
    f16x8* op = reinterpret_cast<f16x8*>(out + ((size_t)b * HW + tok) * 128 + h * 16);
    op[0] = o0; op[1] = o1;
}

// LayerNorm2d(64) + GELU on rows of 64 channels (output_upscaling.1 / .2 of the fork's MaskDecoder): one wave per row
__global__ __launch_bounds__(256) void sd_ln64_gelu_kernel(const float* x, const float* g, const float* b, f16* y, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t r = blockIdx.x * (size_t)4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float v = x[r * 64 + lane];
    const float mean = wave_sum(v) * (1.f / 64.f);
    v -= mean;
    const float rstd = rsqrtf(wave_sum(v * v) * (1.f / 64.f) + 1e-6f);
    y[r * 64 + lane] = (f16)gelu_fast(v * rstd * g[lane] + b[lane]);
}

// low-res mask logits: low[b, c, Y, X] = hyper[b, c+1, :] . up[row(b, Y, X), :]   (c = 0, 1: mask tokens 1, 2 — multimask_output)
// up rows are in quad-tree order: row = ((b*HW + y*S + x)*4 + ky1*2 + kx1)*4 + ky2*2 + kx2, pixel (4y + 2ky1 + ky2, 4x + 2kx1 + kx2)
__global__ __launch_bounds__(256) void sd_mask_kernel(const f16* up, const float* hyper, float* low, int B, int S) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;          // one thread per low-res pixel
    const int L = 4 * S;
    if (i >= (size_t)B * L * L) return;
    const int b = (int)(i / ((size_t)L * L)), rem = (int)(i % ((size_t)L * L)), Y = rem / L, X = rem % L;
    const int y = Y >> 2, x = X >> 2, s1 = ((Y >> 1) & 1) * 2 + ((X >> 1) & 1), s2 = (Y & 1) * 2 + (X & 1);
    const size_t row = (((size_t)b * S * S + (size_t)y * S + x) * 4 + s1) * 4 + s2;
    const f16x8* u = reinterpret_cast<const f16x8*>(up + row * 32);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        const f16x8 t = u[c8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a0 += (float)t[e] * hyper[(size_t)(b * 3 + 1) * 32 + c8 * 8 + e];
            a1 += (float)t[e] * hyper[(size_t)(b * 3 + 2) * 32 + c8 * 8 + e];
        }
    }
    low[((size_t)b * 2 + 0) * L * L + rem] = a0;
    low[((size_t)b * 2 + 1) * L * L + rem] = a1;
}

// F.interpolate(low [B,2,L,L], (P,P), mode="bilinear", align_corners=False) -> logits / sigmoid scores [B,P,P,2] (model.py:437-443)
__global__ __launch_bounds__(256) void sd_upsample_kernel(const float* low, float* logits, float* scores, int B, int L, int P) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;          // one thread per output pixel
    if (i >= (size_t)B * P * P) return;
    const int b = (int)(i / ((size_t)P * P)), rem = (int)(i % ((size_t)P * P)), Y = rem / P, X = rem % P;
    const float scale = (float)L / (float)P;
    const float sy = fmaxf(((float)Y + 0.5f) * scale - 0.5f, 0.f), sx = fmaxf(((float)X + 0.5f) * scale - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx, y1 = min(y0 + 1, L - 1), x1 = min(x0 + 1, L - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    float r[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* pl = low + ((size_t)b * 2 + c) * L * L;
        const float top = pl[y0 * L + x0] * (1.f - lx) + pl[y0 * L + x1] * lx;
        const float bot = pl[y1 * L + x0] * (1.f - lx) + pl[y1 * L + x1] * lx;
        r[c] = top * (1.f - ly) + bot * ly;
    }
    if (logits) reinterpret_cast<float2*>(logits)[i] = make_float2(r[0], r[1]);
    if (scores) reinterpret_cast<float2*>(scores)[i] = make_float2(sigmoidf_(r[0]), sigmoidf_(r[1]));
}

#define SD_OK() (hipGetLastError() == hipSuccess ? 0 : -3)

int launch_sd_add_channel(const float* emb, const float* vec, float* out_f32, f16* out_f16, size_t rows, hipStream_t s) {
    const size_t n4 = rows * 64;
    hipLaunchKernelGGL(sd_add_channel_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, emb, vec, out_f32, out_f16, n4);
    return SD_OK();
}
int launch_sd_tok_linear(const SdLinearParams& p, hipStream_t s) {
    if (p.rows <= 0) return 0;
    hipLaunchKernelGGL(sd_tok_linear_kernel, dim3((p.N + 3) / 4, p.rows), dim3(256), 0, s, p);
    return SD_OK();
}
int launch_sd_tok_ln(const float* x, const float* resid, const float* g, const float* b, float* y, int rows, hipStream_t s) {
    hipLaunchKernelGGL(sd_tok_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, resid, g, b, y, rows);
    return SD_OK();
}
int launch_sd_tok_selfattn(const float* q, const float* k, const float* v, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(sd_tok_selfattn_kernel, dim3(B), dim3(256), 0, s, q, k, v, out);
    return SD_OK();
}
int launch_sd_t2i_attn(const float* q, const f16* K, const f16* V, float* out, int B, int HW, hipStream_t s) {
    const size_t lds = ((size_t)4 * HW + 16 + 4 * 4 * 17) * 4;
    static OncePerDevice opt_in;      // the attribute is per device (a process may drive several GPUs, one context each)
    if (!opt_in.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(sd_t2i_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess; }))
        return -3;
    if (lds > 80 * 1024) return -2;
    hipLaunchKernelGGL(sd_t2i_attn_kernel, dim3(B, 8), dim3(256), lds, s, q, K, V, out, HW);
    return SD_OK();
}
int launch_sd_i2t_attn(const f16* Q, const float* k, const float* v, f16* out, int B, int HW, hipStream_t s) {
    hipLaunchKernelGGL(sd_i2t_attn_kernel, dim3((HW * 8 + 255) / 256, B), dim3(256), 0, s, Q, k, v, out, HW);
    return SD_OK();
}
int launch_sd_ln64_gelu(const float* x, const float* g, const float* b, f16* y, size_t rows, hipStream_t s) {
    hipLaunchKernelGGL(sd_ln64_gelu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, g, b, y, rows);
    return SD_OK();
}
int launch_sd_mask(const f16* up, const float* hyper, float* low, int B, int S, hipStream_t s) {
    const size_t n = (size_t)B * 16 * S * S;
    hipLaunchKernelGGL(sd_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, up, hyper, low, B, S);
    return SD_OK();
}
int launch_sd_upsample(const float* low, float* logits, float* scores, int B, int L, int P, hipStream_t s) {
    const size_t n = (size_t)B * P * P;
    hipLaunchKernelGGL(sd_upsample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, low, logits, scores, B, L, P);
    return SD_OK();
}

}  // namespace srh

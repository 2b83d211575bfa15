// Fused TopoNet trunk: pair_proj + the three post-LN nn.TransformerEncoderLayer's + output_proj + sigmoid in ONE kernel.
// Replaces reference model.py:118-148 (TopoNet.forward after the feature gather) — 23 launches of the layer-by-layer path
// (8 small GEMMs with K = 128, 3 attentions, 6 LayerNorms, ...) whose activations bounced through HBM between every pair
// of them: that path ran the 45 GFLOP of a 16-tile batch at ~60 TFLOP/s.
//
// gfx950 design.  One wave owns one 16-token sequence (the K = 16 candidate pairs of one source point) for the whole trunk
// and keeps it in registers; nothing but the weights moves.
//   * Everything is computed TRANSPOSED, Y^T[feature, token] = W[feature, :] . X^T[:, token], with v_mfma_f32_16x16x32_f16:
//     A = a 16x32 weight fragment, B = the activations.  In the C/D layout lane l holds token (l & 15) and features
//     16 t + 4 (l >> 4) + r of row tile t; the B operand of the NEXT MFMA wants, per lane, 8 k-values of a 32-wide k block.
//     Two consecutive row tiles give exactly 8 values per lane, i.e. a C tile pair IS a B operand if k index 8 g + j is read
//     as feature 32 kb + 16 (j >> 2) + 4 g + (j & 3).  That permutation is baked into the packed weights (api.hip,
//     pack_topo_fused), so GEMM -> GEMM chains need no LDS round trip, no shuffles: cvt_pk only.
//   * For one sequence the same register image is also the A operand "rows = tokens": V is produced token-major,
//     V[token, d] = X . Wv^T, by swapping the operands of the same MFMA with the same packed fragments.
//   * Attention per head (head dim 32 = one k block): S^T[key, query] = one 16x16x32 MFMA of the K^T and Q^T tile pairs;
//     softmax over keys is 4 lane-local values + two cross-lane exchanges; P^T in C layout is the B operand of the
//     16x16x16 MFMA with A = V (token-major tile read as A[d, key]) giving O^T, again a B operand for out_proj.
//   * Residual stream and LayerNorm in f32 registers (32 per lane); LN statistics: lane-local + xor-16 / xor-32 exchanges.
//   * Weights: 656 packed 1 KiB fragments (656 KiB fp16) streamed from L2 through a 4 x 16 KiB LDS ring by LDS-DMA
//     (buffer_load ... lds), one barrier per 16-fragment chunk, counted vmcnt (each wave issues one 1 KiB piece per chunk).
//     A workgroup is TF_NW waves = sequences sharing the ring, one workgroup per CU.  Every wave reads every fragment,
//     so the kernel is LDS-read bound (TF_NW x 16 KiB per chunk at 128 B/clk against 272 clk of MFMA per wave).
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

typedef __attribute__((address_space(3))) void* lds_vptr;

constexpr int TF_CHUNK = 16384, TF_NBUF = 4, TF_RING = TF_CHUNK * TF_NBUF;
constexpr int TF_NW = 8;                      // waves (= sequences) per workgroup; each issues 16 / TF_NW DMA pieces per chunk

__device__ __forceinline__ f32x4 mfma16k16(f16x4 a, f16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16x8 tf_pack8(const f32x4& a, const f32x4& b) {
    f16x8 r;
    r[0] = (f16)a[0]; r[1] = (f16)a[1]; r[2] = (f16)a[2]; r[3] = (f16)a[3];
    r[4] = (f16)b[0]; r[5] = (f16)b[1]; r[6] = (f16)b[2]; r[7] = (f16)b[3];
    return r;
}
__device__ __forceinline__ f16x4 tf_pack4(const f32x4& a) {
    f16x4 r;
    r[0] = (f16)a[0]; r[1] = (f16)a[1]; r[2] = (f16)a[2]; r[3] = (f16)a[3];
    return r;
}
// sum / max over the four lanes (l & 15) + 16 g that hold one token
__device__ __forceinline__ float tf_sum4(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float tf_max4(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// post-LN: x = LayerNorm(y) * gamma + beta over the 128 features of the lane's token (eps 1e-5, biased variance)
__device__ __forceinline__ void tf_layernorm(f32x4 (&x)[8], const f32x4 (&y)[8], const float* gam, const float* bet, int g) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += (y[t][0] + y[t][1]) + (y[t][2] + y[t][3]);
    const float mean = tf_sum4(s) * (1.0f / 128.0f);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = y[t][r] - mean; v = fmaf(d, d, v); }
    const float rstd = rsqrtf(tf_sum4(v) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + 16 * t + 4 * g);
        const f32x4 be = *reinterpret_cast<const f32x4*>(bet + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] = fmaf((y[t][r] - mean) * rstd, ga[r], be[r]);
    }
}

template <int NL>
__global__ __launch_bounds__(TF_NW * 64) void topo_fused_kernel(TopoFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const prm = reinterpret_cast<float*>(smem + TF_RING);
    constexpr int NCH = 5 + 12 * NL;
    constexpr int NPRM = 128 + 1280 * NL + 132;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    for (int i = tid; i < NPRM; i += TF_NW * 64) prm[i] = p.params[i];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.stream, 0, 0x7fffffff, 0x00020000);

    // fragment fi of the stream (compile-time index): ring buffer (fi >> 4) & 3, 1 KiB slot fi & 15, 16 B per lane
#define TF_FR(fi) (*reinterpret_cast<const f16x8*>(ring + (((fi) >> 4) & (TF_NBUF - 1)) * TF_CHUNK + ((fi) & 15) * 1024 + lane * 16))
#define TF_DMA(ci) { _Pragma("unroll") for (int pc_ = 0; pc_ < 16 / TF_NW; ++pc_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(ring + ((ci) & (TF_NBUF - 1)) * TF_CHUNK + (wave + TF_NW * pc_) * 1024), 16, \
                                                 lane * 16, (ci) * TF_CHUNK + (wave + TF_NW * pc_) * 1024, 0, 0); }
    // before the MFMAs of chunk ci: this wave's piece of chunk ci has landed (the pieces of the <= 2 younger chunks may
    // still fly), its own reads of the previous chunk have returned, then every wave's piece is visible; the buffer of
    // chunk ci - 1 is free after the barrier and receives chunk ci + 3
#define TF_STEP(ci) { \
        static_assert(TF_NW == 8 || TF_NW == 16, "vmcnt immediates below"); \
        if ((ci) + 2 < NCH) { if (TF_NW == 8) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); } \
        else if ((ci) + 1 < NCH) { if (TF_NW == 8) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); } \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
        if ((ci) + 3 < NCH) { TF_DMA((ci) + 3) } }
#define TF_B4(off) (*reinterpret_cast<const f32x4*>(prm + (off)))

    const float c_exp = 0.17677669529663687f * 1.4426950408889634f;      // 32^-0.5 * log2(e)
    for (int sg = blockIdx.x; sg * TF_NW < p.nseq; sg += gridDim.x) {
        const int seq_raw = sg * TF_NW + wave;
        const bool live = seq_raw < p.nseq;
        const int seq = live ? seq_raw : p.nseq - 1;
        // the sequence's 16 pair rows as B operands (natural k order: pair_proj's fragments are packed to match)
        const f16* row = p.pair + ((size_t)seq * 16 + n) * p.ld_pair + 8 * g;
        f16x8 xin[10];
#pragma unroll
        for (int kb = 0; kb < 10; ++kb) xin[kb] = *reinterpret_cast<const f16x8*>(row + 32 * kb);
        uint32_t vbits = *reinterpret_cast<const uint32_t*>(p.valid + (size_t)seq * 16 + 4 * g);   // keys 4g .. 4g+3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // params visible / the previous pass has released the ring
        TF_DMA(0) TF_DMA(1) TF_DMA(2)
        // model.py:129-130: a sequence without any valid key attends to every key
        {
            int any = vbits != 0;
            any |= __shfl_xor(any, 16, 64);
            any |= __shfl_xor(any, 32, 64);
            if (!any) vbits = 0x01010101u;
        }

        f32x4 xs[8];        // residual stream, C layout: token n, features 16 t + 4 g + r
        f16x8 xp[4];        // the same as MFMA operand (k block kb = row tiles 2 kb, 2 kb + 1)
        // ---- pair_proj (K = 320 = 10 k blocks, fragments k-major) + ReLU
        {
            f32x4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                TF_STEP(j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] = mfma16(TF_FR(16 * j + 8 * kk + t), xin[2 * j + kk], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 b = TF_B4(16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[t][r] = fmaxf(acc[t][r] + b[r], 0.f);
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xp[kb] = tf_pack8(xs[2 * kb], xs[2 * kb + 1]);
        }

#pragma unroll
        for (int L = 0; L < NL; ++L) {
            const int FB = 80 + 192 * L, CB = 5 + 12 * L, PB = 128 + 1280 * L;
            // ---- V, token-major: V[key, d] = X . Wv^T  (A = the sequence, B = the packed fragment), 8 d tiles
            f16x4 vp[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                TF_STEP(CB + q)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = 4 * q + cc;
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) a = mfma16(xp[kb], TF_FR(FB + c * 4 + kb), a);
                    const float bv = prm[PB + 256 + 16 * c + n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] += bv;
                    vp[c] = tf_pack4(a);
                }
            }
            // ---- per head: Q^T, K^T tile pairs -> S^T -> softmax over keys -> O^T = V^T P^T
            f16x8 op[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                TF_STEP(CB + 2 + h)
                f32x4 qk[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) qk[i] = mfma16(TF_FR(FB + 32 + 16 * h + i * 4 + kb), xp[kb], qk[i]);
                    const f32x4 b = TF_B4(PB + (i >> 1) * 128 + 16 * (2 * h + (i & 1)) + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) qk[i][r] += b[r];
                }
                const f16x8 qp = tf_pack8(qk[0], qk[1]), kp = tf_pack8(qk[2], qk[3]);
                f32x4 s = mfma16(kp, qp, f32x4{0.f, 0.f, 0.f, 0.f});       // s[r]: key 4 g + r, query n
                float m = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) { if (!((vbits >> (8 * r)) & 0xffu)) s[r] = -INFINITY; m = fmaxf(m, s[r]); }
                m = tf_max4(m);
                const float mc = -m * c_exp;
                f32x4 e;
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c_exp, mc)); sum += e[r]; }
                const float inv = 1.0f / tf_sum4(sum);
                const f16x4 pp = tf_pack4(e);
                f32x4 o0 = mfma16k16(vp[2 * h], pp, f32x4{0.f, 0.f, 0.f, 0.f});
                f32x4 o1 = mfma16k16(vp[2 * h + 1], pp, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) { o0[r] *= inv; o1[r] *= inv; }
                op[h] = tf_pack8(o0, o1);
            }
            // ---- out_proj + residual -> LayerNorm 1
            f32x4 y[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                TF_STEP(CB + 6 + q)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = 4 * q + tt;
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) a = mfma16(TF_FR(FB + 96 + t * 4 + kb), op[kb], a);
                    const f32x4 b = TF_B4(PB + 384 + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[t][r] = a[r] + b[r] + xs[t][r];
                }
            }
            tf_layernorm(xs, y, prm + PB + 512, prm + PB + 640, g);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xp[kb] = tf_pack8(xs[2 * kb], xs[2 * kb + 1]);
            // ---- FFN: relu(W1 x + b1), W2 h + b2 + residual -> LayerNorm 2
            f16x8 hp[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                TF_STEP(CB + 8 + q)
                f32x4 hh[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = 4 * q + tt;
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) a = mfma16(TF_FR(FB + 128 + t * 4 + kb), xp[kb], a);
                    const f32x4 b = TF_B4(PB + 768 + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) hh[tt][r] = fmaxf(a[r] + b[r], 0.f);
                }
                hp[2 * q] = tf_pack8(hh[0], hh[1]);
                hp[2 * q + 1] = tf_pack8(hh[2], hh[3]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                TF_STEP(CB + 10 + q)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = 4 * q + tt;
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) a = mfma16(TF_FR(FB + 160 + t * 4 + kb), hp[kb], a);
                    const f32x4 b = TF_B4(PB + 896 + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[t][r] = a[r] + b[r] + xs[t][r];
                }
            }
            tf_layernorm(xs, y, prm + PB + 1024, prm + PB + 1152, g);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xp[kb] = tf_pack8(xs[2 * kb], xs[2 * kb + 1]);
        }

        // ---- output_proj (128 -> 1) + sigmoid
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 wv = TF_B4(128 + 1280 * NL + 16 * t + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) d = fmaf(xs[t][r], wv[r], d);
        }
        d = tf_sum4(d) + prm[128 + 1280 * NL + 128];
        if (live && g == 0) {
            const size_t o = (size_t)seq * 16 + n;
            if (p.logits) p.logits[o] = d;
            if (p.scores) p.scores[o] = sigmoidf_(d);
        }
    }
#undef TF_FR
#undef TF_DMA
#undef TF_STEP
#undef TF_B4
}

int launch_topo_fused(const TopoFusedParams& p, hipStream_t s) {
    if (p.nseq <= 0) return 0;
    if (p.nlayers != 0 && p.nlayers != 3) return -2;
    const int nprm = 128 + 1280 * p.nlayers + 132;
    const int lds = TF_RING + nprm * 4;
    static OncePerDevice opt_in;
    if (!opt_in.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(topo_fused_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, TF_RING + (128 + 1280 * 3 + 132) * 4) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(topo_fused_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, TF_RING + (128 + 132) * 4) == hipSuccess;
        }))
        return -3;
    const int groups = (p.nseq + TF_NW - 1) / TF_NW;
    const int grid = groups < 256 ? groups : 256;
    if (p.nlayers == 3) hipLaunchKernelGGL(topo_fused_kernel<3>, dim3(grid), dim3(TF_NW * 64), lds, s, p);
    else hipLaunchKernelGGL(topo_fused_kernel<0>, dim3(grid), dim3(TF_NW * 64), lds, s, p);
    return SRH_CHECK_LAUNCH();
}

}  // namespace srh

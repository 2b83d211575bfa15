// Host-side launch interface of the HIP kernels (internal; the public surface is include/samroad_hip.h).
#pragma once
#include "common.hpp"

namespace srh {

struct GemmParams {
    const f16* A = nullptr; int lda = 0;       // activations [M,K] row-major fp16
    const f16* W = nullptr; int ldw = 0;       // weights     [N,K] row-major fp16
    int M = 0, N = 0, K = 0;
    const float* bias = nullptr;               // [N]
    const float* resid = nullptr; int ldr = 0; // f32 [M,N], may alias out_f32 (in-place residual)
    const float* pos = nullptr; int pos_rows = 1;  // f32 [pos_rows,N], row m % pos_rows
    float* out_f32 = nullptr; int ldc = 0;
    f16* out_f16 = nullptr; int ldc16 = 0;
    int act = 0;                               // 0 none, 1 exact-erf GELU, 2 ReLU (applied after bias)
    int conv_S = 0, conv_C = 0;                // >0: implicit 3x3 conv over [B,S,S,C]
    int tile_gr = 0;                           // tuning aid (env SRH_Q192_GR): tile rows per XCD-local group, 0 = default 4
    int prio_mode = 0;                         // tuning aid (env SRH_Q192_PRIO): 0 s_setprio 1 around the MFMA segment, 1 none, 2 around the read segment, 3 one static s_setprio 1 for waves 4-7
    unsigned long long* dbg = nullptr;         // tuning aid (gemm_q192 ablation 3): per-segment cycle sums
    int variant = 0;                           // 0 LDS-DMA 128x128 (default), 1 register-staged 128x128, 2 register-staged 256x256
    // split-K for layers with too few 128x128 tiles to fill the chip (small M: ViT-L / ViT-H at 256 px): splitk workgroups
    // per tile write f32 partials to split_ws [splitk, M, N]; a second kernel sums them in a fixed order (deterministic)
    // and applies bias / residual / activation.  0 / 1 = off.
    int splitk = 0; float* split_ws = nullptr;
};
int launch_gemm(const GemmParams& p, hipStream_t s);
int gemm_splitk_factor(const GemmParams& p);   // what launch_gemm would use if a workspace were provided (1 = no split)
// persistent 256x192 kernel with the deferred epilogue (gemm_q192.hip); fp16 output, no residual / pos-embed
bool q192_supported(const GemmParams& p);
bool q192_preferred(const GemmParams& p);
int launch_gemm_q192(const GemmParams& p, hipStream_t s, int ablation = 0);

// Row LayerNorm (biased variance) f32 [M,D] -> fp16 and/or f32; gamma == nullptr => cast only.
struct NormParams {
    const float* x = nullptr; int M = 0, D = 0;
    const float* gamma = nullptr; const float* beta = nullptr; float eps = 1e-6f;
    int act = 0;                               // 1: GELU after affine
    f16* out_f16 = nullptr; float* out_f32 = nullptr;
    // optional fused residual add: x' = x + delta16 (fp16 [M,D], a GEMM's deferred-epilogue output) is what gets
    // normalised, and is written back to x_out (may alias x) when x_out != nullptr
    const f16* delta16 = nullptr; float* x_out = nullptr;
    // optional second branch output folded AFTER the first: x' = (x + delta16) + delta16b (needs delta16)
    const f16* delta16b = nullptr;
};
int launch_layernorm(const NormParams& p, hipStream_t s);

// Tile batcher + pixel normalise + patch im2col (K1/K2 prologue).
struct PatchParams {
    const void* src = nullptr;   // f32 [B,P,P,3] tiles (scene_S == 0) or u8/f32 scene [S,S,3]
    int src_is_u8 = 0;
    int B = 0, P = 0;
    int scene_S = 0;             // >0: crop tiles out of a resident scene at tile_xy
    const int* tile_xy = nullptr;  // device [B,2] (x0,y0)
    f16* out = nullptr;          // [B*(P/16)^2, 768] with k = ky*48 + kx*3 + c
};
int launch_patch_im2col(const PatchParams& p, hipStream_t s);

// Windowed / global attention with rel-pos bias; pad tokens are real keys with k = b_k, v = b_v.
struct AttnParams {
    const f16* qkv = nullptr; int ld = 0;   // [tokens, 3*D]: q | k | v, each head-major
    const f16* table_h = nullptr;           // [2*win-1, hd] fp16 rel-pos tables: the kernels derive the bias from the unscaled q
    const f16* table_w = nullptr;
    const f16* bias_qkv = nullptr;          // [3*D] fp16 (pad-token k / v rows)
    f16* out = nullptr; int ldo = 0;        // [tokens, D]
    int B = 0, S = 0, heads = 0, hd = 0, win = 0;  // win == S => global
    float scale = 0.125f;
    int ablate = 0;                         // tuning aid (env SRH_ATTN_ABL): 1 no key loop, 2 no K/V staging, 3 no fused rel-pos
};
int launch_attention(const AttnParams& p, hipStream_t s);
// attention_hdx.hip: MFMA attention for head dim 80 on 14x14 windows / the 16x16 global window (ViT-H at 256 px)
bool attention_hdx_supported(const AttnParams& p);
int launch_attention_hdx(const AttnParams& p, hipStream_t s);

// last two decoder layers fused (decoder.hip decode_tail_kernel): ConvT(64->32) + GELU + ConvT(32->2) + sigmoid + NHWC scatter
struct DecodeTailParams {
    const f16* x = nullptr;      // [B*S*S*16, 64] rows in quad-tree order (px, sub1, sub2): the dec3 GEMM's output
    const f16* w5 = nullptr;     // [128, 64]: n = sub3 * 32 + co
    const float* b5 = nullptr;   // [128]
    const float* w7 = nullptr;   // [8, 32]: n = (ky*2+kx)*2 + class
    const float* b7 = nullptr;   // [2]
    int B = 0, S = 0;
    float* logits = nullptr;     // nullable [B,P,P,2]
    float* scores = nullptr;     // nullable [B,P,P,2]
};
int launch_decode_tail(const DecodeTailParams& p, hipStream_t s);

// Scene canvases -> u8 masks (divide by analytic coverage count, x255, truncate; uncovered -> 0).
struct SceneNormParams {
    const float* canvas_kp = nullptr; const float* canvas_road = nullptr;
    const float* counter = nullptr;  // f32 [S,S] coverage count
    uint8_t* kp_u8 = nullptr; uint8_t* road_u8 = nullptr; int n = 0;
};
int launch_scene_normalise(const SceneNormParams& p, hipStream_t s);
int launch_scene_count(float* counter, int scene_S, const int* tile_xy, int n_tiles, int P, hipStream_t s);
int launch_scene_add(const float* scores, int B, int P, const int* tile_xy, float* kp, float* road, int scene_S, hipStream_t s);

// Bilinear sampler (F.grid_sample, align_corners=False, zeros) on channels-last embeddings.
struct SampleParams {
    const float* emb = nullptr;    // [B,h,w,C] channels-last f32
    const void* points = nullptr; int points_i64 = 0;  // [B,N,2] (x,y) pixels, int64 or f32
    int B = 0, N = 0, h = 0, w = 0, C = 0; float patch = 512.f;
    f16* out_f16 = nullptr; float* out_f32 = nullptr;   // [B*N, C]
};
int launch_sample(const SampleParams& p, hipStream_t s);

// TopoNet helpers.
struct PairGatherParams {
    const f16* pf = nullptr;        // relu(feature_proj) [B*N,128] fp16
    const void* points = nullptr; int points_i64 = 0;
    const void* pairs = nullptr; int pairs_i64 = 0;   // [B,Ns,K,2]
    int B = 0, N = 0, Ns = 0, Kp = 0; int zero_offset = 0;
    f16* out = nullptr; int ld = 0;  // [B*Ns*K, ld>=258] = src | tgt | dx dy | 0...
};
int launch_pair_gather(const PairGatherParams& p, hipStream_t s);

// fused TopoNet trunk (topo_fused.hip): pair rows -> logits / scores
struct TopoFusedParams {
    const f16* pair = nullptr; int ld_pair = 320;     // [nseq*16, ld_pair] gathered pair rows (src | tgt | dx dy | 0)
    const uint8_t* valid = nullptr;                   // [nseq, 16]
    const char* stream = nullptr;                     // packed MFMA fragments (api.hip pack_topo_fused)
    const float* params = nullptr;                    // biases, LayerNorm affine, output_proj
    int nlayers = 3, nseq = 0;
    float* logits = nullptr; float* scores = nullptr; // [nseq*16], nullable
};
int launch_topo_fused(const TopoFusedParams& p, hipStream_t s);

// ---- SAM MaskDecoder branch (sam_decoder.hip; reference model.py:260-282, :426-443) ----------------------------------------
struct SdLinearParams {                  // y[r,n] = act((x[r,:] + xadd[r % add_rows,:]) . W[n,:] + b[n]); f32, token side
    const float* x = nullptr; int ldx = 0; const float* xadd = nullptr; int add_rows = 1;
    const float* W = nullptr; const float* b = nullptr; int rows = 0, N = 0, K = 0, act = 0;
    float* y = nullptr; int ldy = 0;
};
int launch_sd_add_channel(const float* emb, const float* vec, float* out_f32, f16* out_f16, size_t rows, hipStream_t s);
int launch_sd_tok_linear(const SdLinearParams& p, hipStream_t s);
int launch_sd_tok_ln(const float* x, const float* resid, const float* g, const float* b, float* y, int rows, hipStream_t s);
int launch_sd_tok_selfattn(const float* q, const float* k, const float* v, float* out, int B, hipStream_t s);
int launch_sd_t2i_attn(const float* q, const f16* K, const f16* V, float* out, int B, int HW, hipStream_t s);
int launch_sd_i2t_attn(const f16* Q, const float* k, const float* v, f16* out, int B, int HW, hipStream_t s);
int launch_sd_ln64_gelu(const float* x, const float* g, const float* b, f16* y, size_t rows, hipStream_t s);
int launch_sd_mask(const f16* up, const float* hyper, float* low, int B, int S, hipStream_t s);
int launch_sd_upsample(const float* low, float* logits, float* scores, int B, int L, int P, hipStream_t s);

}  // namespace srh

// C ABI of libsamroad_hip.so (declared in include/samroad_hip.h): context / workspace, weight
// packing, and the launch sequences for SAMRoad.infer_masks_and_img_features, infer_toponet and
// the scene-level pass 1 of infer_one_img.  Host-side C++ only; all arithmetic is in the .hip kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/samroad_hip.h"
#include "kernels.hpp"

using namespace srh;

// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // grow-only; `slack` (a fraction of the request) is added when a buffer's size follows the data (the TopoNet
    // workspaces scale with the number of graph points of a batch): hipFree + hipMalloc synchronise the device, so
    // growing by a few rows per batch would cost milliseconds per call
    int ensure(size_t bytes, double slack = 0.0) {
        if (bytes <= cap) return 0;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + (size_t)(bytes * slack);
        if (hipMalloc(&p, want) != hipSuccess) return SRH_ERR_HIP;
        cap = want;
        return 0;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct ProfEntry { int cls; hipEvent_t e0, e1; double flops, bytes; };

struct srh_ctx {
    int device = 0;
    std::string err;
    // encoder / decoder workspace
    DevBuf a0, x, xn16, delta16, delta16b, qkv16, attn16, hid16, n1, n1_16, n2, emb16;
    DevBuf scores_ws, emb_ws, counter, split_ws;
    ZTileTables ztab;            // gemm_z192's tile-order tables (one bounded slab, freed with the context)
    // non-finite sentinel: NF_SLOTS flags in host-mapped pinned memory (nf_host; nf_dev = the same bytes as the device sees them).
    // The LayerNorm passes set flag `tag` when a row's variance is not finite (NormParams::nf) — an fp16 overflow upstream.  Nothing is
    // copied or synchronised in the hot loop: a set flag crosses PCIe once, the host reads its own memory at the next call.
    unsigned* nf_host = nullptr; unsigned* nf_dev = nullptr;
    // SAM MaskDecoder branch workspace
    DevBuf sd_keys, sd_keys16, sd_k16, sd_v16, sd_a16, sd_u0, sd_u0_16, sd_u1_16, sd_low, sd_tok;
    // toponet workspace
    DevBuf t_feat16, t_pf16, t_pair16;
    // profiling
    bool profiling = false;
    std::vector<std::string> cls_names;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

struct BlockW {
    int win = 0;
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    f16 *qkv_w, *qkv_b16, *rel_h, *rel_w, *proj_w, *fc1_w, *fc2_w;
};
// SAM MaskDecoder branch (USE_SAM_DECODER; sam_decoder.hip).  Attention a: image-side projection weights fp16 + the
// positional term pe . W^T precomputed at pack time; token-side weights f32.
struct SdAttnW { float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b; };                        // token-side self attention (f32)
struct SdT2IW { float *q_w, *q_b, *o_w, *o_b; f16 *k_w, *v_w; float *k_b, *v_b, *k_pos; };     // token -> image
struct SdI2TW { f16 *q_w, *o_w; float *q_b, *o_b, *q_pos, *k_w, *k_b, *v_w, *v_b; };           // image -> token
struct SdLayerW {
    SdAttnW self; SdT2IW t2i; SdI2TW i2t;
    float *n1_g, *n1_b, *n2_g, *n2_b, *n3_g, *n3_b, *n4_g, *n4_b, *l1_w, *l1_b, *l2_w, *l2_b;
};
struct SdW {
    float *no_mask, *tokens;                 // [256], [4,256] = iou_token | mask_tokens
    SdLayerW layer[2];
    SdT2IW fin; float *nf_g, *nf_b;
    f16 *up0_w, *up1_w; float *up0_b, *up_ln_g, *up_ln_b, *up1_b;
    float *hy_w[3][3], *hy_b[3][3];
};
struct srh_weights {
    srh_model_cfg cfg;
    SdW sd;
    int S = 0, D = 0, heads = 0, hd = 0;
    void* arena = nullptr; size_t arena_bytes = 0;
    f16* patch_w; float* patch_b; float* pos;
    std::vector<BlockW> blocks;
    f16 *neck0_w, *neck2_w; float *neck1_g, *neck1_b, *neck3_g, *neck3_b;
    char* dec_frags = nullptr; float* dec_prm = nullptr;          // fused map_decoder (decoder.hip): packed MFMA fragments + f32 parameters
    f16* tp_feat_w; float* tp_feat_b;
    char* tp_stream = nullptr; float* tp_params = nullptr;       // fused trunk (topo_fused.hip)
    int tp_layers = 0;                                            // encoder layers of the trunk (0: TOPONET_VERSION no_transformer)
};

constexpr int NF_SLOTS = 128, NF_NECK = 64, NF_DECODER = 66;   // tags: 2 * block + (0 norm1 | 1 norm2), neck LN2d 64 / 65, map_decoder LN2d 66

static int fail(srh_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
static int hip_fail(srh_ctx* c, hipError_t e, const char* where) {
    return fail(c, SRH_ERR_HIP, std::string(where) + ": " + hipGetErrorString(e));
}

// ---- profiling -----------------------------------------------------------------------------------
static int cls_id(srh_ctx* c, const char* name) {
    for (size_t i = 0; i < c->cls_names.size(); ++i)
        if (c->cls_names[i] == name) return (int)i;
    c->cls_names.push_back(name);
    return (int)c->cls_names.size() - 1;
}
static hipEvent_t next_event(srh_ctx* c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}
template <class F>
static int run(srh_ctx* c, const char* cls, double flops, double bytes, hipStream_t s, F&& f) {
    if (!c->profiling) return f();
    ProfEntry pe;
    pe.cls = cls_id(c, cls);
    pe.flops = flops; pe.bytes = bytes;
    pe.e0 = next_event(c);
    pe.e1 = next_event(c);
    hipEventRecord(pe.e0, s);
    const int rc = f();
    hipEventRecord(pe.e1, s);
    c->prof.push_back(pe);
    return rc;
}

static int gemm(srh_ctx* c, const char* cls, const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    p.ztab = &c->ztab;
    const int sk = gemm_splitk_factor(p);
    if (sk > 1) {       // small-M layers (ViT-L / ViT-H at 256 px): deterministic split-K through a ctx-owned f32 workspace
        if (c->split_ws.ensure((size_t)sk * p.M * p.N * 4)) return fail(c, SRH_ERR_HIP, "split-K workspace allocation failed");
        p.splitk = sk; p.split_ws = c->split_ws.as<float>();
    }
    const double fl = 2.0 * p.M * (double)p.N * p.K;
    const int rc = run(c, cls, fl, 0.0, s, [&] { return launch_gemm(p, s); });
    if (rc) return fail(c, rc == -2 ? SRH_ERR_UNSUPPORTED : SRH_ERR_HIP, std::string("gemm ") + cls + " launch failed");
    return 0;
}

// ---- C ABI: lifetime -----------------------------------------------------------------------------
extern "C" int srh_abi_version(void) { return SRH_ABI_VERSION; }

#ifndef SRH_BUILD_ID_HEX
#define SRH_BUILD_ID_HEX "unstamped-build!"
#endif
// sha256 of the sources this library was compiled from (sam_road_amd/build.py source_id); the marker is also what build.py
// greps the file for
static const char kBuildId[] = "SRH_BUILD_ID=" SRH_BUILD_ID_HEX;
extern "C" const char* srh_build_id(void) { return kBuildId + 13; }

extern "C" int srh_ctx_create(int device, srh_ctx** out) {
    if (!out) return SRH_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SRH_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SRH_ERR_HIP;
    srh_ctx* c = new srh_ctx();
    c->device = device;
    void* h = nullptr; void* d = nullptr;
    if (hipHostMalloc(&h, NF_SLOTS * sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        if (h) hipHostFree(h);
        delete c;
        return SRH_ERR_HIP;
    }
    memset(h, 0, NF_SLOTS * sizeof(unsigned));
    c->nf_host = (unsigned*)h; c->nf_dev = (unsigned*)d;
    *out = c;
    return SRH_OK;
}

extern "C" void srh_ctx_destroy(srh_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    DevBuf* bufs[] = {&c->a0, &c->x, &c->xn16, &c->delta16, &c->delta16b, &c->qkv16, &c->attn16, &c->hid16, &c->n1, &c->n1_16, &c->n2,
                      &c->emb16, &c->scores_ws, &c->emb_ws, &c->counter, &c->split_ws,
                      &c->sd_keys, &c->sd_keys16, &c->sd_k16, &c->sd_v16, &c->sd_a16, &c->sd_u0, &c->sd_u0_16, &c->sd_u1_16,
                      &c->sd_low, &c->sd_tok,
                      &c->t_feat16, &c->t_pf16, &c->t_pair16};
    for (DevBuf* b : bufs) b->release();
    c->ztab.release();
    if (c->nf_host) hipHostFree(c->nf_host);
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    delete c;
}

extern "C" size_t srh_ctx_device_bytes(const srh_ctx* c) {
    if (!c) return 0;
    const DevBuf* bufs[] = {&c->a0, &c->x, &c->xn16, &c->delta16, &c->delta16b, &c->qkv16, &c->attn16, &c->hid16, &c->n1, &c->n1_16, &c->n2,
                            &c->emb16, &c->scores_ws, &c->emb_ws, &c->counter, &c->split_ws,
                            &c->sd_keys, &c->sd_keys16, &c->sd_k16, &c->sd_v16, &c->sd_a16, &c->sd_u0, &c->sd_u0_16, &c->sd_u1_16,
                            &c->sd_low, &c->sd_tok, &c->t_feat16, &c->t_pf16, &c->t_pair16};
    size_t n = c->ztab.device_bytes();
    for (const DevBuf* b : bufs) n += b->cap;
    return n;
}

extern "C" const char* srh_last_error(const srh_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

// The sentinel flags as the host sees them NOW (no synchronisation): SRH_ERR_NONFINITE + a message naming the first stage that saw an
// Inf / NaN, flags cleared; 0 if none is set.  The reference's own guards (inferencer.py:206 NaN -> -100, :219 assert 0 <= score <= 1)
// only look at TopoNet's output; an fp16 overflow in the encoder would otherwise come out as silently wrong masks.
static int nonfinite_check(srh_ctx* c, const char* who) {
    if (!c->nf_host) return 0;
    int first = -1;
    for (int i = 0; i < NF_SLOTS; ++i)
        if (reinterpret_cast<volatile unsigned*>(c->nf_host)[i]) { if (first < 0) first = i; c->nf_host[i] = 0; }
    if (first < 0) return 0;
    std::string where;
    if (first < NF_NECK) where = "encoder block " + std::to_string(first / 2) + (first & 1 ? ", norm2 (the block's attention branch output or its residual stream)"
                                                                                         : ", norm1 (the previous block's MLP output, the patch embedding for block 0, or the residual stream)");
    else if (first < NF_DECODER) where = std::string("neck LayerNorm2d ") + (first == NF_NECK ? "1 (the last block's output)" : "2");
    else where = "map_decoder LayerNorm2d";
    return fail(c, SRH_ERR_NONFINITE, std::string(who) + ": non-finite activations (fp16 overflow?) in an earlier call on this context, first seen by " + where +
                                      "; the outputs of that call are invalid");
}

extern "C" int srh_ctx_check(srh_ctx* c, void* stream, int synchronize) {
    if (!c) return SRH_ERR_BAD_ARG;
    if (synchronize) {
        hipSetDevice(c->device);
        const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(c, e, "srh_ctx_check");
    }
    return nonfinite_check(c, "srh_ctx_check");
}

// ---- weight packing --------------------------------------------------------------------------------
namespace {
struct Packer {
    srh_ctx* c;
    std::map<std::string, const srh_named_tensor*> by_name;
    std::vector<char> host;                       // staging image of the device arena
    std::vector<std::pair<void**, size_t>> fix;   // pointer slots to patch with arena + offset
    std::vector<float> tmp;
    std::string missing;
    bool layout_only = false;                     // srh_weights_import: only the arena layout is wanted, no tensor is read

    size_t alloc(size_t bytes) {
        const size_t off = (host.size() + 255) & ~size_t(255);
        host.resize(off + bytes);
        return off;
    }
    // fetch tensor as host f32 (copying from device if needed); checks element count
    const float* get(const std::string& name, size_t expect) {
        if (layout_only) return nullptr;
        auto it = by_name.find(name);
        if (it == by_name.end()) { if (missing.empty()) missing = name; return nullptr; }
        const srh_named_tensor* t = it->second;
        size_t n = 1;
        for (int i = 0; i < t->ndim; ++i) n *= (size_t)t->shape[i];
        if (n != expect) { if (missing.empty()) missing = name + " (shape mismatch)"; return nullptr; }
        if (!t->on_device) return reinterpret_cast<const float*>(t->data);
        tmp.resize(n);
        if (hipMemcpy(tmp.data(), t->data, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            if (missing.empty()) missing = name + " (D2H copy failed)";
            return nullptr;
        }
        return tmp.data();
    }
    template <class T> void slot(T** dst, size_t off) { fix.push_back({reinterpret_cast<void**>(dst), off}); }

    void put_f32(float** dst, const std::string& name, size_t n) {
        const float* src = get(name, n);
        const size_t off = alloc(n * 4);
        if (src) memcpy(host.data() + off, src, n * 4);
        slot(dst, off);
    }
    // fp16 copy with an index map: out[i] = src[map(i)]
    template <class Map>
    void put_f16(f16** dst, const std::string& name, size_t n_src, size_t n_out, Map map) {
        const float* src = get(name, n_src);
        const size_t off = alloc(n_out * 2);
        if (src) {
            f16* o = reinterpret_cast<f16*>(host.data() + off);
            for (size_t i = 0; i < n_out; ++i) {
                const long j = map(i);
                o[i] = j < 0 ? (f16)0.f : (f16)src[j];
            }
        }
        slot(dst, off);
    }
    void put_f16_same(f16** dst, const std::string& name, size_t n) {
        put_f16(dst, name, n, n, [](size_t i) { return (long)i; });
    }
};

// Weights of the fused TopoNet trunk (topo_fused.hip): every matrix cut into 16 (out) x 32 (in) MFMA A fragments of 1 KiB,
// lane l = (row i = l & 15, k group g = l >> 4) holding 8 halves W[16 rt + i][k(kb, g, j)], in the exact order the kernel
// consumes them.  pair_proj reads its input from memory, so k = 32 kb + 8 g + j; every later matrix reads activations that
// are C-layout tile pairs of the previous MFMA, so k = 32 kb + 16 (j >> 2) + 4 g + (j & 3).
static void pack_topo_fused(Packer& pk, srh_weights* w, int nl) {
    const std::string T = "topo_net.";
    const size_t nfrag = 80 + (size_t)192 * nl, nprm = 128 + (size_t)1280 * nl + 132;
    const size_t soff = pk.alloc(nfrag * 1024), poff = pk.alloc(nprm * 4);
    size_t f = 0;      // next fragment
    auto frag = [&](const float* W, int ldw, int row0, int kb, bool perm, int kmax) {
        f16* o = reinterpret_cast<f16*>(pk.host.data() + soff) + f * 512;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int i = l & 15, g = l >> 4;
                const int k = perm ? 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3) : 32 * kb + 8 * g + j;
                o[l * 8 + j] = (W && k < kmax) ? (f16)W[(size_t)(row0 + i) * ldw + k] : (f16)0.f;
            }
        ++f;
    };
    auto prm = [&](size_t at, const std::string& name, size_t n) {
        const float* src = pk.get(name, n);
        if (src) memcpy(pk.host.data() + poff + at * 4, src, n * 4);
    };
    {
        const float* W = pk.get(T + "pair_proj.weight", 128 * 258);
        for (int kb = 0; kb < 10; ++kb)
            for (int rt = 0; rt < 8; ++rt) frag(W, 258, 16 * rt, kb, false, 258);
    }
    prm(0, T + "pair_proj.bias", 128);
    for (int l = 0; l < nl; ++l) {
        const std::string L = T + "transformer_encoder.layers." + std::to_string(l) + ".";
        const size_t pb = 128 + (size_t)1280 * l;
        {
            const float* W = pk.get(L + "self_attn.in_proj_weight", 384 * 128);
            for (int c = 0; c < 8; ++c)
                for (int kb = 0; kb < 4; ++kb) frag(W, 128, 256 + 16 * c, kb, true, 128);                    // V
            for (int h = 0; h < 4; ++h)
                for (int i = 0; i < 4; ++i)                                                                  // Q 2h, Q 2h+1, K 2h, K 2h+1
                    for (int kb = 0; kb < 4; ++kb) frag(W, 128, (i >> 1) * 128 + 16 * (2 * h + (i & 1)), kb, true, 128);
        }
        for (const char* m : {"self_attn.out_proj.weight", "linear1.weight", "linear2.weight"}) {
            const float* W = pk.get(L + m, 128 * 128);
            for (int rt = 0; rt < 8; ++rt)
                for (int kb = 0; kb < 4; ++kb) frag(W, 128, 16 * rt, kb, true, 128);
        }
        prm(pb + 0, L + "self_attn.in_proj_bias", 384);
        prm(pb + 384, L + "self_attn.out_proj.bias", 128);
        prm(pb + 512, L + "norm1.weight", 128);
        prm(pb + 640, L + "norm1.bias", 128);
        prm(pb + 768, L + "linear1.bias", 128);
        prm(pb + 896, L + "linear2.bias", 128);
        prm(pb + 1024, L + "norm2.weight", 128);
        prm(pb + 1152, L + "norm2.bias", 128);
    }
    prm(128 + (size_t)1280 * nl, T + "output_proj.weight", 128);
    prm(128 + (size_t)1280 * nl + 128, T + "output_proj.bias", 1);
    pk.slot(&w->tp_stream, soff);
    pk.slot(&w->tp_params, poff);
}

// Weights of the fused map_decoder (decoder.hip decode_fused_kernel; reference model.py:286-295).  A ConvTranspose2d(k2, s2) layer is a
// per-pixel GEMM to 4 x Cout columns, n = (ky * 2 + kx) * Cout + co, W_gemm[n][ci] = w[ci][co][ky][kx]; every matrix is cut into
// 16 (out) x 32 (in) MFMA A fragments of 1 KiB as in pack_topo_fused (lane l = row i = l & 15, k group g = l >> 4, 8 halves).  Layer 0
// reads its input from memory (natural k order); layers 3 and 5 read C-layout tile pairs of the previous MFMA (permuted k order).
//   frags: L0 [sub1 4][kb 8][rt 8] (64 KiB per sub1), L3 [sub2 4][kb 4][rt 4] (64 KiB), L5 [kb 2][rt 8] (16 KiB)
//   prm  : b0[128] | ln gamma[128] | ln beta[128] | b3[64] | b5[32] | w7[8][32] (n = (ky*2+kx)*2 + class) | b7[2]
static void pack_decoder_fused(Packer& pk, srh_weights* w) {
    const size_t nfrag = 256 + 64 + 16, nprm = 768;         // 738 parameters, padded to 3 KiB: the kernel stages them by three 1-KiB LDS-DMA pieces
    const size_t foff = pk.alloc(nfrag * 1024), poff = pk.alloc(nprm * 4);
    size_t f = 0;
    std::vector<float> wg;                                 // the layer's GEMM weight [4 * cout][cin]
    auto load = [&](const std::string& name, int cin, int cout) -> bool {
        const float* src = pk.get(name, (size_t)cin * cout * 4);
        if (!src) return false;
        wg.assign((size_t)4 * cout * cin, 0.f);
        for (int sub = 0; sub < 4; ++sub)
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci) wg[((size_t)sub * cout + co) * cin + ci] = src[((size_t)ci * cout + co) * 4 + sub];
        return true;
    };
    auto frag = [&](bool have, int ldw, int row0, int kb, bool perm) {
        f16* o = reinterpret_cast<f16*>(pk.host.data() + foff) + f * 512;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int i = l & 15, g = l >> 4;
                const int k = perm ? 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3) : 32 * kb + 8 * g + j;
                o[l * 8 + j] = have ? (f16)wg[(size_t)(row0 + i) * ldw + k] : (f16)0.f;
            }
        ++f;
    };
    {
        const bool have = load("map_decoder.0.weight", 256, 128);
        for (int s1 = 0; s1 < 4; ++s1)
            for (int kb = 0; kb < 8; ++kb)
                for (int rt = 0; rt < 8; ++rt) frag(have, 256, s1 * 128 + 16 * rt, kb, false);
    }
    {
        const bool have = load("map_decoder.3.weight", 128, 64);
        for (int s2 = 0; s2 < 4; ++s2)
            for (int kb = 0; kb < 4; ++kb)
                for (int rt = 0; rt < 4; ++rt) frag(have, 128, s2 * 64 + 16 * rt, kb, true);
    }
    {
        const bool have = load("map_decoder.5.weight", 64, 32);
        for (int kb = 0; kb < 2; ++kb)
            for (int rt = 0; rt < 8; ++rt) frag(have, 64, 16 * rt, kb, true);
    }
    auto prm = [&](size_t at, const std::string& name, size_t n) {
        const float* src = pk.get(name, n);
        if (src) memcpy(pk.host.data() + poff + at * 4, src, n * 4);
    };
    prm(0, "map_decoder.0.bias", 128);
    prm(128, "map_decoder.1.weight", 128);
    prm(256, "map_decoder.1.bias", 128);
    prm(384, "map_decoder.3.bias", 64);
    prm(448, "map_decoder.5.bias", 32);
    if (const float* src = pk.get("map_decoder.7.weight", 32 * 2 * 4)) {
        float* o = reinterpret_cast<float*>(pk.host.data() + poff) + 480;
        for (int nn = 0; nn < 8; ++nn)
            for (int ci = 0; ci < 32; ++ci) o[nn * 32 + ci] = src[(ci * 2 + (nn & 1)) * 4 + (nn >> 1)];
    }
    prm(736, "map_decoder.7.bias", 2);
    pk.slot(&w->dec_frags, foff);
    pk.slot(&w->dec_prm, poff);
}

// SAM MaskDecoder branch: prompt_encoder.* / mask_decoder.* (fork key names; oracle/sam_decoder.py).  The random-Fourier
// positional encoding of the S x S grid (get_dense_pe) is constant: pe and every pe . W^T the decoder needs are computed here.
static void pack_sam_decoder(Packer& pk, srh_weights* w) {
    const int S = w->S, HW = S * S;
    SdW& d = w->sd;
    const std::string PE = "prompt_encoder.", MD = "mask_decoder.", TR = "mask_decoder.transformer.";
    pk.put_f32(&d.no_mask, PE + "no_mask_embed.weight", 256);
    // parameters the no-prompt path never reads must still be present in a checkpoint of this branch
    for (const char* k : {"point_embeddings.0.weight", "point_embeddings.1.weight", "point_embeddings.2.weight",
                          "point_embeddings.3.weight", "not_a_point_embed.weight"}) (void)pk.get(PE + k, 256);
    {
        const float* it = pk.get(MD + "iou_token.weight", 256);
        std::vector<float> tok(4 * 256, 0.f);
        if (it) memcpy(tok.data(), it, 256 * 4);
        const float* mt = pk.get(MD + "mask_tokens.weight", 3 * 256);
        if (mt) memcpy(tok.data() + 256, mt, 3 * 256 * 4);
        const size_t off = pk.alloc(4 * 256 * 4);
        memcpy(pk.host.data() + off, tok.data(), 4 * 256 * 4);
        pk.slot(&d.tokens, off);
    }
    // pe[y*S + x][c]: coords ((x + .5)/S, (y + .5)/S) -> 2c - 1 -> @ G[2,128] -> 2 pi -> sin | cos
    std::vector<float> pe((size_t)HW * 256, 0.f);
    {
        const float* G = pk.get(PE + "pe_layer.positional_encoding_gaussian_matrix", 2 * 128);
        if (G)
            for (int y = 0; y < S; ++y)
                for (int x = 0; x < S; ++x) {
                    const float cx = 2.f * (((float)x + 0.5f) / (float)S) - 1.f, cy = 2.f * (((float)y + 0.5f) / (float)S) - 1.f;
                    for (int j = 0; j < 128; ++j) {
                        const float a = 2.f * 3.14159265358979323846f * (cx * G[j] + cy * G[128 + j]);
                        pe[((size_t)y * S + x) * 256 + j] = sinf(a);
                        pe[((size_t)y * S + x) * 256 + 128 + j] = cosf(a);
                    }
                }
    }
    auto put_pos = [&](float** dst, const float* W) {      // pe [HW,256] . W[128,256]^T -> [HW,128] f32
        const size_t off = pk.alloc((size_t)HW * 128 * 4);
        if (W) {
            float* o = reinterpret_cast<float*>(pk.host.data() + off);
            for (int t = 0; t < HW; ++t)
                for (int n = 0; n < 128; ++n) {
                    double a = 0.0;
                    for (int k = 0; k < 256; ++k) a += (double)pe[(size_t)t * 256 + k] * W[(size_t)n * 256 + k];
                    o[(size_t)t * 128 + n] = (float)a;
                }
        }
        pk.slot(dst, off);
    };
    auto put_t2i = [&](SdT2IW& a, const std::string& base) {
        pk.put_f32(&a.q_w, base + "q_proj.weight", 128 * 256); pk.put_f32(&a.q_b, base + "q_proj.bias", 128);
        pk.put_f16_same(&a.k_w, base + "k_proj.weight", 128 * 256); pk.put_f32(&a.k_b, base + "k_proj.bias", 128);
        pk.put_f16_same(&a.v_w, base + "v_proj.weight", 128 * 256); pk.put_f32(&a.v_b, base + "v_proj.bias", 128);
        pk.put_f32(&a.o_w, base + "out_proj.weight", 256 * 128); pk.put_f32(&a.o_b, base + "out_proj.bias", 256);
        std::vector<float> kw(128 * 256);
        const float* src = pk.get(base + "k_proj.weight", 128 * 256);
        if (src) memcpy(kw.data(), src, kw.size() * 4);      // get() may reuse its staging buffer: copy before the next get
        put_pos(&a.k_pos, src ? kw.data() : nullptr);
    };
    auto put_i2t = [&](SdI2TW& a, const std::string& base) {
        pk.put_f16_same(&a.q_w, base + "q_proj.weight", 128 * 256); pk.put_f32(&a.q_b, base + "q_proj.bias", 128);
        pk.put_f32(&a.k_w, base + "k_proj.weight", 128 * 256); pk.put_f32(&a.k_b, base + "k_proj.bias", 128);
        pk.put_f32(&a.v_w, base + "v_proj.weight", 128 * 256); pk.put_f32(&a.v_b, base + "v_proj.bias", 128);
        pk.put_f16_same(&a.o_w, base + "out_proj.weight", 256 * 128); pk.put_f32(&a.o_b, base + "out_proj.bias", 256);
        std::vector<float> qw(128 * 256);
        const float* src = pk.get(base + "q_proj.weight", 128 * 256);
        if (src) memcpy(qw.data(), src, qw.size() * 4);
        put_pos(&a.q_pos, src ? qw.data() : nullptr);
    };
    for (int l = 0; l < 2; ++l) {
        SdLayerW& L = d.layer[l];
        const std::string B_ = TR + "layers." + std::to_string(l) + ".";
        pk.put_f32(&L.self.q_w, B_ + "self_attn.q_proj.weight", 256 * 256); pk.put_f32(&L.self.q_b, B_ + "self_attn.q_proj.bias", 256);
        pk.put_f32(&L.self.k_w, B_ + "self_attn.k_proj.weight", 256 * 256); pk.put_f32(&L.self.k_b, B_ + "self_attn.k_proj.bias", 256);
        pk.put_f32(&L.self.v_w, B_ + "self_attn.v_proj.weight", 256 * 256); pk.put_f32(&L.self.v_b, B_ + "self_attn.v_proj.bias", 256);
        pk.put_f32(&L.self.o_w, B_ + "self_attn.out_proj.weight", 256 * 256); pk.put_f32(&L.self.o_b, B_ + "self_attn.out_proj.bias", 256);
        put_t2i(L.t2i, B_ + "cross_attn_token_to_image.");
        put_i2t(L.i2t, B_ + "cross_attn_image_to_token.");
        pk.put_f32(&L.n1_g, B_ + "norm1.weight", 256); pk.put_f32(&L.n1_b, B_ + "norm1.bias", 256);
        pk.put_f32(&L.n2_g, B_ + "norm2.weight", 256); pk.put_f32(&L.n2_b, B_ + "norm2.bias", 256);
        pk.put_f32(&L.n3_g, B_ + "norm3.weight", 256); pk.put_f32(&L.n3_b, B_ + "norm3.bias", 256);
        pk.put_f32(&L.n4_g, B_ + "norm4.weight", 256); pk.put_f32(&L.n4_b, B_ + "norm4.bias", 256);
        pk.put_f32(&L.l1_w, B_ + "mlp.lin1.weight", 2048 * 256); pk.put_f32(&L.l1_b, B_ + "mlp.lin1.bias", 2048);
        pk.put_f32(&L.l2_w, B_ + "mlp.lin2.weight", 256 * 2048); pk.put_f32(&L.l2_b, B_ + "mlp.lin2.bias", 256);
    }
    put_t2i(d.fin, TR + "final_attn_token_to_image.");
    pk.put_f32(&d.nf_g, TR + "norm_final_attn.weight", 256); pk.put_f32(&d.nf_b, TR + "norm_final_attn.bias", 256);
    // output_upscaling: ConvTranspose2d weight [Cin,Cout,2,2] -> GEMM weight [n = (ky*2+kx)*Cout + co][ci], bias replicated x4
    auto convt = [&](f16** dst, float** bdst, const std::string& idx, int cin, int cout) {
        pk.put_f16(dst, MD + "output_upscaling." + idx + ".weight", (size_t)cin * cout * 4, (size_t)4 * cout * cin,
                   [cin, cout](size_t i) {
                       const size_t nidx = i / cin, ci = i % cin, sub = nidx / cout, co = nidx % cout;
                       return (long)((ci * cout + co) * 4 + sub);
                   });
        const float* bsrc = pk.get(MD + "output_upscaling." + idx + ".bias", cout);
        const size_t off = pk.alloc((size_t)4 * cout * 4);
        if (bsrc)
            for (int r = 0; r < 4; ++r) memcpy(pk.host.data() + off + (size_t)r * cout * 4, bsrc, (size_t)cout * 4);
        pk.slot(bdst, off);
    };
    convt(&d.up0_w, &d.up0_b, "0", 256, 64);
    pk.put_f32(&d.up_ln_g, MD + "output_upscaling.1.weight", 64); pk.put_f32(&d.up_ln_b, MD + "output_upscaling.1.bias", 64);
    convt(&d.up1_w, &d.up1_b, "3", 64, 32);
    for (int i = 0; i < 3; ++i)
        for (int l = 0; l < 3; ++l) {
            const std::string H = MD + "output_hypernetworks_mlps." + std::to_string(i) + ".layers." + std::to_string(l) + ".";
            const int out = l == 2 ? 32 : 256;
            pk.put_f32(&d.hy_w[i][l], H + "weight", (size_t)out * 256); pk.put_f32(&d.hy_b[i][l], H + "bias", out);
        }
    // iou_prediction_head: its output is discarded by the reference (model.py:430 `low_res_logits, iou_predictions`); only presence is checked
    for (int l = 0; l < 3; ++l) {
        const std::string H = MD + "iou_prediction_head.layers." + std::to_string(l) + ".";
        (void)pk.get(H + "weight", (size_t)(l == 2 ? 3 : 256) * 256); (void)pk.get(H + "bias", l == 2 ? 3 : 256);
    }
}
}  // namespace

// arena_src == nullptr: pack from the state_dict entries.  Otherwise (srh_weights_import): the arena LAYOUT depends on cfg alone
// (the same sequence of allocations), so the pointer table is rebuilt without reading a tensor and the packed bytes are copied
// device-to-device from arena_src.
static int pack_impl(srh_ctx* c, const srh_model_cfg* cfg, const srh_named_tensor* tensors, int n, const void* arena_src,
                     size_t arena_src_bytes, srh_weights** out) {
    *out = nullptr;
    const int D = cfg->embed_dim, heads = cfg->num_heads;
    if (D <= 0 || heads <= 0 || D % heads) return fail(c, SRH_ERR_BAD_ARG, "bad embed_dim / num_heads");
    const int hd = D / heads;
    if (hd != 64 && hd != 80) return fail(c, SRH_ERR_UNSUPPORTED, "head_dim must be 64 (MFMA attention kernels) or 80 (ViT-H: generic kernel)");
    if (cfg->patch_size % 16) return fail(c, SRH_ERR_BAD_ARG, "PATCH_SIZE must be a multiple of 16");
    const int S = cfg->patch_size / 16;
    if (S != 16 && S != 32 && S != 64) return fail(c, SRH_ERR_UNSUPPORTED, "PATCH_SIZE must be 256, 512 or 1024");
    if (cfg->window_size != 14) return fail(c, SRH_ERR_UNSUPPORTED, "window_size must be 14");
    if (D % 128 || (D != 768 && D != 1024 && D != 1280)) return fail(c, SRH_ERR_UNSUPPORTED, "embed_dim must be 768, 1024 or 1280");
    hipSetDevice(c->device);

    srh_weights* w = new srh_weights();
    w->cfg = *cfg; w->S = S; w->D = D; w->heads = heads; w->hd = hd;
    Packer pk;
    pk.c = c;
    pk.layout_only = arena_src != nullptr;
    for (int i = 0; i < n; ++i) pk.by_name[tensors[i].name] = &tensors[i];
    const std::string E = "image_encoder.";

    pk.put_f16(&w->patch_w, E + "patch_embed.proj.weight", (size_t)D * 768, (size_t)D * 768, [](size_t i) {
        const size_t nidx = i / 768, k = i % 768;
        const size_t ky = k / 48, kx = (k % 48) / 3, ch = k % 3;
        return (long)(nidx * 768 + ch * 256 + ky * 16 + kx);
    });
    pk.put_f32(&w->patch_b, E + "patch_embed.proj.bias", D);
    pk.put_f32(&w->pos, E + "pos_embed", (size_t)S * S * D);

    w->blocks.resize(cfg->depth);
    for (int i = 0; i < cfg->depth; ++i) {
        BlockW& b = w->blocks[i];
        bool global = false;
        for (int g = 0; g < cfg->n_global; ++g) global |= cfg->global_attn_indexes[g] == i;
        b.win = global ? S : cfg->window_size;
        const std::string B_ = E + "blocks." + std::to_string(i) + ".";
        pk.put_f32(&b.ln1_g, B_ + "norm1.weight", D);
        pk.put_f32(&b.ln1_b, B_ + "norm1.bias", D);
        pk.put_f32(&b.ln2_g, B_ + "norm2.weight", D);
        pk.put_f32(&b.ln2_b, B_ + "norm2.bias", D);
        pk.put_f16_same(&b.qkv_w, B_ + "attn.qkv.weight", (size_t)3 * D * D);
        pk.put_f32(&b.qkv_b, B_ + "attn.qkv.bias", (size_t)3 * D);
        pk.put_f16_same(&b.qkv_b16, B_ + "attn.qkv.bias", (size_t)3 * D);
        pk.put_f16_same(&b.rel_h, B_ + "attn.rel_pos_h", (size_t)(2 * b.win - 1) * hd);
        pk.put_f16_same(&b.rel_w, B_ + "attn.rel_pos_w", (size_t)(2 * b.win - 1) * hd);
        pk.put_f16_same(&b.proj_w, B_ + "attn.proj.weight", (size_t)D * D);
        pk.put_f32(&b.proj_b, B_ + "attn.proj.bias", D);
        pk.put_f16_same(&b.fc1_w, B_ + "mlp.lin1.weight", (size_t)4 * D * D);
        pk.put_f32(&b.fc1_b, B_ + "mlp.lin1.bias", (size_t)4 * D);
        pk.put_f16_same(&b.fc2_w, B_ + "mlp.lin2.weight", (size_t)4 * D * D);
        pk.put_f32(&b.fc2_b, B_ + "mlp.lin2.bias", D);
    }
    pk.put_f16_same(&w->neck0_w, E + "neck.0.weight", (size_t)256 * D);
    pk.put_f32(&w->neck1_g, E + "neck.1.weight", 256);
    pk.put_f32(&w->neck1_b, E + "neck.1.bias", 256);
    pk.put_f16(&w->neck2_w, E + "neck.2.weight", (size_t)256 * 256 * 9, (size_t)256 * 2304, [](size_t i) {
        const size_t nidx = i / 2304, k = i % 2304, tap = k / 256, ch = k % 256;
        return (long)((nidx * 256 + ch) * 9 + tap);
    });
    pk.put_f32(&w->neck3_g, E + "neck.3.weight", 256);
    pk.put_f32(&w->neck3_b, E + "neck.3.bias", 256);

    if (cfg->use_sam_decoder) {
        pack_sam_decoder(pk, w);
    } else {
        pack_decoder_fused(pk, w);
    }

    // TopoNet
    const std::string T = "topo_net.";
    pk.put_f16_same(&w->tp_feat_w, T + "feature_proj.weight", 128 * 256);
    pk.put_f32(&w->tp_feat_b, T + "feature_proj.bias", 128);
    w->tp_layers = cfg->toponet_version != 2 ? 3 : 0;
    pack_topo_fused(pk, w, w->tp_layers);

    if (!pk.missing.empty()) {
        delete w;
        return fail(c, SRH_ERR_MISSING_WEIGHT, "state_dict entry missing or mis-shaped: " + pk.missing);
    }
    if (arena_src && arena_src_bytes != pk.host.size()) {
        delete w;
        return fail(c, SRH_ERR_BAD_ARG, "srh_weights_import: the packed arena has " + std::to_string(arena_src_bytes) +
                                        " bytes, this configuration packs to " + std::to_string(pk.host.size()));
    }
    hipError_t e = hipMalloc(&w->arena, pk.host.size());
    if (e != hipSuccess) { delete w; return hip_fail(c, e, "hipMalloc(weights)"); }
    w->arena_bytes = pk.host.size();
    e = arena_src ? hipMemcpy(w->arena, arena_src, pk.host.size(), hipMemcpyDeviceToDevice)
                  : hipMemcpy(w->arena, pk.host.data(), pk.host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(w->arena); delete w; return hip_fail(c, e, "hipMemcpy(weights)"); }
    for (auto& f : pk.fix) *f.first = reinterpret_cast<char*>(w->arena) + f.second;
    *out = w;
    return SRH_OK;
}

extern "C" int srh_weights_pack(srh_ctx* c, const srh_model_cfg* cfg, const srh_named_tensor* tensors, int n,
                                srh_weights** out) {
    if (!c || !cfg || !tensors || !out) return fail(c, SRH_ERR_BAD_ARG, "srh_weights_pack: null argument");
    return pack_impl(c, cfg, tensors, n, nullptr, 0, out);
}

extern "C" int srh_weights_export(srh_ctx* c, const srh_weights* w, void* dst, size_t capacity, size_t* bytes) {
    if (!c || !w || !bytes) return fail(c, SRH_ERR_BAD_ARG, "srh_weights_export: null argument");
    *bytes = w->arena_bytes;
    if (!dst) return SRH_OK;
    if (capacity < w->arena_bytes) return fail(c, SRH_ERR_BAD_ARG, "srh_weights_export: destination too small");
    hipSetDevice(c->device);
    const hipError_t e = hipMemcpy(dst, w->arena, w->arena_bytes, hipMemcpyDeviceToDevice);
    return e == hipSuccess ? SRH_OK : hip_fail(c, e, "srh_weights_export");
}

extern "C" int srh_weights_import(srh_ctx* c, const srh_model_cfg* cfg, const void* src, size_t bytes, srh_weights** out) {
    if (!c || !cfg || !src || !out) return fail(c, SRH_ERR_BAD_ARG, "srh_weights_import: null argument");
    return pack_impl(c, cfg, nullptr, 0, src, bytes, out);
}

extern "C" void srh_weights_free(srh_weights* w) {
    if (!w) return;
    if (w->arena) hipFree(w->arena);
    delete w;
}

// ---- encoder + decoder -------------------------------------------------------------------------------
static double attn_flops(int B, int S, int heads, int hd, int win) {
    if (win == S) return 4.0 * B * heads * (double)S * S * S * S * hd;
    const int nw = (S + win - 1) / win;
    double q = 0;
    for (int wy = 0; wy < nw; ++wy)
        for (int wx = 0; wx < nw; ++wx)
            q += (double)std::min(win, S - wy * win) * std::min(win, S - wx * win);
    return 4.0 * B * heads * q * win * win * hd;
}

static int ensure_encoder_ws(srh_ctx* c, const srh_weights* w, int B) {
    const size_t T = (size_t)B * w->S * w->S, D = w->D;
    int rc = 0;
    rc |= c->a0.ensure(T * 768 * 2);
    rc |= c->x.ensure(T * D * 4);
    rc |= c->xn16.ensure(T * D * 2);
    rc |= c->delta16.ensure(T * D * 2);
    rc |= c->delta16b.ensure(T * D * 2);
    rc |= c->qkv16.ensure(T * 3 * D * 2);
    rc |= c->attn16.ensure(T * D * 2);
    rc |= c->hid16.ensure(T * 4 * D * 2);
    rc |= c->n1.ensure(T * 256 * 4);
    rc |= c->n1_16.ensure(T * 256 * 2);
    rc |= c->n2.ensure(T * 256 * 4);
    rc |= c->emb16.ensure(T * 256 * 2);
    return rc ? fail(c, SRH_ERR_HIP, "workspace allocation failed") : 0;
}

#define TRY(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)
#define TRYK(c, cls, fl, by, s, call) do { const int rc_ = run(c, cls, fl, by, s, [&] { return (call); }); \
    if (rc_) return fail(c, rc_ == -2 ? SRH_ERR_UNSUPPORTED : SRH_ERR_HIP, std::string(cls) + ": kernel launch failed"); } while (0)

// ---- SAM MaskDecoder branch (model.py:426-443 / :471-488; kernels in sam_decoder.hip, semantics in oracle/sam_decoder.py) ----
static int sam_decode(srh_ctx* c, const srh_weights* w, int B, const float* emb, float* logits, float* scores, hipStream_t s) {
    const int S = w->S, HW = S * S, R = B * 4, P = w->cfg.patch_size;
    const size_t T = (size_t)B * HW;
    const SdW& d = w->sd;
    int rc = 0;
    rc |= c->sd_keys.ensure(T * 256 * 4);  rc |= c->sd_keys16.ensure(T * 256 * 2);
    rc |= c->sd_k16.ensure(T * 128 * 2);   rc |= c->sd_v16.ensure(T * 128 * 2);  rc |= c->sd_a16.ensure(T * 128 * 2);
    rc |= c->sd_u0.ensure(T * 256 * 4);    rc |= c->sd_u0_16.ensure(T * 256 * 2); rc |= c->sd_u1_16.ensure(T * 4 * 128 * 2);
    rc |= c->sd_low.ensure((size_t)B * 2 * 16 * HW * 4);
    rc |= c->sd_tok.ensure((size_t)R * (256 * 8 + 2048 + 128 * 4) * 4 + (size_t)B * 3 * (256 * 2 + 32) * 4);
    if (rc) return fail(c, SRH_ERR_HIP, "SAM decoder workspace allocation failed");
    float* keys = c->sd_keys.as<float>();
    f16* keys16 = c->sd_keys16.as<f16>();
    // token-side scratch (f32): q (queries), t0..t6 temporaries [R,256], h [R,2048], small [R,128] x 4, hyper
    float* tb = c->sd_tok.as<float>();
    float* q = tb;              float* t0 = tb + (size_t)R * 256; float* t1 = t0 + (size_t)R * 256; float* t2 = t1 + (size_t)R * 256;
    float* t3 = t2 + (size_t)R * 256; float* qn = t3 + (size_t)R * 256;   /* qn, qn+R*256: two more [R,256] */
    float* hid = tb + (size_t)R * 256 * 8;
    float* s0 = hid + (size_t)R * 2048; float* s1 = s0 + (size_t)R * 128; float* s2 = s1 + (size_t)R * 128; float* s3 = s2 + (size_t)R * 128;
    float* hy0 = s3 + (size_t)R * 128; float* hy1 = hy0 + (size_t)B * 3 * 256; float* hyper = hy1 + (size_t)B * 3 * 256;

    auto lin = [&](const float* x, int ldx, const float* xadd, int add_rows, const float* W, const float* b, int rows, int N, int K,
                   int act, float* y, int ldy) -> int {
        SdLinearParams lp;
        lp.x = x; lp.ldx = ldx; lp.xadd = xadd; lp.add_rows = add_rows; lp.W = W; lp.b = b; lp.rows = rows; lp.N = N; lp.K = K;
        lp.act = act; lp.y = y; lp.ldy = ldy;
        return launch_sd_tok_linear(lp, s);
    };
    auto img_gemm = [&](const char* cls, const f16* A, int K, const f16* W, int N, const float* bias, const float* pos,
                        const float* resid, int act, float* o32, f16* o16, size_t rows) -> int {
        GemmParams g;
        g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = (int)rows; g.N = N; g.K = K; g.bias = bias;
        g.pos = pos; g.pos_rows = HW; g.resid = resid; g.ldr = N; g.act = act;
        g.out_f32 = o32; g.ldc = N; g.out_f16 = o16; g.ldc16 = N;
        return gemm(c, cls, g, s);
    };
    // token -> image attention + residual + LayerNorm on the tokens: q <- LN(q + attn((q + pe_tok) Wq, (keys + pe) Wk, keys Wv) Wo)
    auto t2i = [&](const SdT2IW& a, const float* ng, const float* nb) -> int {
        TRY(img_gemm("sd_gemm", keys16, 256, a.k_w, 128, a.k_b, a.k_pos, nullptr, 0, nullptr, c->sd_k16.as<f16>(), T));
        TRY(img_gemm("sd_gemm", keys16, 256, a.v_w, 128, a.v_b, nullptr, nullptr, 0, nullptr, c->sd_v16.as<f16>(), T));
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, d.tokens, 4, a.q_w, a.q_b, R, 128, 256, 0, s0, 128));
        TRYK(c, "sd_attn", 0, 0, s, launch_sd_t2i_attn(s0, c->sd_k16.as<f16>(), c->sd_v16.as<f16>(), s1, B, HW, s));
        TRYK(c, "sd_token", 0, 0, s, lin(s1, 128, nullptr, 1, a.o_w, a.o_b, R, 256, 128, 0, t0, 256));
        TRYK(c, "sd_token", 0, 0, s, launch_sd_tok_ln(q, t0, ng, nb, q, R, s));
        return 0;
    };

    TRYK(c, "sd_prep", 0, (double)T * 256 * 10, s, launch_sd_add_channel(emb, d.no_mask, keys, keys16, T, s));
    // queries = the 4 output tokens, identical for every tile
    for (int b = 0; b < B; ++b)
        if (hipMemcpyAsync(q + (size_t)b * 1024, d.tokens, 4 * 256 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(c, SRH_ERR_HIP, "token broadcast failed");
    for (int l = 0; l < 2; ++l) {
        const SdLayerW& L = d.layer[l];
        // (1) self attention of the tokens; layer 0 skips the positional term and REPLACES the queries (skip_first_layer_pe)
        const float* pe_tok = l == 0 ? nullptr : d.tokens;
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, pe_tok, 4, L.self.q_w, L.self.q_b, R, 256, 256, 0, t0, 256));
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, pe_tok, 4, L.self.k_w, L.self.k_b, R, 256, 256, 0, t1, 256));
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, nullptr, 1, L.self.v_w, L.self.v_b, R, 256, 256, 0, t2, 256));
        TRYK(c, "sd_attn", 0, 0, s, launch_sd_tok_selfattn(t0, t1, t2, t3, B, s));
        TRYK(c, "sd_token", 0, 0, s, lin(t3, 256, nullptr, 1, L.self.o_w, L.self.o_b, R, 256, 256, 0, t0, 256));
        TRYK(c, "sd_token", 0, 0, s, launch_sd_tok_ln(t0, l == 0 ? nullptr : q, L.n1_g, L.n1_b, q, R, s));
        // (2) tokens attend to the image
        TRY(t2i(L.t2i, L.n2_g, L.n2_b));
        // (3) token MLP
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, nullptr, 1, L.l1_w, L.l1_b, R, 2048, 256, 2, hid, 2048));
        TRYK(c, "sd_token", 0, 0, s, lin(hid, 2048, nullptr, 1, L.l2_w, L.l2_b, R, 256, 2048, 0, t0, 256));
        TRYK(c, "sd_token", 0, 0, s, launch_sd_tok_ln(q, t0, L.n3_g, L.n3_b, q, R, s));
        // (4) image attends to the tokens: keys <- LN(keys + attn((keys + pe) Wq, (q + pe_tok) Wk, q Wv) Wo)
        TRY(img_gemm("sd_gemm", keys16, 256, L.i2t.q_w, 128, L.i2t.q_b, L.i2t.q_pos, nullptr, 0, nullptr, c->sd_k16.as<f16>(), T));
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, d.tokens, 4, L.i2t.k_w, L.i2t.k_b, R, 128, 256, 0, s2, 128));
        TRYK(c, "sd_token", 0, 0, s, lin(q, 256, nullptr, 1, L.i2t.v_w, L.i2t.v_b, R, 128, 256, 0, s3, 128));
        TRYK(c, "sd_attn", 0, 0, s, launch_sd_i2t_attn(c->sd_k16.as<f16>(), s2, s3, c->sd_a16.as<f16>(), B, HW, s));
        TRY(img_gemm("sd_gemm", c->sd_a16.as<f16>(), 128, L.i2t.o_w, 256, L.i2t.o_b, nullptr, keys, 0, c->sd_u0.as<float>(), nullptr, T));
        NormParams ln;
        ln.x = c->sd_u0.as<float>(); ln.M = (int)T; ln.D = 256; ln.eps = 1e-5f; ln.gamma = L.n4_g; ln.beta = L.n4_b;
        ln.out_f32 = keys; ln.out_f16 = keys16;
        TRYK(c, "layernorm", 0, (double)T * 256 * 10, s, launch_layernorm(ln, s));
    }
    TRY(t2i(d.fin, d.nf_g, d.nf_b));
    // output_upscaling: ConvT(256 -> 64) -> LayerNorm2d(64) -> GELU -> ConvT(64 -> 32) -> GELU, rows in quad-tree order
    TRY(img_gemm("sd_gemm", keys16, 256, d.up0_w, 256, d.up0_b, nullptr, nullptr, 0, c->sd_u0.as<float>(), nullptr, T));
    TRYK(c, "sd_prep", 0, (double)T * 256 * 6, s, launch_sd_ln64_gelu(c->sd_u0.as<float>(), d.up_ln_g, d.up_ln_b, c->sd_u0_16.as<f16>(), T * 4, s));
    TRY(img_gemm("sd_gemm", c->sd_u0_16.as<f16>(), 64, d.up1_w, 128, d.up1_b, nullptr, nullptr, 1, nullptr, c->sd_u1_16.as<f16>(), T * 4));
    // hyper-network MLPs on the three mask tokens (tokens 1..3 of every tile): [B,3,256] -> [B,3,32]
    for (int i = 0; i < 3; ++i) {
        TRYK(c, "sd_token", 0, 0, s, lin(q + (size_t)(1 + i) * 256, 4 * 256, nullptr, 1, d.hy_w[i][0], d.hy_b[i][0], B, 256, 256, 2, hy0 + (size_t)i * 256, 3 * 256));
        TRYK(c, "sd_token", 0, 0, s, lin(hy0 + (size_t)i * 256, 3 * 256, nullptr, 1, d.hy_w[i][1], d.hy_b[i][1], B, 256, 256, 2, hy1 + (size_t)i * 256, 3 * 256));
        TRYK(c, "sd_token", 0, 0, s, lin(hy1 + (size_t)i * 256, 3 * 256, nullptr, 1, d.hy_w[i][2], d.hy_b[i][2], B, 32, 256, 0, hyper + (size_t)i * 32, 3 * 32));
    }
    TRYK(c, "sd_prep", 0, (double)T * 16 * 72, s, launch_sd_mask(c->sd_u1_16.as<f16>(), hyper, c->sd_low.as<float>(), B, S, s));
    TRYK(c, "sd_prep", 0, (double)B * P * P * 16, s, launch_sd_upsample(c->sd_low.as<float>(), logits, scores, B, 4 * S, P, s));
    (void)qn; (void)t3;
    return 0;
}

static int encode_batch(srh_ctx* c, const srh_weights* w, PatchParams pp, int B, float* logits, float* scores,
                        float* emb, hipStream_t s) {
    const int S = w->S, D = w->D, heads = w->heads, hd = w->hd;
    const int T = B * S * S;
    TRY(ensure_encoder_ws(c, w, B));
    pp.B = B; pp.P = w->cfg.patch_size; pp.out = c->a0.as<f16>();
    TRYK(c, "patch_im2col", 0, (double)T * 768 * (pp.src_is_u8 ? 3 : 6), s, launch_patch_im2col(pp, s));
    // Residual stream: x stays fp32.  Where the persistent z192 GEMM applies (gemm_z192.hip: fp16 output only), proj / fc2
    // write their branch output (bias included) as fp16 into delta16 and the NEXT LayerNorm pass folds "x += delta" into
    // its read of x — the same HBM bytes as the GEMM-epilogue residual add, but moved out of the GEMM's exposed epilogue
    // into a streaming kernel.  Otherwise the GEMM epilogue adds the residual itself.
    // When BOTH branch GEMMs of a block go through z192, the attention branch (delta16) is not written back to x by the second
    // LayerNorm — it only normalises x + delta16 — and the next block's first LayerNorm folds both branches, (x + delta16) +
    // delta16b, and writes x once per block: 275 instead of 300 MB of LayerNorm traffic per block at B = 16, same sums in the
    // same order bit for bit.
    // Small-M models (ViT-H / ViT-L at 256 px): fc2 runs with split-K, and instead of a reduce pass its f32 partials stay in the
    // workspace for the next LayerNorm pass (or the neck's cast) to fold — x += (slice 0 + slice 1 + ...) + bias, the reduce kernel's
    // order, same bits — one launch and one round trip of x less per block.
    // Patch embedding: where z192 applies (ViT-B widths, enough tiles) it is just another fp16 branch output — conv + bias into delta16 —
    // and block 0's first LayerNorm pass computes the initial residual x = pos_embed[token] + delta16 (a row-modulo read of the
    // [S*S, D] table, NormParams::x_period) and writes x: the proj-shaped GEMM takes 22 instead of 51 us on the f32 + pos epilogue.
    bool pend_a = false, pend_b = false;                  // delta16 (patch embed / proj) / delta16b (fc2) hold a branch output not yet added to x
    bool x_is_pos = false;                                // x has not been written yet: its value is pos_embed (block 0's first pass)
    {
        GemmParams g;
        g.A = c->a0.as<f16>(); g.lda = 768; g.W = w->patch_w; g.ldw = 768; g.M = T; g.N = D; g.K = 768; g.bias = w->patch_b;
        GemmParams gz = g;
        gz.out_f16 = c->delta16.as<f16>(); gz.ldc16 = D;
        if (!w->blocks.empty() && z192_preferred(gz)) {
            TRY(gemm(c, "gemm_patch_embed", gz, s));
            pend_a = true; x_is_pos = true;
        } else {
            g.pos = w->pos; g.pos_rows = S * S; g.out_f32 = c->x.as<float>(); g.ldc = D;
            TRY(gemm(c, "gemm_patch_embed", g, s));
        }
    }
    int pend_slices = 0; const float* pend_bias = nullptr; // split-K partials of the last branch GEMM wait in c->split_ws
    auto branch_gemm = [&](const char* cls, const f16* A, int lda, const f16* W, int K, const float* bias, bool second, int a_blocked = 0) -> int {
        GemmParams gq;
        gq.A = A; gq.lda = lda; gq.W = W; gq.ldw = K; gq.M = T; gq.N = D; gq.K = K; gq.bias = bias; gq.a_blocked16 = a_blocked;
        gq.out_f16 = second ? c->delta16b.as<f16>() : c->delta16.as<f16>(); gq.ldc16 = D;
        if (z192_preferred(gq)) { (second ? pend_b : pend_a) = true; return gemm(c, cls, gq, s); }
        // Small-M models (ViT-L / ViT-H at 256 px), the attention branch: proj runs unsplit on 128 x 128 ring tiles with ~9 us of fixed cost
        // on a ~9 us loop, and its f32 read-modify-write of x sat in that exposed epilogue.  Its output goes out as fp16 instead (half the
        // store instructions) and the LayerNorm pass that follows anyway folds it into x — the same bytes, moved into the streaming kernel
        // (what the z192 path does for ViT-B).
        if (!second && !pend_a && !pend_slices && (D == 1024 || D == 1280) && gemm_splitk_factor(gq) <= 1) { pend_a = true; return gemm(c, cls, gq, s); }
        GemmParams gp = gq;
        gp.out_f16 = nullptr; gp.resid = c->x.as<float>(); gp.ldr = D; gp.out_f32 = c->x.as<float>(); gp.ldc = D;
        if (const int sk = gemm_splitk_factor(gp); sk > 1 && bias && (D == 1024 || D == 1280) && !pend_slices && !pend_a && !pend_b) {
            gp.defer_reduce = 1;
            pend_slices = sk; pend_bias = bias;
        }
        return gemm(c, cls, gp, s);
    };
    auto fold_pending = [&](NormParams& ln, int& reads) -> int {  // what the pass has to add to x before normalising / casting
        if (pend_slices && (pend_a || pend_b))            // never both: the fp16 branch would be dropped (branch_gemm defers only when neither is pending)
            return fail(c, SRH_ERR_HIP, "internal: split-K partials and an fp16 branch output pending at the same LayerNorm pass");
        if (pend_slices) {
            ln.slices = c->split_ws.as<float>(); ln.nslices = pend_slices; ln.slice_stride = (size_t)T * D; ln.slice_bias = pend_bias;
            reads = 2 * pend_slices;                      // in units of 2 bytes per element, as the fp16 branches
        } else if (pend_a && pend_b) { ln.delta16 = c->delta16.as<f16>(); ln.delta16b = c->delta16b.as<f16>(); reads = 2; }
        else if (pend_a) { ln.delta16 = c->delta16.as<f16>(); reads = 1; }
        else if (pend_b) { ln.delta16 = c->delta16b.as<f16>(); reads = 1; }
        return 0;
    };
    bool defer_x = false;                                 // both branch GEMMs of the blocks take z192 (same shapes in every block)
    {
        GemmParams gq;
        gq.M = T; gq.N = D; gq.K = D; gq.lda = D; gq.ldw = D; gq.ldc16 = D; gq.out_f16 = c->delta16.as<f16>();
        gq.A = c->attn16.as<f16>(); gq.W = w->blocks.empty() ? nullptr : w->blocks[0].proj_w; gq.bias = w->blocks.empty() ? nullptr : w->blocks[0].proj_b;
        GemmParams g2 = gq;
        g2.K = 4 * D; g2.lda = 4 * D; g2.ldw = 4 * D;
        defer_x = !w->blocks.empty() && z192_preferred(gq) && z192_preferred(g2);
    }
    // a LayerNorm pass over x (+ pending branches).  write_x: fold the pending branches into x for good.
    auto block_ln = [&](const float* gamma, const float* beta, bool write_x, int nf_tag) -> int {
        NormParams ln;
        ln.x = c->x.as<float>(); ln.M = T; ln.D = D; ln.eps = 1e-6f; ln.out_f16 = c->xn16.as<f16>();
        ln.gamma = gamma; ln.beta = beta; ln.nf = c->nf_dev; ln.nf_tag = std::min(nf_tag, NF_NECK - 1);
        if (x_is_pos) { ln.x = w->pos; ln.x_period = S * S; }
        int reads = 0;
        TRY(fold_pending(ln, reads));
        const bool wr = (write_x || pend_slices || x_is_pos) && reads > 0;   // partials cannot wait: the next split-K GEMM overwrites the workspace
        ln.x_out = wr ? c->x.as<float>() : nullptr;
        TRYK(c, "layernorm", 0, (double)T * D * (4 + 2 + 2 * reads + (wr ? 4 : 0)), s, launch_layernorm(ln, s));
        if (wr) { pend_a = pend_b = false; pend_slices = 0; x_is_pos = false; }
        return 0;
    };
    int blk = -1;
    for (const BlockW& b : w->blocks) {
        ++blk;
        TRY(block_ln(b.ln1_g, b.ln1_b, true, 2 * blk));
        GemmParams g;
        g.A = c->xn16.as<f16>(); g.lda = D; g.W = b.qkv_w; g.ldw = D; g.M = T; g.N = 3 * D; g.K = D;
        g.bias = b.qkv_b; g.out_f16 = c->qkv16.as<f16>(); g.ldc16 = 3 * D;
        TRY(gemm(c, "gemm_qkv", g, s));
        AttnParams ap;
        ap.table_h = b.rel_h; ap.table_w = b.rel_w;      // rel-pos bias derived inside the attention kernel (fused_relpos)
        ap.qkv = c->qkv16.as<f16>(); ap.ld = 3 * D; ap.bias_qkv = b.qkv_b16;
        ap.out = c->attn16.as<f16>(); ap.ldo = D; ap.B = B; ap.S = S; ap.heads = heads; ap.hd = hd; ap.win = b.win;
        ap.scale = 1.0f / sqrtf((float)hd);
        TRYK(c, b.win == S ? "attn_global" : "attn_window", attn_flops(B, S, heads, hd, b.win), 0, s, launch_attention(ap, s));
        TRY(branch_gemm("gemm_proj", c->attn16.as<f16>(), D, b.proj_w, D, b.proj_b, false));
        TRY(block_ln(b.ln2_g, b.ln2_b, !defer_x, 2 * blk + 1));
        GemmParams g1;
        g1.A = c->xn16.as<f16>(); g1.lda = D; g1.W = b.fc1_w; g1.ldw = D; g1.M = T; g1.N = 4 * D; g1.K = D;
        g1.bias = b.fc1_b; g1.act = 1; g1.out_f16 = c->hid16.as<f16>(); g1.ldc16 = 4 * D;
        // the hidden activation lives only between these two launches: when both take gemm_z192 it is kept in the blocked-16 layout
        // (fc1 stores 1 KiB contiguous per instruction, no LDS transposition; fc2's LDS-DMA reads it through per-lane addresses)
        int hid_blocked = 0;
        {
            GemmParams t1 = g1, t2;
            t1.out_blocked16 = 1;
            t2.A = c->hid16.as<f16>(); t2.lda = 4 * D; t2.W = b.fc2_w; t2.ldw = 4 * D; t2.M = T; t2.N = D; t2.K = 4 * D; t2.bias = b.fc2_b;
            t2.out_f16 = c->delta16b.as<f16>(); t2.ldc16 = D; t2.a_blocked16 = 1;
            hid_blocked = z192_preferred(t1) && z192_preferred(t2);
        }
        g1.out_blocked16 = hid_blocked;
        TRY(gemm(c, "gemm_fc1", g1, s));
        TRY(branch_gemm("gemm_fc2", c->hid16.as<f16>(), 4 * D, b.fc2_w, 4 * D, b.fc2_b, true, hid_blocked));
    }
    // neck: 1x1 conv -> LN2d -> 3x3 conv -> LN2d  (channels-last: LN2d is a row LN)
    {
        NormParams cast;
        cast.x = c->x.as<float>(); cast.M = T; cast.D = D; cast.out_f16 = c->xn16.as<f16>();
        int reads = 0;                                               // the last block's branch outputs, if still pending
        TRY(fold_pending(cast, reads));
        TRYK(c, "layernorm", 0, (double)T * D * (6 + 2 * reads), s, launch_layernorm(cast, s));
        GemmParams g;
        g.A = c->xn16.as<f16>(); g.lda = D; g.W = w->neck0_w; g.ldw = D; g.M = T; g.N = 256; g.K = D;
        g.out_f32 = c->n1.as<float>(); g.ldc = 256;
        TRY(gemm(c, "gemm_neck", g, s));
        NormParams ln;
        ln.x = c->n1.as<float>(); ln.M = T; ln.D = 256; ln.eps = 1e-6f; ln.gamma = w->neck1_g; ln.beta = w->neck1_b;
        ln.out_f16 = c->n1_16.as<f16>(); ln.nf = c->nf_dev; ln.nf_tag = NF_NECK;
        TRYK(c, "layernorm", 0, (double)T * 256 * 6, s, launch_layernorm(ln, s));
        GemmParams g3;
        g3.A = c->n1_16.as<f16>(); g3.lda = 256; g3.W = w->neck2_w; g3.ldw = 2304; g3.M = T; g3.N = 256; g3.K = 2304;
        g3.conv_S = S; g3.conv_C = 256; g3.out_f32 = c->n2.as<float>(); g3.ldc = 256;
        TRY(gemm(c, "gemm_neck", g3, s));
        ln.x = c->n2.as<float>(); ln.gamma = w->neck3_g; ln.beta = w->neck3_b;
        ln.out_f16 = c->emb16.as<f16>(); ln.out_f32 = emb; ln.nf_tag = NF_NECK + 1;
        TRYK(c, "layernorm", 0, (double)T * 256 * 10, s, launch_layernorm(ln, s));
    }
    if (!logits && !scores) return 0;
    if (w->cfg.use_sam_decoder) return sam_decode(c, w, B, emb, logits, scores, s);
    // map_decoder: all four ConvT layers + LayerNorm2d + GELUs + sigmoid + scatter in ONE kernel (decoder.hip), 8 MB in, the masks out
    {
        DecodeFusedParams dp;
        dp.emb16 = c->emb16.as<f16>(); dp.frags = w->dec_frags; dp.prm = w->dec_prm; dp.B = B; dp.S = S;
        dp.logits = logits; dp.scores = scores; dp.nf = c->nf_dev; dp.nf_tag = NF_DECODER;
        const double fl = 2.0 * T * (256.0 * 512 + 4 * 128.0 * 256 + 16 * 64.0 * 128 + 64 * 32.0 * 8);
        TRYK(c, "map_decoder", fl, (double)T * 512 + (double)T * 256 * ((logits ? 8 : 0) + (scores ? 8 : 0)), s, launch_decode_fused(dp, s));
    }
    return 0;
}

extern "C" int srh_encode_decode(srh_ctx* c, const srh_weights* w, const void* rgb, int rgb_dtype, int B,
                                 float* mask_logits, float* mask_scores, float* embeddings, void* stream) {
    if (!c || !w || !rgb || !embeddings || B <= 0) return fail(c, SRH_ERR_BAD_ARG, "srh_encode_decode: bad argument");
    if (rgb_dtype != SRH_F32 && rgb_dtype != SRH_U8) return fail(c, SRH_ERR_BAD_ARG, "rgb dtype must be f32 or u8");
    TRY(nonfinite_check(c, "srh_encode_decode"));          // lazily: what an EARLIER call's LayerNorm passes flagged (no sync here)
    hipSetDevice(c->device);
    PatchParams pp;
    pp.src = rgb; pp.src_is_u8 = rgb_dtype == SRH_U8;
    return encode_batch(c, w, pp, B, mask_logits, mask_scores, embeddings, (hipStream_t)stream);
}

// ---- TopoNet --------------------------------------------------------------------------------------------
static int toponet_impl(srh_ctx* c, const srh_weights* w, const float* embeddings, const void* points,
                        int points_dtype, const void* pairs, int pairs_dtype, const uint8_t* valid, int B, int N,
                        int Ns, int K, float* logits, float* scores, const int* point_tile, int n_tiles, long pair_base, void* stream) {
    if (!c || !w || !embeddings || !points || !pairs || !valid) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet: null argument");
    if (B <= 0 || N < 0 || Ns < 0) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet: bad sizes");
    if (K != 16) return fail(c, SRH_ERR_UNSUPPORTED, "n_pairs must be 16 (MAX_NEIGHBOR_QUERIES)");
    if (points_dtype != SRH_I64 && points_dtype != SRH_F32) return fail(c, SRH_ERR_BAD_ARG, "points dtype must be i64 or f32");
    if (pairs_dtype != SRH_I64 && pairs_dtype != SRH_I32) return fail(c, SRH_ERR_BAD_ARG, "pairs dtype must be i64 or i32");
    if (N == 0 || Ns == 0) return 0;
    hipSetDevice(c->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t NP = (size_t)B * N, R = (size_t)B * Ns * K;
    int rc = 0;
    rc |= c->t_feat16.ensure(NP * 256 * 2, 0.25);
    rc |= c->t_pf16.ensure(NP * 128 * 2, 0.25);
    rc |= c->t_pair16.ensure(R * 320 * 2, 0.25);
    if (rc) return fail(c, SRH_ERR_HIP, "toponet workspace allocation failed");

    SampleParams sp;
    sp.emb = embeddings; sp.points = points; sp.points_i64 = points_dtype == SRH_I64; sp.B = B; sp.N = N;
    sp.h = w->S; sp.w = w->S; sp.C = 256; sp.patch = (float)w->cfg.patch_size; sp.out_f16 = c->t_feat16.as<f16>();
    sp.point_tile = point_tile; sp.n_tiles = n_tiles;
    TRYK(c, "bilinear_sample", 0, (double)NP * 256 * 18, s, launch_sample(sp, s));
    GemmParams g;
    g.A = c->t_feat16.as<f16>(); g.lda = 256; g.W = w->tp_feat_w; g.ldw = 256; g.M = (int)NP; g.N = 128; g.K = 256;
    g.bias = w->tp_feat_b; g.act = 2; g.out_f16 = c->t_pf16.as<f16>(); g.ldc16 = 128;
    TRY(gemm(c, "gemm_toponet", g, s));
    PairGatherParams pg;
    pg.pf = c->t_pf16.as<f16>(); pg.points = points; pg.points_i64 = points_dtype == SRH_I64;
    pg.pairs = pairs; pg.pairs_i64 = pairs_dtype == SRH_I64; pg.B = B; pg.N = N; pg.Ns = Ns; pg.Kp = K;
    pg.zero_offset = w->cfg.toponet_version == 1; pg.out = c->t_pair16.as<f16>(); pg.ld = 320; pg.index_base = pair_base;
    TRYK(c, "pair_gather", 0, (double)R * (512 + 640), s, launch_pair_gather(pg, s));
    {
        // pair_proj + encoder layers + output_proj in one register-resident kernel (topo_fused.hip)
        TopoFusedParams tf;
        tf.pair = c->t_pair16.as<f16>(); tf.ld_pair = 320; tf.valid = valid; tf.stream = w->tp_stream; tf.params = w->tp_params;
        tf.nlayers = w->tp_layers; tf.nseq = B * Ns; tf.logits = logits; tf.scores = scores;
        const double fl = (double)R * (2.0 * 320 * 128 + tf.nlayers * (2.0 * 128 * 768 + 4.0 * 16 * 128) + 256);
        TRYK(c, "topo_fused", fl, 0, s, launch_topo_fused(tf, s));
        return 0;
    }
}

extern "C" int srh_toponet(srh_ctx* c, const srh_weights* w, const float* embeddings, const void* points,
                           int points_dtype, const void* pairs, int pairs_dtype, const uint8_t* valid, int B, int N,
                           int Ns, int K, float* logits, float* scores, void* stream) {
    return toponet_impl(c, w, embeddings, points, points_dtype, pairs, pairs_dtype, valid, B, N, Ns, K, logits, scores, nullptr, 0, 0, stream);
}

// The query rows of MANY tiles in one call, without padding every tile to the longest one of its batch: rows are the concatenated
// per-tile point lists (srh_pass2_pack_ragged), every point names the tile whose embeddings it samples, pairs index the flat list.
// The sampler, feature_proj, pair gather and the fused trunk treat every row on its own, so the scores are those of srh_toponet.
extern "C" int srh_toponet_ragged(srh_ctx* c, const srh_weights* w, const float* embeddings, int n_tiles, const float* points,
                                  const int32_t* point_tile, const int32_t* pairs, const uint8_t* valid, int64_t R, int K,
                                  const int64_t* tile_offsets, float* scores, void* stream) {
    if (!point_tile || !scores) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet_ragged: null argument");
    if (n_tiles <= 0) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet_ragged: n_tiles must be the number of tiles in `embeddings`");
    if (R < 0 || K != 16) return fail(c, R < 0 ? SRH_ERR_BAD_ARG : SRH_ERR_UNSUPPORTED, "srh_toponet_ragged: bad row count / n_pairs must be 16");
    // Workspace bound (the reference's pass 2 is bounded by INFER_BATCH_SIZE, inferencer.py:179-207): with the tiles' row offsets the
    // scene is scored in chunks of whole tiles of at most RAGGED_CHUNK_ROWS rows — rows are independent and a pair only names rows of
    // its own tile, so the chunks' scores are those of the one launch, bit for bit — and the pair workspace stays below ~210 MB
    // however large the scene (it grew with the scene before: 0.6 GB for a 48 k-row CityScale scene, ~10 GB for an 8192^2 one).
    // Without offsets the caller's rows go through ONE launch and must fit the same bound.
    constexpr int64_t RAGGED_CHUNK_ROWS = 16384;
    if (!tile_offsets) {
        if (R > 4 * RAGGED_CHUNK_ROWS) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet_ragged: more than 65536 rows need tile_offsets (chunked at tile boundaries)");
        return toponet_impl(c, w, embeddings, points, SRH_F32, pairs, SRH_I32, valid, 1, (int)R, (int)R, K, nullptr, scores, point_tile, n_tiles, 0, stream);
    }
    if (tile_offsets[0] != 0 || tile_offsets[n_tiles] != R) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet_ragged: tile_offsets must run from 0 to R");
    for (int t = 0; t < n_tiles; ++t)
        if (tile_offsets[t + 1] < tile_offsets[t]) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet_ragged: tile_offsets must ascend");
    for (int ta = 0; ta < n_tiles;) {
        int tb = ta + 1;                                                          // at least one tile (a tile above the bound is its own chunk)
        while (tb < n_tiles && tile_offsets[tb + 1] - tile_offsets[ta] <= RAGGED_CHUNK_ROWS) ++tb;
        const int64_t r0 = tile_offsets[ta], n = tile_offsets[tb] - r0;
        if (n > 0x7fffffffLL / (2 * K)) return fail(c, SRH_ERR_BAD_ARG, "srh_toponet_ragged: a single tile has too many rows");
        if (n > 0)
            TRY(toponet_impl(c, w, embeddings, points + 2 * r0, SRH_F32, pairs + 2 * (int64_t)K * r0, SRH_I32, valid + (int64_t)K * r0, 1, (int)n, (int)n,
                             K, nullptr, scores + (int64_t)K * r0, point_tile + r0, n_tiles, (long)r0, stream));
        ta = tb;
    }
    return 0;
}

// ---- scene level ------------------------------------------------------------------------------------------
extern "C" int srh_scene_pass1(srh_ctx* c, const srh_weights* w, const uint8_t* scene, int S, const int32_t* tile_xy,
                               int n_tiles, int B, float* canvas_kp, float* canvas_road, float* embeddings_all,
                               void* stream) {
    if (!c || !w || !scene || !tile_xy || !canvas_kp || !canvas_road || !embeddings_all)
        return fail(c, SRH_ERR_BAD_ARG, "srh_scene_pass1: null argument");
    if (n_tiles < 0 || B <= 0 || S < w->cfg.patch_size) return fail(c, SRH_ERR_BAD_ARG, "srh_scene_pass1: bad sizes");
    TRY(nonfinite_check(c, "srh_scene_pass1"));
    hipSetDevice(c->device);
    hipStream_t s = (hipStream_t)stream;
    const int P = w->cfg.patch_size;
    if (c->scores_ws.ensure((size_t)B * P * P * 2 * 4)) return fail(c, SRH_ERR_HIP, "scores workspace allocation failed");
    const size_t emb_per_tile = (size_t)w->S * w->S * 256;
    for (int off = 0; off < n_tiles; off += B) {
        const int nb = std::min(B, n_tiles - off);
        PatchParams pp;
        pp.src = scene; pp.src_is_u8 = 1; pp.scene_S = S; pp.tile_xy = tile_xy + 2 * off;
        TRY(encode_batch(c, w, pp, nb, nullptr, c->scores_ws.as<float>(), embeddings_all + emb_per_tile * off, s));
        TRYK(c, "scene_add", 0, (double)nb * P * P * 8 * 3, s,
             launch_scene_add(c->scores_ws.as<float>(), nb, P, tile_xy + 2 * off, canvas_kp, canvas_road, S, s));
    }
    return 0;
}

extern "C" int srh_scene_normalise(srh_ctx* c, const float* canvas_kp, const float* canvas_road, int S,
                                   const int32_t* tile_xy, int n_tiles, int P, uint8_t* kp_u8, uint8_t* road_u8,
                                   void* stream) {
    if (!c || !canvas_kp || !canvas_road || !tile_xy || !kp_u8 || !road_u8)
        return fail(c, SRH_ERR_BAD_ARG, "srh_scene_normalise: null argument");
    hipSetDevice(c->device);
    hipStream_t s = (hipStream_t)stream;
    if (c->counter.ensure((size_t)S * S * 4)) return fail(c, SRH_ERR_HIP, "counter allocation failed");
    TRYK(c, "scene_count", 0, (double)S * S * 4, s, launch_scene_count(c->counter.as<float>(), S, tile_xy, n_tiles, P, s));
    SceneNormParams np;
    np.canvas_kp = canvas_kp; np.canvas_road = canvas_road; np.counter = c->counter.as<float>();
    np.kp_u8 = kp_u8; np.road_u8 = road_u8; np.n = S * S;
    TRYK(c, "scene_normalise", 0, (double)S * S * 14, s, launch_scene_normalise(np, s));
    return 0;
}

// ---- op level ------------------------------------------------------------------------------------------------
extern "C" int srh_op_gemm_ex(srh_ctx* c, const void* A, const void* W, const float* bias, const float* resid, int M,
                              int N, int K, int act, float* out_f32, void* out_f16, int flags, void* stream) {
    if (!c || !A || !W) return fail(c, SRH_ERR_BAD_ARG, "srh_op_gemm: null argument");
    if (flags & ~(SRH_GEMM_A_BLOCKED16 | SRH_GEMM_OUT_BLOCKED16)) return fail(c, SRH_ERR_BAD_ARG, "srh_op_gemm_ex: unknown flag");
    hipSetDevice(c->device);
    GemmParams g;
    g.A = (const f16*)A; g.lda = K; g.W = (const f16*)W; g.ldw = K; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.resid = resid; g.ldr = N; g.act = act;
    g.out_f32 = out_f32; g.ldc = N; g.out_f16 = (f16*)out_f16; g.ldc16 = N;
    g.a_blocked16 = (flags & SRH_GEMM_A_BLOCKED16) != 0; g.out_blocked16 = (flags & SRH_GEMM_OUT_BLOCKED16) != 0;
    return gemm(c, "gemm_op", g, (hipStream_t)stream);
}

extern "C" int srh_op_gemm(srh_ctx* c, const void* A, const void* W, const float* bias, const float* resid, int M,
                           int N, int K, int act, float* out_f32, void* out_f16, void* stream) {
    return srh_op_gemm_ex(c, A, W, bias, resid, M, N, K, act, out_f32, out_f16, 0, stream);
}

extern "C" int srh_op_conv3x3(srh_ctx* c, const void* A, const void* W, int B, int S, int C, int N, float* out_f32,
                              void* stream) {
    if (!c || !A || !W || !out_f32) return fail(c, SRH_ERR_BAD_ARG, "srh_op_conv3x3: null argument");
    hipSetDevice(c->device);
    GemmParams g;
    g.A = (const f16*)A; g.lda = C; g.W = (const f16*)W; g.ldw = 9 * C; g.M = B * S * S; g.N = N; g.K = 9 * C;
    g.conv_S = S; g.conv_C = C; g.out_f32 = out_f32; g.ldc = N;
    return gemm(c, "gemm_op", g, (hipStream_t)stream);
}

extern "C" int srh_op_layernorm(srh_ctx* c, const float* x, const float* gamma, const float* beta, float eps, int M,
                                int D, int gelu, float* out_f32, void* out_f16, void* stream) {
    if (!c || !x) return fail(c, SRH_ERR_BAD_ARG, "srh_op_layernorm: null argument");
    hipSetDevice(c->device);
    NormParams ln;
    ln.x = x; ln.M = M; ln.D = D; ln.gamma = gamma; ln.beta = beta; ln.eps = eps; ln.act = gelu;
    ln.out_f32 = out_f32; ln.out_f16 = (f16*)out_f16;
    hipStream_t s = (hipStream_t)stream;
    TRYK(c, "layernorm", 0, 0, s, launch_layernorm(ln, s));
    return 0;
}

static int op_attention_impl(srh_ctx* c, const char* who, const void* qkv, const void* rel_h, const void* rel_w, const void* bias_qkv,
                             int B, int S, int heads, int hd, int win, void* out, void* stream) {
    if (!c || !qkv || !rel_h || !rel_w || !bias_qkv || !out) return fail(c, SRH_ERR_BAD_ARG, std::string(who) + ": null argument");
    if (hd != 64 && hd != 80) return fail(c, SRH_ERR_BAD_ARG, std::string(who) + ": head dim must be 64 or 80");
    hipSetDevice(c->device);
    hipStream_t s = (hipStream_t)stream;
    const int D = heads * hd;
    AttnParams ap;
    ap.table_h = (const f16*)rel_h; ap.table_w = (const f16*)rel_w;     // fused rel-pos bias (default)
    ap.qkv = (const f16*)qkv; ap.ld = 3 * D; ap.bias_qkv = (const f16*)bias_qkv;
    ap.out = (f16*)out; ap.ldo = D; ap.B = B; ap.S = S; ap.heads = heads; ap.hd = hd; ap.win = win;
    ap.scale = 1.0f / sqrtf((float)hd);
    TRYK(c, "attention", attn_flops(B, S, heads, hd, win), 0, s, launch_attention(ap, s));
    return 0;
}

extern "C" int srh_op_attention(srh_ctx* c, const void* qkv, const void* rel_h, const void* rel_w, const void* bias_qkv,
                                int B, int S, int heads, int win, void* out, void* stream) {
    return op_attention_impl(c, "srh_op_attention", qkv, rel_h, rel_w, bias_qkv, B, S, heads, 64, win, out, stream);
}

// head dim as an argument: 64 (ViT-B/L, attention.hip) or 80 (ViT-H, attention_hdx.hip)
extern "C" int srh_op_attention_hd(srh_ctx* c, const void* qkv, const void* rel_h, const void* rel_w, const void* bias_qkv,
                                   int B, int S, int heads, int hd, int win, void* out, void* stream) {
    return op_attention_impl(c, "srh_op_attention_hd", qkv, rel_h, rel_w, bias_qkv, B, S, heads, hd, win, out, stream);
}

// ---- profiling ------------------------------------------------------------------------------------------------
extern "C" int srh_profile_enable(srh_ctx* c, int on) {
    if (!c) return SRH_ERR_BAD_ARG;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    c->profiling = on != 0;
    c->prof.clear();
    c->ev_used = 0;
    return 0;
}

// one wave that spins until the 100 MHz constant-rate clock has advanced by `ticks`: a kernel of KNOWN duration
__global__ void srh_spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

// What an event pair around ONE launch adds beyond the time the kernel's waves run: median over 32 launches of (event-to-event
// time around a kernel whose single wave spins for exactly 50 us) - 50 us, measured on `stream` with the device otherwise idle
// (~3 us on MI355X: dispatch-to-first-wave, last-wave-to-completion and the marker packets).  Informational: rocprofv3 counts
// most of it as kernel duration too — bench.py's raw event times agree with `rocprofv3 --kernel-trace --stats` to ~1 %
// (profiles/r03_event_overhead_check.txt).
extern "C" int srh_profile_overhead(srh_ctx* c, void* stream, double* ms_per_launch) {
    if (!c || !ms_per_launch) return SRH_ERR_BAD_ARG;
    hipSetDevice(c->device);
    hipStream_t s = (hipStream_t)stream;
    const int N = 32;
    const unsigned long long ticks = 5000;                      // 50 us at the 100 MHz wall clock
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return fail(c, SRH_ERR_HIP, "hipEventCreate failed");
    for (int warm = 0; warm < 2; ++warm)
        for (int i = 0; i < N; ++i) {
            hipEventRecord(ev[2 * i], s);
            hipLaunchKernelGGL(srh_spin_kernel, dim3(1), dim3(64), 0, s, ticks);
            hipEventRecord(ev[2 * i + 1], s);
        }
    hipError_t e = hipStreamSynchronize(s);
    std::vector<float> d(N);
    for (int i = 0; i < N; ++i) hipEventElapsedTime(&d[i], ev[2 * i], ev[2 * i + 1]);
    for (auto& x : ev) hipEventDestroy(x);
    if (e != hipSuccess) return hip_fail(c, e, "srh_profile_overhead");
    std::sort(d.begin(), d.end());
    *ms_per_launch = std::max(0.0, (double)d[N / 2] - 0.050);
    return 0;
}

extern "C" int srh_profile_read(srh_ctx* c, srh_profile_row* rows, int max_rows, int* n_rows) {
    if (!c || !rows || !n_rows) return SRH_ERR_BAD_ARG;
    hipSetDevice(c->device);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(c, e, "hipDeviceSynchronize");
    std::vector<srh_profile_row> acc(c->cls_names.size());
    for (size_t i = 0; i < acc.size(); ++i) {
        memset(&acc[i], 0, sizeof(srh_profile_row));
        snprintf(acc[i].name, sizeof(acc[i].name), "%s", c->cls_names[i].c_str());
    }
    for (const ProfEntry& pe : c->prof) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, pe.e0, pe.e1);
        acc[pe.cls].launches += 1;
        acc[pe.cls].ms += ms;
        acc[pe.cls].flops += pe.flops;
        acc[pe.cls].bytes += pe.bytes;
    }
    int k = 0;
    for (size_t i = 0; i < acc.size() && k < max_rows; ++i)
        if (acc[i].launches) rows[k++] = acc[i];
    *n_rows = k;
    c->prof.clear();
    c->ev_used = 0;
    return 0;
}

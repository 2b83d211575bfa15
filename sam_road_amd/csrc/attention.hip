// SAM ViT attention with decomposed rel-pos bias: 14x14 windowed (8 of 12 ViT-B blocks) and global
// (blocks 2/5/8/11).  Replaces K5/K6 of SURVEY.md §2.1 (fork image_encoder.py Attention.forward +
// window_partition / window_unpartition), semantics per SURVEY App. B.3/B.4:
//   * softmax(scale * q.k^T + rel_h[q,kh] + rel_w[q,kw]) v, rel terms from the unscaled q;
//   * window padding happens AFTER LayerNorm, so pad tokens are REAL keys with k = b_k, v = b_v
//     (they receive softmax mass); pad QUERIES are never evaluated (their rows are cropped).
//
// gfx950 design.  Everything is issued transposed so that one lane owns one query:
//   S^T[key, q] = K[key,:] . Q[q,:]      A operand = K rows (LDS), B operand = Q rows (registers)
//   O^T[d,   q] = V^T[d,key] . P^T[key,q] A operand = V^T rows (LDS), B operand = P (registers)
// with v_mfma_f32_32x32x16_f16.  In the C/D layout lane l holds query (l & 31) and 16 of the 32 keys
// (rows (r&3) + 8(r>>2) + 4(l>>5)), so the softmax max/sum are lane-local plus ONE cross-half
// exchange, the running rescale is lane-local, and the exponentiated tile is ALREADY the B fragment
// of the P.V product if V^T is stored with the same key permutation — no P round trip through LDS.
// The rel-pos bias enters as the MFMA accumulator's initial value (f32, pre-divided by scale).
// Key tiles hold 32 MFMA rows = whole window rows (2 rows of 14 -> 28 valid keys, 2 rows of 16, or
// 1 row of 32), so the per-lane rel_w values repeat for every tile and rel_h is 1-2 scalars per tile.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace srh {

constexpr int HD = 64;  // head dim of ViT-B / ViT-L (ViT-H's 80: attention_hdx.hip, else attn_generic_kernel below)

template <int WIN> struct Geom;
template <> struct Geom<14> { static constexpr int KPT = 28, RPT = 2, NT = 7, WP = 16; };
template <> struct Geom<16> { static constexpr int KPT = 32, RPT = 2, NT = 8, WP = 16; };
template <> struct Geom<32> { static constexpr int KPT = 32, RPT = 1, NT = 32, WP = 32; };

// local MFMA row i of a key tile -> (row-in-tile, col) of the window
template <int WIN> __device__ __forceinline__ void tile_rc(int i, int& r, int& c) {
    if (WIN == 32) { r = 0; c = i; }
    else if (WIN == 16) { r = i >> 4; c = i & 15; }
    else { r = i >= 14; c = i - 14 * r; }
}

// V^T slot of local key i: the key permutation that makes exp(S^T) registers the P^T B-fragment
__device__ __forceinline__ int vt_slot(int i) {
    const int half = (i >> 2) & 1, reg = (i & 3) + 4 * (i >> 3);
    return ((reg >> 3) * 2 + half) * 8 + (reg & 7);
}

struct QState {
    f16x8 q[4];        // B fragments of the lane's query, 4 k-steps of 16
    f32x16 relw;       // rel_w / scale at the lane's 16 keys of a tile (tile-invariant): the S^T MFMA chain's C operand
    float m, l;        // running max (raw units) and this half's partial row sum
    f32x16 o[2];       // O^T accumulators, d tiles 0..31 / 32..63
};

// K fragments (A operand of S^T) of one key tile: 4 k-steps
__device__ __forceinline__ void read_kfrag(f16x8 (&kf)[4], const char* k_lds, int lane) {
    const int half = lane >> 5, row = lane & 31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        kf[ks] = *reinterpret_cast<const f16x8*>(k_lds + row * 128 + swz8(row, ks * 2 + half) * 16);
}

// One 32-row key tile.  The V^T fragment reads are issued right after the S^T MFMAs and BEFORE the softmax
// VALU block (order pinned with sched_barrier), so their LDS latency hides behind ~150 VALU instructions
// instead of stalling the P.V MFMAs — with only 2 waves per SIMD nothing else would cover it.
// The kernel is VALU-ISSUE bound (PMC: the two waves of a SIMD keep its issue port ~85 % busy at ~10 VALU instructions per
// score element, while dropping the exps or half the MFMAs changes nothing), so the tile is written for instruction count:
// rel_w enters as the C operand of the first MFMA (no accumulator init), rel_h — one or two scalars per tile — is folded
// into the row-max and into the addend of the exp argument's FMA, the row sum uses packed adds.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// row max of one key tile's 16 scores per lane, rel_h included (before the cross-half exchange).
// key rows of the lane's 16 scores: WIN 32: one window row per tile; WIN 16: r < 8 -> row 0, r >= 8 -> row 1;
// WIN 14: keys 0..13 / 14..27 -> r < 6 row 0, r = 6, 7 row `half`, r = 8..11 row 1, r >= 12: row 1 (half 0) / no key (half 1)
template <int WIN>
__device__ __forceinline__ float tile_max(const f32x16& s, float rh0, float rh1, int half) {
    if (WIN == 32) {
        float m = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, s[r]);
        return m + rh0;
    } else if (WIN == 16) {
        float ma = s[0], mb = s[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) { ma = fmaxf(ma, s[r]); mb = fmaxf(mb, s[8 + r]); }
        return fmaxf(ma + rh0, mb + rh1);
    } else {
        const float rhm = half ? rh1 : rh0;                          // rel_h of r = 6, 7
        float ma = s[0], mb = s[8];
#pragma unroll
        for (int r = 1; r < 6; ++r) ma = fmaxf(ma, s[r]);
#pragma unroll
        for (int r = 9; r < 12; ++r) mb = fmaxf(mb, s[r]);
        const float mt = fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]));
        mb = fmaxf(mb, half ? -INFINITY : mt);                       // rows 28..31 are not keys
        return fmaxf(fmaxf(ma + rh0, mb + rh1), fmaxf(s[6], s[7]) + rhm);
    }
}

// P^T = exp2(s * c_exp + rel_h * c_exp - m_new * c_exp) of one key tile as the B fragment of the P.V product (fp16), and its row sum
template <int WIN>
__device__ __forceinline__ void tile_exp(const f32x16& s, float rh0, float rh1, float m_new, float c_exp, int half,
                                         f16x8 (&pb)[2], f32x2& sum2) {
    const float mc = -m_new * c_exp;
    const float mc0 = fmaf(rh0, c_exp, mc), mc1 = WIN == 32 ? mc0 : fmaf(rh1, c_exp, mc), mcm = WIN == 14 ? (half ? mc1 : mc0) : mc0;
    const float mct = WIN == 14 ? (half ? -INFINITY : mc1) : mc1;  // r >= 12: exp2(-inf) = 0 for the rows that are not keys
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 pv;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rr = r + e;
            const float ad = WIN == 32 ? mc0 : WIN == 16 ? (rr >= 8 ? mc1 : mc0) : (rr < 6 ? mc0 : rr < 8 ? mcm : rr < 12 ? mc1 : mct);
            pv[e] = __builtin_amdgcn_exp2f(fmaf(s[rr], c_exp, ad));   // raw v_exp_f32: exp2(-inf) = 0
            pb[rr >> 3][rr & 7] = (f16)pv[e];
        }
        asm("v_pk_add_f32 %0, %0, %1" : "+v"(sum2) : "v"(pv));      // hipcc scalarises a plain f32x2 add here
    }
}

__device__ __forceinline__ void read_vfrag(f16x8 (&vf)[2][2], const char* vt_lds, int lane) {
    const int half = lane >> 5, row = lane & 31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int d = dt * 32 + row;
            const int c = (sx * 2 + half) ^ ((d >> 2) & 3);
            vf[dt][sx] = *reinterpret_cast<const f16x8*>(vt_lds + d * 64 + c * 16);
        }
}

// running max / rescale shared by the one- and two-tile forms: returns the new max (raw units)
__device__ __forceinline__ float update_max(QState& st, float mloc, float c_exp) {
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(st.m, mloc);
    // rescale the running state only when some lane's max moved (wave-uniform branch; after the first
    // few key tiles the max is usually stable and the 32 accumulator multiplies are skipped)
    if (__any(m_new != st.m)) {
        const float alpha = __builtin_amdgcn_exp2f((st.m - m_new) * c_exp);
        st.l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
        st.m = m_new;
    }
    return m_new;
}

template <int WIN>
__device__ __forceinline__ void attn_tile(QState& st, const f16x8 (&kf)[4], const char* vt_lds,
                                          float rh0, float rh1, float c_exp, int lane) {
    const int half = lane >> 5;
    f32x16 s = mfma32(kf[0], st.q[0], st.relw);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) s = mfma32(kf[ks], st.q[ks], s);
    f16x8 vf[2][2];
    read_vfrag(vf, vt_lds, lane);
    __builtin_amdgcn_sched_barrier(0);
    const float m_new = update_max(st, tile_max<WIN>(s, rh0, rh1, half), c_exp);
    f32x2 sum2 = {0.f, 0.f};
    f16x8 pb[2];
    tile_exp<WIN>(s, rh0, rh1, m_new, c_exp, half, pb, sum2);
    st.l += sum2[0] + sum2[1];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) st.o[dt] = mfma32(vf[dt][sx], pb[sx], st.o[dt]);
}

// TWO 32-row key tiles at once (round 3).  A wave's key loop was bound by the dependency CHAIN of one tile, not by any
// pipe: four dependent S^T MFMAs (64 clk of latency each at 32 clk of issue), the 16-deep row-max tree, the cross-half
// exchange (an LDS round trip), the exps, then two 2-deep P.V chains — about 1000 clk end to end with one more wave per SIMD to
// fill the gaps (2470 clk per tile and wave measured, MFMA busy 21 %, VALU ~50 %).  With two tiles in flight the two S^T chains
// interleave (the matrix pipe is paced instead of waiting on its own result), the two max trees are independent, there is ONE
// exchange / rescale decision / row-sum update per 64 keys, twice as many independent exps per dependency level, and the
// eight P.V MFMAs alternate between the two O^T accumulators.  Same arithmetic as two attn_tile calls except that both tiles
// are exponentiated against the max over all 64 keys.
template <int WIN>
__device__ __forceinline__ void attn_tile2(QState& st, const f16x8 (&kfa)[4], const f16x8 (&kfb)[4], const char* vta, const char* vtb,
                                           float rh0a, float rh1a, float rh0b, float rh1b, float c_exp, int lane) {
    const int half = lane >> 5;
    f32x16 sa = mfma32(kfa[0], st.q[0], st.relw);
    f32x16 sb = mfma32(kfb[0], st.q[0], st.relw);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) { sa = mfma32(kfa[ks], st.q[ks], sa); sb = mfma32(kfb[ks], st.q[ks], sb); }
    f16x8 vfa[2][2], vfb[2][2];
    read_vfrag(vfa, vta, lane);
    read_vfrag(vfb, vtb, lane);
    __builtin_amdgcn_sched_barrier(0);
    const float m_new = update_max(st, fmaxf(tile_max<WIN>(sa, rh0a, rh1a, half), tile_max<WIN>(sb, rh0b, rh1b, half)), c_exp);
    f32x2 sum2 = {0.f, 0.f};
    f16x8 pa[2], pb[2];
    tile_exp<WIN>(sa, rh0a, rh1a, m_new, c_exp, half, pa, sum2);
    tile_exp<WIN>(sb, rh0b, rh1b, m_new, c_exp, half, pb, sum2);
    st.l += sum2[0] + sum2[1];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) st.o[dt] = mfma32(vfa[dt][sx], pa[sx], st.o[dt]);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) st.o[dt] = mfma32(vfb[dt][sx], pb[sx], st.o[dt]);
}

template <int WIN>
__device__ __forceinline__ void load_query(QState& st, const AttnParams& p, size_t tok, int head, int lane) {
    const int half = lane >> 5;
    const f16* q = p.qkv + tok * p.ld + head * HD;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st.q[ks] = *reinterpret_cast<const f16x8*>(q + (ks * 2 + half) * 8);
    st.m = -INFINITY;
    st.l = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.o[dt][r] = 0.f;
}

// Fused decomposed rel-pos bias (replaces the separate relpos kernel and its [tokens, heads, 2*Wp] f32 round trip
// through HBM).  For the wave's 32 queries:  P^T[j, q] = T[j, :] . Q[q, :]  over ALL 2*WIN-1 table rows j (one or two
// 32-row MFMA tiles, A operand = table rows straight from L2, B operand = the query fragments already in registers),
// then  rel[q, k] = P[q, qc - k + WIN - 1] / scale  is scattered into the wave's LDS table buf[q][k] (row stride STRIDE
// floats).  The w table goes first: its 16 per-lane values (tile-invariant) are read back into st.relw, then the same
// buffer is overwritten with the h table, which the key loop reads one or two scalars per tile.
template <int WIN, int STRIDE>
__device__ __forceinline__ void fused_relpos(QState& st, const AttnParams& p, int qy, int qx, float* buf, int lane) {
    constexpr int NTJ = (2 * WIN - 1 + 31) / 32;
    const int half = lane >> 5, row = lane & 31;
    const float inv_scale = 1.0f / p.scale;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {            // 0: w table -> st.relw, 1: h table -> buf
        const f16* table = pass == 0 ? p.table_w : p.table_h;
        const int qc = pass == 0 ? qx : qy;
#pragma unroll
        for (int jt = 0; jt < NTJ; ++jt) {
            const int j = jt * 32 + row;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 a;
                if (j < 2 * WIN - 1) a = *reinterpret_cast<const f16x8*>(table + (size_t)j * HD + (ks * 2 + half) * 8);
                else for (int e = 0; e < 8; ++e) a[e] = (f16)0.f;
                acc = mfma32(a, st.q[ks], acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = qc - (jt * 32 + mfma32_row(r, lane)) + WIN - 1;
                if (k >= 0 && k < WIN) buf[row * STRIDE + k] = acc[r] * inv_scale;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (pass == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rr, cc;
                tile_rc<WIN>(mfma32_row(r, lane), rr, cc);
                st.relw[r] = (cc < WIN) ? buf[row * STRIDE + cc] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ __forceinline__ void store_query(const QState& st, const AttnParams& p, size_t tok, int head,
                                            int lane, bool valid) {
    const int half = lane >> 5;
    const float inv = 1.0f / (st.l + __shfl_xor(st.l, 32, 64));
    if (!valid) return;
    f16* o = p.out + tok * p.ldo + head * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (f16)(st.o[dt][qd * 4 + e] * inv);
            *reinterpret_cast<f16x4*>(o + dt * 32 + 8 * qd + 4 * half) = h;
        }
}

// Stage a 4-key x 8-dim block of V into the transposed, key-permuted V^T tile.
__device__ __forceinline__ uint32_t u4_word(const uint4& v, int i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}
__device__ __forceinline__ uint32_t u4_half(const uint4& v, int e) {   // e-th fp16 of the 8
    const uint32_t w = u4_word(v, e >> 1);
    return (e & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __forceinline__ void vt_write(char* vt, int kq, int dc, const uint4& v0, const uint4& v1,
                                         const uint4& v2, const uint4& v3) {
    const int slot = vt_slot(kq * 4);
    const int c = slot >> 3, eo = slot & 7;   // eo is 0 or 4
    // 4 keys x 8 dims -> 8 dims x 4 keys: every output word pairs the same fp16 of two keys = ONE v_perm_b32
    // (byte select 0x05040100: low halves, 0x07060302: high halves) instead of shift / and / or chains
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int d = dc * 8 + e;
        const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
        uint2 w;
        w.x = __builtin_amdgcn_perm(u4_word(v1, e >> 1), u4_word(v0, e >> 1), sel);
        w.y = __builtin_amdgcn_perm(u4_word(v3, e >> 1), u4_word(v2, e >> 1), sel);
        *reinterpret_cast<uint2*>(vt + d * 64 + ((c ^ ((d >> 2) & 3)) * 16) + eo * 2) = w;
    }
}

// ---------------------------------------------------------------------------------------------
// Windowed attention: one workgroup per (image, head, window); the whole window's K and V^T
// (196 keys incl. pad keys) are staged once, then each wave walks its 32-query tiles.
// (Round 2 tried a PERSISTENT 8-wave form — one workgroup per CU, one wave per query tile, the next item's K / V / q prefetched
// into registers during the key loop: correct but 48 % SLOWER, 85 vs 57 us per launch, profiles/r02_attn_window_p8.txt.  A
// query tile's key loop costs ~6.5 us of VALU-issue-bound work, so an edge / corner window (2 / 1 tiles) parks 6 / 7 of the 8
// waves for a whole tile time; with two independent 4-wave workgroups per CU the idle waves of one leave their SIMD's issue
// slots to the other, and one workgroup's prologue overlaps the other's key loop.  History: git log.)
// ---------------------------------------------------------------------------------------------
constexpr int WIN_LDS_K = 7 * 4096, WIN_LDS_VT = 7 * 4096, WIN_LDS_RH = 4 * 32 * 17 * 4;
constexpr int WIN_LDS = WIN_LDS_K + WIN_LDS_VT + WIN_LDS_RH;

__global__ __launch_bounds__(256, 2) void attn_window_kernel(AttnParams p) {
    constexpr int WIN = 14;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* vt_lds = smem + WIN_LDS_K;
    float* rh_lds = reinterpret_cast<float*>(smem + WIN_LDS_K + WIN_LDS_VT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S, nw = (S + WIN - 1) / WIN, D = p.heads * HD;
    int u = blockIdx.x;
    const int head = u % p.heads; u /= p.heads;
    const int widx = u % (nw * nw); u /= (nw * nw);
    const int b = u;
    const int wy = widx / nw, wx = widx % nw;
    const int nry = min(WIN, S - wy * WIN), nrx = min(WIN, S - wx * WIN);
    const int nreal = nry * nrx;
    const int ntq = (nreal + 31) / 32;
    const int half = lane >> 5;
    const bool fused = p.ablate != 3;

    // ---- Every global load of the workgroup is issued up front, so their latencies overlap ONCE: the K / V rows to
    // stage, the query fragments of this wave's (up to two) query tiles and the rel-pos table fragments.  (Measured
    // before: the key loop was 23 % of the kernel; staging, per-tile query / table loads and their exposed latencies
    // were the rest.)
    // K items: thread (chunk c, local row i) of every tile t = 0..6
    const int s_c = tid & 7, s_i = (tid >> 3) & 31;
    uint4 kreg[7];
    {
        int rr, cc;
        tile_rc<WIN>(s_i < 28 ? s_i : 0, rr, cc);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int y = wy * WIN + t * 2 + rr, x = wx * WIN + cc;
            const f16* src = (y < S && x < S) ? p.qkv + (((size_t)b * S + y) * S + x) * p.ld + D + head * HD
                                              : p.bias_qkv + D + head * HD;
            kreg[t] = *reinterpret_cast<const uint4*>(src + s_c * 8);
        }
    }
    // V items: thread (d chunk dc, key quad kq) of tiles t = wave and wave + 4
    const int s_dc = tid & 7, s_kq = (tid >> 3) & 7;
    uint4 vreg[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = wave + 4 * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = s_kq * 4 + e;
            int rr, cc;
            tile_rc<WIN>(i < 28 ? i : 0, rr, cc);
            const int y = wy * WIN + min(t, 6) * 2 + rr, x = wx * WIN + cc;
            const f16* src = (y < S && x < S) ? p.qkv + (((size_t)b * S + y) * S + x) * p.ld + 2 * D + head * HD
                                              : p.bias_qkv + 2 * D + head * HD;
            vreg[j][e] = *reinterpret_cast<const uint4*>(src + s_dc * 8);
        }
    }
    // query fragments of this wave's tiles jt = wave, wave + 4
    f16x8 qpre[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qi = min((wave + 4 * j) * 32 + (lane & 31), nreal - 1);
        const size_t tokq = ((size_t)b * S + wy * WIN + qi / nrx) * S + wx * WIN + qi % nrx;
        const f16* q = p.qkv + tokq * p.ld + head * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qpre[j][ks] = *reinterpret_cast<const f16x8*>(q + (ks * 2 + half) * 8);
    }
    // rel-pos table fragments (A operand rows j = lane & 31 of the 27-row tables)
    f16x8 tfrag[2][4];
    if (fused) {
        const int j = min(lane & 31, 2 * WIN - 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tfrag[0][ks] = *reinterpret_cast<const f16x8*>(p.table_w + (size_t)j * HD + (ks * 2 + half) * 8);
            tfrag[1][ks] = *reinterpret_cast<const f16x8*>(p.table_h + (size_t)j * HD + (ks * 2 + half) * 8);
        }
    }

    // ---- stage K rows and the transposed, key-permuted V^T tiles
    if (p.ablate != 2) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const uint4 v = s_i < 28 ? kreg[t] : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(k_lds + t * 4096 + s_i * 128 + swz8(s_i, s_c) * 16) = v;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = wave + 4 * j;
            if (t < 7) {
                const uint4 z = make_uint4(0, 0, 0, 0);
                const int i0 = s_kq * 4;
                vt_write(vt_lds + t * 4096, s_kq, s_dc, i0 < 28 ? vreg[j][0] : z, i0 + 1 < 28 ? vreg[j][1] : z,
                         i0 + 2 < 28 ? vreg[j][2] : z, i0 + 3 < 28 ? vreg[j][3] : z);
            }
        }
    }
    __syncthreads();

    const float c_exp = p.scale * 1.4426950408889634f;
    const float inv_scale = 1.0f / p.scale;
    float* rh = rh_lds + wave * 32 * 17;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int jt = wave + 4 * j;
        if (jt >= ntq) break;
        const int qi_raw = jt * 32 + (lane & 31);
        const bool valid = qi_raw < nreal;
        const int qi = valid ? qi_raw : nreal - 1;
        const int ry = qi / nrx, rx = qi % nrx;
        const size_t tok = ((size_t)b * S + wy * WIN + ry) * S + wx * WIN + rx;
        QState st;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st.q[ks] = qpre[j][ks];
        st.m = -INFINITY;
        st.l = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[dt][r] = 0.f;
        if (fused) {
            // fused rel-pos bias (see fused_relpos): w table -> st.relw, then h table -> rh
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int qc = pass == 0 ? rx : ry;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = mfma32(tfrag[pass][ks], st.q[ks], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jrow = mfma32_row(r, lane);
                    const int k = qc - jrow + WIN - 1;
                    if (k >= 0 && k < WIN && jrow < 2 * WIN - 1) rh[(lane & 31) * 17 + k] = acc[r] * inv_scale;
                }
                __builtin_amdgcn_wave_barrier();
                if (pass == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int rr, cc;
                        tile_rc<WIN>(mfma32_row(r, lane), rr, cc);
                        st.relw[r] = (cc < WIN) ? rh[(lane & 31) * 17 + cc] : 0.f;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (p.ablate != 1) {
            // tiles 0..5 as three PAIRS (attn_tile2: two independent S^T chains / max trees in flight, one exchange and rescale
            // decision per 56 keys), then tile 6.  The next pair's K fragments are read right behind the current pair's call: by then
            // its S^T MFMAs have consumed the registers, and the reads' latency hides behind the pair's softmax
            f16x8 kfA[4], kfB[4];
            read_kfrag(kfA, k_lds, lane);
            read_kfrag(kfB, k_lds + 4096, lane);
            const float* rhq = rh + (lane & 31) * 17;
#pragma unroll 1
            for (int t = 0; t < 6; t += 2) {
                const float rh0a = rhq[2 * t], rh1a = rhq[2 * t + 1], rh0b = rhq[2 * t + 2], rh1b = rhq[2 * t + 3];
                attn_tile2<WIN>(st, kfA, kfB, vt_lds + t * 4096, vt_lds + (t + 1) * 4096, rh0a, rh1a, rh0b, rh1b, c_exp, lane);
                read_kfrag(kfA, k_lds + (t + 2) * 4096, lane);
                if (t < 4) read_kfrag(kfB, k_lds + (t + 3) * 4096, lane);
            }
            attn_tile<WIN>(st, kfA, vt_lds + 6 * 4096, rhq[12], rhq[13], c_exp, lane);
        }
        store_query(st, p, tok, head, lane, valid);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// Global attention: one workgroup per (image, head, 128-query block); K / V^T streamed through a
// double-buffered LDS ring, 2 key tiles (64 MFMA rows) per stage, next stage's global loads issued
// before the current stage's MFMAs.
// ---------------------------------------------------------------------------------------------
template <int WIN>
__global__ __launch_bounds__(256, 2) void attn_global_kernel(AttnParams p) {
    constexpr int NT = Geom<WIN>::NT, WP = Geom<WIN>::WP, RPT = Geom<WIN>::RPT;
    constexpr int STAGE = 2 * 4096 + 2 * 4096;   // 2 K tiles + 2 V^T tiles
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 4 * 32 * (WP + 1) * 4];
    float* rh_lds = reinterpret_cast<float*>(smem + 2 * STAGE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S, D = p.heads * HD;
    const int nqb = (S * S) / 128;
    // XCD-aware order (speed only): workgroup u runs on XCD u % 8; give one XCD all nqb query blocks of an
    // (image, head) back to back so K / V of that head are fetched from HBM once and re-read from its L2.
    int qb, bh;
    {
        const int nbh = p.B * p.heads;
        const int u = blockIdx.x, xcd = u & 7, j = u >> 3;
        if ((nbh & 7) == 0) { bh = (j / nqb) * 8 + xcd; qb = j % nqb; }
        else { bh = u / nqb; qb = u % nqb; }
    }
    const int head = bh % p.heads, b = bh / p.heads;
    const size_t tok0 = (size_t)b * S * S;

    const int qi = qb * 128 + wave * 32 + (lane & 31);
    const size_t tok = tok0 + qi;
    QState st;
    load_query<WIN>(st, p, tok, head, lane);
    float* rh = rh_lds + wave * 32 * (WP + 1);
    fused_relpos<WIN, WP + 1>(st, p, qi / S, qi % S, rh, lane);

    // staging registers (named, not arrays: hipcc keeps lambda-captured staging arrays in scratch):
    // 2 K chunks per thread, one 4-key x 8-dim V block for threads < 128
    uint4 rk0, rk1, rv0, rv1, rv2, rv3;
    const int s_c = tid & 7, s_i = (tid >> 3) & 31;                 // K item: chunk, row (tile = e)
    const int s_dc = tid & 7, s_kq = (tid >> 3) & 7, s_tl = tid >> 6;  // V item (tid < 128)
    // Staging loads as buffer loads: the per-thread offsets are loop-invariant VGPRs and the stage advances a scalar
    // offset, so the key loop carries no address arithmetic on the (binding) VALU.  Offsets are relative to the image's
    // first token: S*S*ld*2 bytes < 2^31 for every supported S.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qkv + tok0 * p.ld), 0, 0x7fffffff, 0x00020000);
    const int ldb = p.ld * 2;
    const int ko0 = s_i * ldb + (D + head * HD + s_c * 8) * 2, ko1 = ko0 + 32 * ldb;
    const int vo0 = (s_tl * 32 + s_kq * 4) * ldb + (2 * D + head * HD + s_dc * 8) * 2;
    const int vo1 = vo0 + ldb, vo2 = vo0 + 2 * ldb, vo3 = vo0 + 3 * ldb;
#define SRH_LD128(dst, vo, so) { const u32x4 t_ = __builtin_amdgcn_raw_buffer_load_b128(rsq, vo, so, 0); dst = make_uint4(t_[0], t_[1], t_[2], t_[3]); }
#define SRH_LOAD_STAGE(sidx) { const int so_ = (sidx) * 64 * ldb; \
        SRH_LD128(rk0, ko0, so_) SRH_LD128(rk1, ko1, so_) \
        if (tid < 128) { SRH_LD128(rv0, vo0, so_) SRH_LD128(rv1, vo1, so_) SRH_LD128(rv2, vo2, so_) SRH_LD128(rv3, vo3, so_) } }
#define SRH_STORE_STAGE(buf) { char* base_ = smem + (buf) * STAGE; \
        *reinterpret_cast<uint4*>(base_ + s_i * 128 + swz8(s_i, s_c) * 16) = rk0; \
        *reinterpret_cast<uint4*>(base_ + 4096 + s_i * 128 + swz8(s_i, s_c) * 16) = rk1; \
        if (tid < 128) vt_write(base_ + 8192 + s_tl * 4096, s_kq, s_dc, rv0, rv1, rv2, rv3); }

    const float c_exp = p.scale * 1.4426950408889634f;
    constexpr int NSTAGE = NT / 2;
    SRH_LOAD_STAGE(0)
    SRH_STORE_STAGE(0)
    __syncthreads();
    // two stages per trip so that the LDS buffer index is a compile-time constant: every ds_read / ds_write address is then
    // a loop-invariant lane offset plus an immediate (no per-access address VALU in the issue-bound key loop)
    static_assert(NSTAGE % 2 == 0, "the key loop is unrolled by two stages");
    const float* rhp = rh + (lane & 31) * (WP + 1);
#define SRH_STAGE(sidx, buf) { \
        const int snext_ = (sidx) + 1 < NSTAGE ? (sidx) + 1 : (sidx);   /* last stage re-loads itself (no branch) */ \
        if (p.ablate != 2 && p.ablate != 8) SRH_LOAD_STAGE(snext_) \
        const char* base = smem + (buf) * STAGE; \
        if (p.ablate != 1) { \
            f16x8 kfA[4], kfB[4]; \
            read_kfrag(kfA, base, lane); \
            read_kfrag(kfB, base + 4096, lane); \
            const float rh0a = rhp[((buf) * 2) * RPT], rh1a = RPT == 2 ? rhp[((buf) * 2) * RPT + 1] : 0.f; \
            const float rh0b = rhp[((buf) * 2 + 1) * RPT], rh1b = RPT == 2 ? rhp[((buf) * 2 + 1) * RPT + 1] : 0.f; \
            attn_tile2<WIN>(st, kfA, kfB, base + 8192, base + 8192 + 4096, rh0a, rh1a, rh0b, rh1b, c_exp, lane); \
        } \
        if (p.ablate != 2 && p.ablate != 8) SRH_STORE_STAGE((buf) ^ 1) \
        if (p.ablate != 8) __syncthreads(); }
    for (int sidx = 0; sidx < NSTAGE; sidx += 2) {
        SRH_STAGE(sidx, 0)
        SRH_STAGE(sidx + 1, 1)
        rhp += 4 * RPT;
    }
    store_query(st, p, tok, head, lane, true);
}

// ---------------------------------------------------------------------------------------------
// Generic attention for head dims other than 64 (ViT-H: 80).  Correctness-first fallback: one thread
// owns one query (q and the output row live in f32 registers), keys are streamed through LDS in chunks
// of 128 shared by the workgroup's 256 queries, scores / online softmax / P.V on the VALU in f32.
// Same semantics as the MFMA kernels: pad tokens of a window are real keys with k = b_k, v = b_v, pad
// queries are skipped, the rel-pos bias comes from the unscaled q and the tables.
// grid = (image, head, window, block of 256 real queries).
// ---------------------------------------------------------------------------------------------
constexpr int GEN_KC = 128;
template <int NCH>      // head dim / 8 (compile time: 10 for ViT-H's 80, 8 for 64)
__global__ __launch_bounds__(256) void attn_generic_kernel(AttnParams p) {
    constexpr int hd = NCH * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int win = p.win, S = p.S, nw = (S + win - 1) / win, D = p.heads * hd;
    const int KC = win > 32 ? GEN_KC / 2 : GEN_KC;                    // keys per LDS chunk (the rel table grows with win)
    f16* k_lds = reinterpret_cast<f16*>(smem);                        // [KC][hd] fp16 (consumed by v_dot2_f32_f16)
    float* v_lds = reinterpret_cast<float*>(k_lds + KC * hd);         // [KC][hd] f32 (converted once, at staging)
    float* rel_lds = v_lds + KC * hd;                                 // [256][2 * win]
    const int tid = threadIdx.x;
    const int nqb = (win * win + 255) / 256;
    int u = blockIdx.x;
    const int qb = u % nqb; u /= nqb;
    const int head = u % p.heads; u /= p.heads;
    const int widx = u % (nw * nw); u /= (nw * nw);
    const int b = u;
    const int wy = widx / nw, wx = widx % nw;
    const int nry = min(win, S - wy * win), nrx = min(win, S - wx * win);
    const int nreal = nry * nrx;
    const int qi = qb * 256 + tid;
    const bool active = qi < nreal;
    const int ry = active ? qi / nrx : 0, rx = active ? qi % nrx : 0;
    const size_t tok = ((size_t)b * S + wy * win + ry) * S + wx * win + rx;
    f16x2 q2[NCH * 4];                                                // the query as fp16 pairs (exactly what the MFMA path multiplies)
    float o[hd];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (active) t = *reinterpret_cast<const f16x8*>(p.qkv + tok * p.ld + head * hd + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) q2[c * 4 + e] = f16x2{t[2 * e], t[2 * e + 1]};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[c * 8 + e] = 0.f;
    }
    // rel-pos bias of this query: rel[k] = q . T[qc - k + win - 1]   (fp16 products, f32 accumulation)
    float* rel = rel_lds + tid * 2 * win;
    for (int k = 0; k < win; ++k) {
        const f16* th = p.table_h + (size_t)(ry - k + win - 1) * hd;
        const f16* tw = p.table_w + (size_t)(rx - k + win - 1) * hd;
        float ah = 0.f, aw = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(th + c * 8), w8 = *reinterpret_cast<const f16x8*>(tw + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ah = __builtin_amdgcn_fdot2(q2[c * 4 + e], f16x2{a[2 * e], a[2 * e + 1]}, ah, false);
                aw = __builtin_amdgcn_fdot2(q2[c * 4 + e], f16x2{w8[2 * e], w8[2 * e + 1]}, aw, false);
            }
        }
        rel[k] = ah; rel[win + k] = aw;
    }
    float m = -INFINITY, l = 0.f;
    const int nkeys = win * win;
    for (int k0 = 0; k0 < nkeys; k0 += KC) {
        const int kc = min(KC, nkeys - k0);
        __syncthreads();
        for (int it = tid; it < kc * NCH; it += 256) {                // stage K (fp16) / V (f32) chunk, 8 dims per item
            const int c = it % NCH, kl = it / NCH, kk = k0 + kl;
            const int y = wy * win + kk / win, x = wx * win + kk % win;
            const bool real = y < S && x < S;
            const f16* src = real ? p.qkv + (((size_t)b * S + y) * S + x) * p.ld + head * hd : p.bias_qkv + head * hd;
            *reinterpret_cast<uint4*>(k_lds + kl * hd + c * 8) = *reinterpret_cast<const uint4*>(src + D + c * 8);
            const f16x8 v8 = *reinterpret_cast<const f16x8*>(src + 2 * D + c * 8);
            *reinterpret_cast<f32x4*>(v_lds + kl * hd + c * 8) = f32x4{(float)v8[0], (float)v8[1], (float)v8[2], (float)v8[3]};
            *reinterpret_cast<f32x4*>(v_lds + kl * hd + c * 8 + 4) = f32x4{(float)v8[4], (float)v8[5], (float)v8[6], (float)v8[7]};
        }
        __syncthreads();
        if (!active) continue;
        for (int j = 0; j < kc; ++j) {                                // every thread reads the same key row: LDS broadcast
            const int kk = k0 + j;
            float sdot = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f16x8 kv = *reinterpret_cast<const f16x8*>(k_lds + j * hd + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) sdot = __builtin_amdgcn_fdot2(q2[c * 4 + e], f16x2{kv[2 * e], kv[2 * e + 1]}, sdot, false);
            }
            const float sc = sdot * p.scale + rel[kk / win] + rel[win + kk % win];
            if (sc > m) {                                             // rare after the first keys: rescale the running state
                const float alpha = __expf(m - sc);
                l *= alpha;
#pragma unroll
                for (int d = 0; d < hd; ++d) o[d] *= alpha;
                m = sc;
            }
            const float pe = __expf(sc - m);
            l += pe;
#pragma unroll
            for (int c = 0; c < NCH * 2; ++c) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(v_lds + j * hd + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[c * 4 + e] = fmaf(pe, vv[e], o[c * 4 + e]);
            }
        }
    }
    if (!active) return;
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (f16)(o[c * 8 + e] * inv);
        *reinterpret_cast<f16x8*>(p.out + tok * p.ldo + head * hd + c * 8) = t;
    }
}

int launch_attention(const AttnParams& p_in, hipStream_t s) {
    AttnParams p = p_in;
#ifdef SRH_TUNING      // probe builds only (tools/probes/build_probes.sh): ablation switches change the RESULT
    static const int env_abl = getenv("SRH_ATTN_ABL") ? atoi(getenv("SRH_ATTN_ABL")) : 0;
    if (!p.ablate) p.ablate = env_abl;
#else
    p.ablate = 0;
#endif
    const bool mfma_path = p.hd == HD && (p.win == 14 || (p.win == p.S && (p.S == 16 || p.S == 32)));
    static const bool use_hdx = !(getenv("SRH_ATTN_HDX") && atoi(getenv("SRH_ATTN_HDX")) == 0);
    if (!mfma_path && use_hdx && attention_hdx_supported(p)) return launch_attention_hdx(p, s);
    if (!mfma_path) {      // other head dims / windows (ViT-H at 512 px, the 64x64 global window of 1024-pixel tiles)
        if ((p.hd != 64 && p.hd != 80) || !p.table_h || !p.table_w || p.win > 64) return -2;
        const int nw = (p.S + p.win - 1) / p.win, nqb = (p.win * p.win + 255) / 256;
        const int lds = (p.win > 32 ? GEN_KC / 2 : GEN_KC) * p.hd * 6 + 256 * 2 * p.win * 4;
        if (lds > 160 * 1024) return -2;
        static bool gattr = false;
        if (!gattr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_generic_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_generic_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            gattr = true;
        }
        const dim3 grid(p.B * nw * nw * p.heads * nqb);
        if (p.hd == 64) hipLaunchKernelGGL(attn_generic_kernel<8>, grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL(attn_generic_kernel<10>, grid, dim3(256), lds, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    if (p.win == p.S) {
        const int grid = p.B * p.heads * (p.S * p.S / 128);
        if (p.S == 32) hipLaunchKernelGGL(attn_global_kernel<32>, dim3(grid), dim3(256), 0, s, p);
        else if (p.S == 16) hipLaunchKernelGGL(attn_global_kernel<16>, dim3(grid), dim3(256), 0, s, p);
        else return -2;
    } else if (p.win == 14) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_window_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS);
            attr_set = true;
        }
        const int nw = (p.S + 13) / 14;
        hipLaunchKernelGGL(attn_window_kernel, dim3(p.B * nw * nw * p.heads), dim3(256), WIN_LDS, s, p);
    } else {
        return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

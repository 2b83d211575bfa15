// SAM ViT attention with decomposed rel-pos bias: 14x14 windowed (8 of 12 ViT-B blocks) and global
// (blocks 2/5/8/11).  Replaces K5/K6 of SURVEY.md §2.1 (fork image_encoder.py Attention.forward +
// window_partition / window_unpartition), semantics per SURVEY App. B.3/B.4:
//   * softmax(scale * q.k^T + rel_h[q,kh] + rel_w[q,kw]) v, rel terms from the unscaled q;
//   * window padding happens AFTER LayerNorm, so pad tokens are REAL keys with k = b_k, v = b_v
//     (they receive softmax mass); pad QUERIES are never evaluated (their rows are cropped).
//
// gfx950 design.  Everything is issued transposed so that one lane owns one query:
//   S^T[key, q] = K[key,:] . Q[q,:]      A operand = K rows (LDS), B operand = Q rows (registers)
//   O^T[d,   q] = V^T[d,key] . P^T[key,q] A operand = V^T fragments (row-major V rows in LDS, read transposed
//                                          with ds_read_b64_tr_b16), B operand = P (registers)
// with v_mfma_f32_32x32x16_f16.  In the C/D layout lane l holds query (l & 31) and 16 of the 32 keys
// (rows (r&3) + 8(r>>2) + 4(l>>5)), so the softmax max/sum are lane-local plus ONE cross-half
// exchange (v_permlane32_swap), the running rescale is lane-local, and the exponentiated tile is ALREADY the B
// fragment of the P.V product if the V^T fragment gathers its keys in the same order — no P round trip through LDS.
// (attention_hdx.hip, head dim 80, still stages a transposed, key-permuted V^T image with v_perm.)
// The rel-pos bias enters as the MFMA accumulator's initial value (f32, pre-divided by scale).
// Key tiles hold 32 MFMA rows = whole window rows (2 rows of 14 -> 28 valid keys, 2 rows of 16, or
// 1 row of 32), so the per-lane rel_w values repeat for every tile and rel_h is 1-2 scalars per tile.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

// Per-phase shader-clock counters of attn_window_kernel (ablate 9 of tools/probes/attn_win_probe): empty in the product
#ifdef SRH_TUNING
#define WK_PHASE_BEGIN unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define WK_PHASE(k) if (p.ablate == 9) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; }
#define WK_PHASE_END if (p.ablate == 9 && lane == 0) { unsigned long long* d_ = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.scratch) + ATTN_SCRATCH_BYTES) + ((size_t)(blockIdx.x & 255) * 8 + wave) * 8; \
        for (int k_ = 0; k_ < 8; ++k_) atomicAdd(d_ + k_, tph[k_]); }
#else
#define WK_PHASE_BEGIN
#define WK_PHASE(k)
#define WK_PHASE_END
#endif

namespace srh {

constexpr int HD = 64;  // head dim of ViT-B / ViT-L (ViT-H's 80: attention_hdx.hip, else attn_generic_kernel below)

template <int WIN> struct Geom;
template <> struct Geom<14> { static constexpr int KPT = 28, RPT = 2, NT = 7, WP = 16; };
template <> struct Geom<16> { static constexpr int KPT = 32, RPT = 2, NT = 8, WP = 16; };
template <> struct Geom<32> { static constexpr int KPT = 32, RPT = 1, NT = 32, WP = 32; };
// 64 x 64 (the global window of 1024-px tiles): a key tile is HALF a window row, a ring stage (two tiles) one whole row — the
// lane's rel_w values differ between the even and the odd tile of a stage (QState::relw / relw_b), rel_h is one scalar per stage
template <> struct Geom<64> { static constexpr int KPT = 32, RPT = 1, NT = 128, WP = 64; };

// local MFMA row i of a key tile -> (row-in-tile, col) of the window
template <int WIN> __device__ __forceinline__ void tile_rc(int i, int& r, int& c) {
    if (WIN >= 32) { r = 0; c = i; }
    else if (WIN == 16) { r = i >> 4; c = i & 15; }
    else { r = i >= 14; c = i - 14 * r; }
}

struct QState {
    f16x8 q[4];        // B fragments of the lane's query, 4 k-steps of 16
    f32x16 relw;       // rel_w / scale at the lane's 16 keys of a tile (tile-invariant): the S^T MFMA chain's C operand
    f32x16 relw_b;     // 64 x 64 window only: the same for the odd tile of a stage (window columns 32..63); unused (no registers) otherwise
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
    if (WIN >= 32) {
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
    const float mc0 = fmaf(rh0, c_exp, mc), mc1 = WIN >= 32 ? mc0 : fmaf(rh1, c_exp, mc), mcm = WIN == 14 ? (half ? mc1 : mc0) : mc0;
    const float mct = WIN == 14 ? (half ? -INFINITY : mc1) : mc1;  // r >= 12: exp2(-inf) = 0 for the rows that are not keys
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 pv;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rr = r + e;
            const float ad = WIN >= 32 ? mc0 : WIN == 16 ? (rr >= 8 ? mc1 : mc0) : (rr < 6 ? mc0 : rr < 8 ? mcm : rr < 12 ? mc1 : mct);
            pv[e] = __builtin_amdgcn_exp2f(fmaf(s[rr], c_exp, ad));   // raw v_exp_f32: exp2(-inf) = 0
            pb[rr >> 3][rr & 7] = (f16)pv[e];
        }
        asm("v_pk_add_f32 %0, %0, %1" : "+v"(sum2) : "v"(pv));      // hipcc scalarises a plain f32x2 add here
    }
}

// running max / rescale shared by the one- and two-tile forms: returns the new max (raw units)
// cross-half exchanges use v_permlane32_swap_b32 (one VALU instruction) instead of a ds_bpermute round trip through LDS
__device__ __forceinline__ float update_max(QState& st, float mloc, float c_exp) {
    {   // max over both halves: after the swap r[0] = {lo, lo}, r[1] = {hi, hi}
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const float m_new = fmaxf(st.m, mloc);
    // rescale the running state only when some lane's max moved (wave-uniform branch).  (A LAZY rule — move the reference only when
    // a row maximum has outgrown it by 2^8, P up to 2^8 — was measured in round 4: with 32 rows per tile some maximum moves in
    // nearly every 64-key step, yet skipping the 17 packed multiplies bought nothing measurable, profiles/r04_attention_asm_global.txt.)
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
__device__ __forceinline__ void fused_relpos(QState& st, const AttnParams& p, int qy, int qx, float* buf, const char* tbl_w, const char* tbl_h, int lane) {
    constexpr int NTJ = (2 * WIN - 1 + 31) / 32;
    static_assert(STRIDE > WIN, "slot WIN of a row is the dump slot");
    const int half = lane >> 5, row = lane & 31;
    const float inv_scale = 1.0f / p.scale;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {            // 0: w table -> st.relw, 1: h table -> buf
        const int qc = pass == 0 ? qx : qy;
#pragma unroll
        for (int jt = 0; jt < NTJ; ++jt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // A operand: table rows j = 32 jt + (lane & 31), staged in LDS in the K-tile format (tile jt of tbl_w / tbl_h; rows past the table's
            // end hold a copy of its last row: their products land in the dump slot).  They used to be read straight from L2 — sixteen
            // 16-byte loads per lane that touch 32 table rows each, ~87 ticks of the CU's address path per instruction; the staged form
            // costs the workgroup 2 NTJ x 4 LDS-DMA pieces in all (profiles/r05_attention_global.txt)
            f16x8 a[4];
            read_kfrag(a, (pass == 0 ? tbl_w : tbl_h) + jt * 4096, lane);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mfma32(a[ks], st.q[ks], acc);
            // P^T[j, q] -> rel[q][k = qc - j + WIN - 1]: one subtract, one unsigned min and an unconditional write per element (k < 0
            // and k >= WIN — also the rows j > 2 WIN - 2 — land in the dump slot WIN) instead of two compares and an exec-masked write
            const unsigned kb = (unsigned)(qc + WIN - 1 - jt * 32 - 4 * half);
            float* rowp = buf + row * STRIDE;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned k = kb - (unsigned)((r & 3) + 8 * (r >> 2));
                rowp[min(k, (unsigned)WIN)] = acc[r] * inv_scale;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (pass == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rr, cc;
                tile_rc<WIN>(mfma32_row(r, lane), rr, cc);
                st.relw[r] = (cc < WIN) ? buf[row * STRIDE + cc] : 0.f;
                if (WIN == 64) st.relw_b[r] = buf[row * STRIDE + 32 + cc];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The lane of query q in half h holds O^T rows d = 32 dt + 8 qd + 4 h + e (4 consecutive dims per register quad).  One v_permlane32_swap
// per pair of quads gives half 0 the 8 consecutive dims 16 k .. 16 k + 7 and half 1 the dims 16 k + 8 .. 16 k + 15 of its query: 4 stores
// of 16 bytes per lane instead of 8 of 8 bytes.  The store path is what this buys: a wave's store instruction touches 32 token rows either
// way (lane = query), ~87 ticks of the CU's address path each (profiles/r04_store_probe.txt) — half as many instructions.
__device__ __forceinline__ void pack_out(const QState& st, float inv, int half, f16x8 (&h)[4]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // register quads qd = 2 k (dims 16 k + 4 h + e) and qd = 2 k + 1 (dims 16 k + 8 + 4 h + e), as packed fp16 pairs
            unsigned a0, a1, b0, b1;
            {
                const f16x2 p0 = {(f16)(st.o[dt][8 * k + 0] * inv), (f16)(st.o[dt][8 * k + 1] * inv)};
                const f16x2 p1 = {(f16)(st.o[dt][8 * k + 2] * inv), (f16)(st.o[dt][8 * k + 3] * inv)};
                const f16x2 p2 = {(f16)(st.o[dt][8 * k + 4] * inv), (f16)(st.o[dt][8 * k + 5] * inv)};
                const f16x2 p3 = {(f16)(st.o[dt][8 * k + 6] * inv), (f16)(st.o[dt][8 * k + 7] * inv)};
                a0 = __builtin_bit_cast(unsigned, p0); a1 = __builtin_bit_cast(unsigned, p1);
                b0 = __builtin_bit_cast(unsigned, p2); b1 = __builtin_bit_cast(unsigned, p3);
            }
            // swap(x, y): lanes 32..63 of x <-> lanes 0..31 of y.  x = quad 2 k, y = quad 2 k + 1: afterwards half 0 holds
            // (own quad 2 k | half 1's quad 2 k) = dims 16 k + 0..7, half 1 holds (half 0's quad 2 k + 1 | own quad 2 k + 1) = dims 16 k + 8..15
            const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            const unsigned w0 = half ? r0[1] : r0[0], w1 = half ? r1[1] : r1[0];       // the lane's own quad: first (half 0) or second (half 1)
            const unsigned x0 = half ? r0[0] : r0[1], x1 = half ? r1[0] : r1[1];       // the quad received from the other half
            const unsigned lo0 = half ? x0 : w0, lo1 = half ? x1 : w1, hi0 = half ? w0 : x0, hi1 = half ? w1 : x1;
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ v = {lo0, lo1, hi0, hi1};
            h[dt * 2 + k] = __builtin_bit_cast(f16x8, v);
        }
}

__device__ __forceinline__ void store_query(const QState& st, const AttnParams& p, size_t tok, int head,
                                            int lane, bool valid) {
    const int half = lane >> 5;
    float lsum;
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(st.l), __float_as_uint(st.l), false, false);
        lsum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float inv = __builtin_amdgcn_rcpf(lsum);               // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
    f16x8 h[4];
    pack_out(st, inv, half, h);                                  // cross-lane: before the exit of the lanes without a query
    if (!valid) return;
    f16* o = p.out + tok * p.ldo + head * HD;
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<f16x8*>(o + c * 16 + 8 * half) = h[c];
}

// ---------------------------------------------------------------------------------------------
// Windowed attention: one workgroup per (image, head, window); the whole window's K and V (196 keys incl. pad keys) are
// staged once, then each wave walks its 32-query tiles.
//  * K rows and ROW-MAJOR V rows go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 key rows per wave
//    instruction): no staging registers, no ds_write, no v_perm transposition.  The DMA destination is lane-linear, so the
//    swizzles live on the SOURCE side: LDS slot (row i, 16-B position c') receives chunk c' ^ ((i >> 1) & 7) of the K row
//    (read_kfrag's swz8) and chunk c' ^ 4 * ((i >> 1) & 1) of the V row.
//  * The P.V A fragments (V^T[d, key]) are read from the row-major V image with ds_read_b64_tr_b16: a 16-lane group hands
//    in the addresses of a [4 keys][16 dims] block (lane i: key i / 4, dims 4 (i % 4) ..), lane c of the group receives
//    dims c of the 4 keys.  Two reads give the 8 keys of a fragment; which 4-key groups is dictated by the S^T C layout
//    (keys 16 sx + 4 half + 0..3 and the same + 8), so that exp(S^T) stays the B fragment as before.  The V swizzle makes
//    the 32-lane phase of a read (4 keys x 64 B) touch each of the 64 banks once.
//  * MFMA rows 28..31 of a tile are not keys: their K / V slots get a copy of key 0 of the tile (finite values, P = 0).
//  * Windows with one or two query tiles (the edge / corner windows: 5 of the 9 windows of a 512-px tile, 196 keys each
//    because pad tokens are real keys) used to park two or three of the four waves for the whole key loop.  They now split
//    the KEY tiles over the waves ([0,4) | [4,7) for two query tiles, [0,2) | [2,4) | [4,6) | [6,7) for one) and merge the
//    partial (max, sum, O) through LDS with the usual online-softmax rescale.
// ---------------------------------------------------------------------------------------------
typedef __fp16 h4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int WIN_LDS_K = 7 * 4096, WIN_LDS_V = 7 * 4096, WIN_LDS_TBL = 2 * 4096, WIN_LDS_RH = 4 * 32 * 17 * 4;
constexpr int WIN_LDS = WIN_LDS_K + WIN_LDS_V + WIN_LDS_TBL + WIN_LDS_RH;     // 74 240 B: two workgroups per CU

template <int AUX = 0>      // AUX 2: nontemporal
__device__ __forceinline__ void dma16(const void* src, char* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, AUX);
}

__device__ __forceinline__ f16x8 tr_read8(const char* a) {
    struct Pair { h4v lo, hi; } pr;
    pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4v*)a);
    pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4v*)(a + 1024));   // keys + 8
    return __builtin_bit_cast(f16x8, pr);
}

// lane bases of the transposed V reads: d tile 0 / d tile 1 (the swizzle XORs bit 2 of the chunk, so dt is not additive)
__device__ __forceinline__ void vtr_bases(int lane, int& vb0, int& vb1) {
    const int i = lane & 15, g = (lane >> 4) & 1, half = lane >> 5;
    const int sw = (i >> 3) & 1;                                   // ((key >> 1) & 1) with key = 4 * group + i / 4
    vb0 = (4 * half + (i >> 2)) * 128 + (((g * 2 + ((i >> 1) & 1)) ^ (4 * sw)) * 16) + (i & 1) * 8;
    vb1 = vb0 ^ 64;
}

__device__ __forceinline__ void read_vfrag(f16x8 (&vf)[2][2], const char* v_tile, int vb0, int vb1) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) vf[dt][sx] = tr_read8(v_tile + (dt ? vb1 : vb0) + sx * 2048);
}

template <int WIN>
__device__ __forceinline__ void attn_tile(QState& st, const f16x8 (&kf)[4], const char* v_tile, int vb0, int vb1,
                                             float rh0, float rh1, float c_exp, int lane) {
    const int half = lane >> 5;
    f32x16 s = mfma32(kf[0], st.q[0], st.relw);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) s = mfma32(kf[ks], st.q[ks], s);
    f16x8 vf[2][2];
    read_vfrag(vf, v_tile, vb0, vb1);
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
__device__ __forceinline__ void attn_tile2(QState& st, const f16x8 (&kfa)[4], const f16x8 (&kfb)[4], const char* va, const char* vb,
                                              int vb0, int vb1, float rh0a, float rh1a, float rh0b, float rh1b, float c_exp, int lane) {
    const int half = lane >> 5;
    f32x16 sa = mfma32(kfa[0], st.q[0], st.relw);
    f32x16 sb = mfma32(kfb[0], st.q[0], WIN == 64 ? st.relw_b : st.relw);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) { sa = mfma32(kfa[ks], st.q[ks], sa); sb = mfma32(kfb[ks], st.q[ks], sb); }
    f16x8 vfa[2][2], vfb[2][2];
    read_vfrag(vfa, va, vb0, vb1);
    read_vfrag(vfb, vb, vb0, vb1);
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

// key tiles [t0, t1) of a 14 x 14 window for one query tile: pairs, then a single tile if the count is odd
__device__ __forceinline__ void window_keys(QState& st, const char* k_lds, const char* v_lds, const float* rhq, int t0, int t1,
                                            int vb0, int vb1, float c_exp, int lane) {
    constexpr int WIN = 14;
    f16x8 kfA[4], kfB[4];
    int t = t0;
    read_kfrag(kfA, k_lds + t * 4096, lane);
    if (t + 1 < t1) read_kfrag(kfB, k_lds + (t + 1) * 4096, lane);
#pragma unroll 1
    for (; t + 1 < t1; t += 2) {
        const float rh0a = rhq[2 * t], rh1a = rhq[2 * t + 1], rh0b = rhq[2 * t + 2], rh1b = rhq[2 * t + 3];
        attn_tile2<WIN>(st, kfA, kfB, v_lds + t * 4096, v_lds + (t + 1) * 4096, vb0, vb1, rh0a, rh1a, rh0b, rh1b, c_exp, lane);
        if (t + 2 < t1) read_kfrag(kfA, k_lds + (t + 2) * 4096, lane);
        if (t + 3 < t1) read_kfrag(kfB, k_lds + (t + 3) * 4096, lane);
    }
    if (t < t1) attn_tile<WIN>(st, kfA, v_lds + t * 4096, vb0, vb1, rhq[2 * t], rhq[2 * t + 1], c_exp, lane);
}

// fused decomposed rel-pos bias of one 32-query tile of a 14 x 14 window (see fused_relpos): w table -> st.relw, h table -> rh
__device__ __forceinline__ void window_relpos(QState& st, const char* tbl_lds, int rx, int ry, float* rh, float inv_scale, int lane) {
    constexpr int WIN = 14;
    f16x8 tfrag[2][4];                                 // A operand rows j = lane & 31 of the 27-row tables, staged in LDS in the K-tile format
    read_kfrag(tfrag[0], tbl_lds, lane);
    read_kfrag(tfrag[1], tbl_lds + 4096, lane);
    const int half = lane >> 5;
    float* row = rh + (lane & 31) * 17;                // rel[q][1 + k], k = 0..13; slots 0 and 15 collect the table rows that map outside the window
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int qc = pass == 0 ? rx : ry;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = mfma32(tfrag[pass][ks], st.q[ks], acc);
        // P^T[j, q] -> rel[q][k = qc - j + 13]: one subtract, one clamp (v_med3) and an unconditional write per element instead of
        // three compares and an exec-masked write; table rows 27..31 (copies of row 26) land on k < 0, i.e. in the dump slot
        const int kb = qc + WIN - 1 - 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kb - ((r & 3) + 8 * (r >> 2));
            row[min(max(k, -1), WIN) + 1] = acc[r] * inv_scale;
        }
        __builtin_amdgcn_wave_barrier();
        if (pass == 0) {
            // the lane's 16 keys of a tile are window columns {0..3, 8..11, 2..5, 10..13} (half 0) / {4..7, 12, 13, 0, 1, 6..9, none} (half 1):
            // lane base + immediate for every read
            const float* rd = row + 1 + 4 * half;
            const float* rd67 = row + 1 + (half ? 0 : 10);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                constexpr int c0[16] = {0, 1, 2, 3, 8, 9, 10, 11, 2, 3, 4, 5, 10, 11, 12, 13};
                float v = (r == 6 || r == 7) ? rd67[r - 6] : rd[c0[r]];
                if (r >= 12) v = half ? 0.f : v;               // MFMA rows 28..31 are not keys
                st.relw[r] = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

struct WinItem { int b, head, wy, wx, nry, nrx, nreal, ntq; };

__device__ __forceinline__ WinItem win_item(const AttnParams& p, int n, int nw, int nbh) {
    WinItem it;
    // item order: window shapes heaviest first — the nf x nf full windows, then the 2 nf edge windows (one partial dimension), then the
    // corner (both partial); inside a shape the (image, head) pairs.  A CU's list is dealt round-robin from this order.
    const int wi = n / nbh, r = n - wi * nbh;
    const int nf = p.S / 14, k = wi - nf * nf;
    it.b = r / p.heads; it.head = r - it.b * p.heads;
    if (k < 0) { it.wy = wi / nf; it.wx = wi - it.wy * nf; }
    else if (k < nf) { it.wy = k; it.wx = nf; }
    else if (k < 2 * nf) { it.wy = nf; it.wx = k - nf; }
    else { it.wy = nf; it.wx = nf; }
    it.nry = min(14, p.S - it.wy * 14); it.nrx = min(14, p.S - it.wx * 14);
    it.nreal = it.nry * it.nrx; it.ntq = (it.nreal + 31) / 32;
    return it;
}

__global__ __launch_bounds__(256, 2) void attn_window_kernel(AttnParams p) {
    constexpr int WIN = 14;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + WIN_LDS_K;
    char* tbl_lds = smem + WIN_LDS_K + WIN_LDS_V;                       // rel-pos tables w | h as two more "K tiles"
    float* rh_lds = reinterpret_cast<float*>(smem + WIN_LDS_K + WIN_LDS_V + WIN_LDS_TBL);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S, nw = (S + WIN - 1) / WIN, D = p.heads * HD;
    // workgroups in win_item's order: the heaviest window shapes first, so that the launch's last round of workgroups is the light ones
    const WinItem it = win_item(p, blockIdx.x, nw, p.B * p.heads);
    const int b = it.b, head = it.head, wy = it.wy, wx = it.wx;
    const int nry = min(WIN, S - wy * WIN), nrx = min(WIN, S - wx * WIN);
    const int nreal = nry * nrx;
    const int ntq = (nreal + 31) / 32;
    const int half = lane >> 5;
    WK_PHASE_BEGIN
    // work split (workgroup-uniform): ntq <= 2 -> the waves share query tiles and split the key tiles
    const bool split = ntq <= 2;
    const int nsplit = split ? 4 / ntq : 1;
    const int part = split ? wave / ntq : 0;
    const int jt0 = split ? wave % ntq : wave;
    int t0 = 0, t1 = 7;
    if (nsplit == 2) { t0 = part ? 4 : 0; t1 = part ? 7 : 4; }
    else if (nsplit == 4) { t0 = 2 * part; t1 = min(2 * part + 2, 7); }

    // ---- Loads: the rel-pos tables (LDS-DMA), the query fragments, then the K / V rows (LDS-DMA, wave w fills MFMA rows
    // 8w .. 8w+7 of every tile)
    const int i = wave * 8 + (lane >> 3), cpos = lane & 7;
    const int ck = cpos ^ ((i >> 1) & 7), cv = cpos ^ (((i >> 1) & 1) * 4);
    {
        const int jrow = min(i, 2 * WIN - 2);                            // table rows 27..31 do not exist: their products are never used
        dma16(p.table_w + (size_t)jrow * HD + ck * 8, tbl_lds + wave * 1024);
        dma16(p.table_h + (size_t)jrow * HD + ck * 8, tbl_lds + 4096 + wave * 1024);
    }
    f16x8 qpre[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int jt = j == 0 ? jt0 : wave + 4;
        const int qi = min(jt * 32 + (lane & 31), nreal - 1);
        const size_t tokq = ((size_t)b * S + wy * WIN + qi / nrx) * S + wx * WIN + qi % nrx;
        const f16* q = p.qkv + tokq * p.ld + head * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qpre[j][ks] = *reinterpret_cast<const f16x8*>(q + (ks * 2 + half) * 8);
    }
    {
        const int ii = i < 28 ? i : 0;
        const int rr = ii >= 14, cc = ii - 14 * rr;
        const int x = wx * WIN + cc;
        const f16* padk = p.bias_qkv + D + head * HD + ck * 8;
        const f16* padv = p.bias_qkv + 2 * D + head * HD + cv * 8;
        const f16* row0 = p.qkv + (((size_t)b * S + wy * WIN + rr) * S + x) * p.ld + head * HD;
        const size_t tstride = (size_t)2 * S * p.ld;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const bool real = (wy * WIN + 2 * t + rr) < S && x < S;
            const f16* row = row0 + t * tstride;
            // nontemporal: every K / V row is read by exactly one workgroup (+0.3 % tiles/s, profiles/r04_layernorm_nt.txt)
            dma16<2>(real ? row + D + ck * 8 : padk, k_lds + t * 4096 + wave * 1024);
            dma16<2>(real ? row + 2 * D + cv * 8 : padv, v_lds + t * 4096 + wave * 1024);
        }
    }
    // (Running the first query tile's rel-pos and the first key tiles under the rest of the K / V DMA was tried with one static LDS
    // array per DMA group and __builtin_amdgcn_s_waitcnt — the recipe of attn_global_kernel below: the compiler still fenced the
    // rel_h ds_writes and the key loop's header with vmcnt(0), and three barriers instead of one made it no faster, 0.407-0.412 vs
    // 0.403-0.408 ms per step.)
    WK_PHASE(0)                                           // 0: decode, DMA issue, query loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA writes have landed; all waves' after the barrier
    WK_PHASE(1)                                           // 1: waiting for this wave's DMA / loads
    __syncthreads();
    WK_PHASE(2)                                           // 2: barrier

    const float c_exp = p.scale * 1.4426950408889634f;
    const float inv_scale = 1.0f / p.scale;
    float* rh = rh_lds + wave * 32 * 17;
    const float* rhq = rh + (lane & 31) * 17 + 1;
    int vb0, vb1;
    vtr_bases(lane, vb0, vb1);
    QState st;
    size_t tok = 0;
    bool valid = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int jt = j == 0 ? jt0 : wave + 4;
        const bool has = jt < ntq && !(j == 1 && split);
        if (has) {
            const int qi_raw = jt * 32 + (lane & 31);
            valid = qi_raw < nreal;
            const int qi = valid ? qi_raw : nreal - 1;
            const int ry = qi / nrx, rx = qi % nrx;
            tok = ((size_t)b * S + wy * WIN + ry) * S + wx * WIN + rx;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st.q[ks] = qpre[j][ks];
            st.m = -INFINITY;
            st.l = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st.o[dt][r] = 0.f;
            window_relpos(st, tbl_lds, rx, ry, rh, inv_scale, lane);
        }
        WK_PHASE(3)                                       // 3: query setup + rel-pos
        if (!has) continue;
        window_keys(st, k_lds, v_lds, rhq, t0, t1, vb0, vb1, c_exp, lane);
        WK_PHASE(4)                                       // 4: key loops
        if (!split) {
            store_query(st, p, tok, head, lane, valid);
            __builtin_amdgcn_wave_barrier();
        }
        WK_PHASE(5)                                       // 5: stores
    }
    if (split) {
        // merge the key-split partials: waves with part > 0 park (max, partial sum, O^T) in the K region (all key loops are done)
        __syncthreads();
        float* mb = reinterpret_cast<float*>(k_lds);
        if (part > 0) {
            float* w = mb + (size_t)(wave - ntq) * 34 * 64 + lane;
            w[0] = st.m; w[64] = st.l;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) w[(2 + dt * 16 + r) * 64] = st.o[dt][r];
        }
        __syncthreads();
        if (part == 0) {
            for (int pp = 1; pp < nsplit; ++pp) {
                const float* rd = mb + (size_t)(wave + pp * ntq - ntq) * 34 * 64 + lane;
                const float m2 = rd[0], l2 = rd[64];
                const float m_new = fmaxf(st.m, m2);
                const float a1 = __builtin_amdgcn_exp2f((st.m - m_new) * c_exp), a2 = __builtin_amdgcn_exp2f((m2 - m_new) * c_exp);
                st.l = st.l * a1 + l2 * a2;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st.o[dt][r] = st.o[dt][r] * a1 + rd[(2 + dt * 16 + r) * 64] * a2;
                st.m = m_new;
            }
            store_query(st, p, tok, head, lane, valid);
        }
    }
    WK_PHASE(6)                                           // 6: merge of the key-split partials + their stores
    WK_PHASE_END
}


// ---------------------------------------------------------------------------------------------
// Global attention: one workgroup per (image, head, 128-query block); K rows and row-major V rows streamed through a
// double-buffered LDS ring, 2 key tiles (64 MFMA rows) per stage, filled by LDS-DMA one stage ahead (buffer_load ... lds with the
// stage's scalar offset: no address arithmetic, no staging registers and no ds_write in the loop; wave w moves rows 8w .. 8w+7 of
// each of the stage's four 32-row tiles K a, K b, V a, V b, swizzles on the source side).  XCD-aware order keeps a head's K / V in
// one XCD's L2.  158 VGPRs and 49.7 KB of LDS: three workgroups per CU.
// What makes the DMA ring expressible in HIP (found in round 4): the compiler tracks in-flight LDS-DMA per LDS OBJECT and understands
// __builtin_amdgcn_s_waitcnt — with every ring stage its own static __shared__ array and the builtin (not inline asm) for the wait,
// it adds no vmcnt(0) of its own in front of the LDS reads of the other stage.  The rules it imposes: DMA issued in straight-line
// code (not under a condition), reads of an object that no DMA writes (the rel_h table) only while nothing is in flight — they are
// taken before the stage's DMA is issued — and one dynamic LDS block or an inline-asm wait fences every LDS read behind all DMA.
// Against the register-staged ring of round 3 (global -> VGPR -> ds_write): bit-identical, 85.9 -> 79.2 us alone, 0.405 -> 0.366
// ms per step in the model (profiles/r04_attention_dma_tr.txt).
// ---------------------------------------------------------------------------------------------
template <int WIN, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_global_kernel(AttnParams p) {
    constexpr int NT = Geom<WIN>::NT, WP = Geom<WIN>::WP, RPT = Geom<WIN>::RPT;
    constexpr int STAGE = 2 * 4096 + 2 * 4096;   // 2 K tiles + 2 V tiles
    __shared__ __attribute__((aligned(16))) char ring0[STAGE];
    __shared__ __attribute__((aligned(16))) char ring1[STAGE];
    __shared__ __attribute__((aligned(16))) float rh_lds[4 * 32 * (WP + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, D = p.heads * HD;
    const int nqb = (S * S) / 128;
    int qb, bh;
    {
        const int nbh = p.B * p.heads;
        const int u = blockIdx.x, xcd = u & 7, j = u >> 3;
        if ((nbh & 7) == 0) { bh = (j / nqb) * 8 + xcd; qb = j % nqb; }
        else { bh = u / nqb; qb = u % nqb; }
    }
    const int head = bh % p.heads, b = bh / p.heads;
    const size_t tok0 = (size_t)b * S * S;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qkv + tok0 * p.ld), 0, 0x7fffffff, 0x00020000);
    const int ldb = p.ld * 2;
    const int r = wave * 8 + (lane >> 3), cpos = lane & 7;
    const int ko = r * ldb + (D + head * HD) * 2 + ((cpos ^ ((r >> 1) & 7)) * 16);               // read_kfrag's swizzle, source side
    const int vo = r * ldb + (2 * D + head * HD) * 2 + ((cpos ^ (((r >> 1) & 1) * 4)) * 16);     // the ds_read_b64_tr_b16 swizzle (vtr_bases)
#define SRH_DMA_STAGE(sidx, ring) { const int so_ = (sidx) * 64 * ldb; char* d_ = (ring) + wave * 1024; \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (lds_ptr)(d_), 16, ko, so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (lds_ptr)(d_ + 4096), 16, ko, so_ + 32 * ldb, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (lds_ptr)(d_ + 8192), 16, vo, so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (lds_ptr)(d_ + 12288), 16, vo, so_ + 32 * ldb, 0, 0); }
    // the two rel-pos tables in the K-tile format (NTJ tiles of 4 KiB each): wave w moves rows 8 w .. 8 w + 7 of a tile.  Up to the 32 x 32
    // window both fit ring1 (K / V stage 1 overwrites it behind stage 0's barrier, which every wave passes after its rel-pos) and stage 0
    // of the K / V stream flies under the prologue; the 64 x 64 window's tables (127 rows each) fill BOTH ring stages — w in ring1, h in
    // ring0 — and the K / V stream starts behind the rel-pos (one exposed stage of 64)
    constexpr int NTJ = (2 * WIN - 1 + 31) / 32;
    constexpr bool TBL2 = 2 * NTJ * 4096 > STAGE;
    static_assert(NTJ * 4096 <= STAGE, "one table per ring stage at most");
    {
        typedef const __attribute__((address_space(1))) void* glb_ptr;
#pragma unroll
        for (int tt = 0; tt < 2 * NTJ; ++tt) {
            const int j = min((tt % NTJ) * 32 + r, 2 * WIN - 2);
            const f16* src = (tt < NTJ ? p.table_w : p.table_h) + (size_t)j * HD + (cpos ^ ((r >> 1) & 7)) * 8;
            char* dst = TBL2 ? (tt < NTJ ? ring1 : ring0) + (tt % NTJ) * 4096 : ring1 + tt * 4096;
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(dst + wave * 1024), 16, 0, 0);
        }
    }
    if constexpr (!TBL2) SRH_DMA_STAGE(0, ring0)                   // in flight under the query loads and the rel-pos prologue

    const int qi = qb * 128 + wave * 32 + (lane & 31);
    const size_t tok = tok0 + qi;
    QState st;
    load_query<WIN>(st, p, tok, head, lane);
    float* rh = rh_lds + wave * 32 * (WP + 1);
    if constexpr (TBL2) __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): nothing but the table pieces is in flight
    else __builtin_amdgcn_s_waitcnt(0x0F74);                       // vmcnt(4): the table pieces have landed (stage 0's four may stay in flight) ...
    __builtin_amdgcn_s_barrier();                                  // ... every wave's
    asm volatile("" ::: "memory");
    fused_relpos<WIN, WP + 1>(st, p, qi / S, qi % S, rh, ring1, TBL2 ? ring0 : ring1 + NTJ * 4096, lane);
    if constexpr (TBL2) {
        __builtin_amdgcn_s_barrier();                              // every wave is done with the h table in ring0
        asm volatile("" ::: "memory");
        SRH_DMA_STAGE(0, ring0)
    }
    int vb0, vb1;
    vtr_bases(lane, vb0, vb1);

    const float c_exp = p.scale * 1.4426950408889634f;
    constexpr int NSTAGE = NT / 2;
    static_assert(NSTAGE % 2 == 0, "the key loop is unrolled by two stages");
    // rel_h entries (one per window row) between the two tiles of a stage / per stage: the 64 x 64 window's stage is ONE row
    constexpr int RSTEP = WIN == 64 ? 0 : RPT, RSTAGE = WIN == 64 ? 1 : 2 * RPT;
    const float* rhp = rh + (lane & 31) * (WP + 1);
    // one hand-over per stage: this wave's four pieces have landed (vmcnt(0): nothing else is in flight), the barrier makes that
    // true for every wave's pieces and says that every wave is done reading the OTHER ring stage — which the next DMA overwrites
#define SRH_STAGE(sidx, ring, other, rbuf) { \
        __builtin_amdgcn_s_waitcnt(0x0F70);                     /* vmcnt(0), lgkmcnt / expcnt untouched */ \
        __builtin_amdgcn_s_barrier(); \
        asm volatile("" ::: "memory"); \
        /* rel_h of the stage's two tiles first: rh_lds is not a DMA target, so its reads must come while no DMA is in flight */ \
        const float rh0a = rhp[(rbuf) * RSTAGE], rh1a = RPT == 2 ? rhp[(rbuf) * RSTAGE + 1] : 0.f; \
        const float rh0b = rhp[(rbuf) * RSTAGE + RSTEP], rh1b = RPT == 2 ? rhp[(rbuf) * RSTAGE + RSTEP + 1] : 0.f; \
        asm volatile("" :: "v"(rh0a), "v"(rh1a), "v"(rh0b), "v"(rh1b) : "memory"); \
        const int snext_ = (sidx) + 1 < NSTAGE ? (sidx) + 1 : (sidx);   /* last stage re-loads itself (no branch) */ \
        SRH_DMA_STAGE(snext_, other) \
        f16x8 kfA[4], kfB[4]; \
        read_kfrag(kfA, ring, lane); \
        read_kfrag(kfB, ring + 4096, lane); \
        attn_tile2<WIN>(st, kfA, kfB, ring + 8192, ring + 8192 + 4096, vb0, vb1, rh0a, rh1a, rh0b, rh1b, c_exp, lane); }
    for (int sidx = 0; sidx < NSTAGE; sidx += 2) {
        SRH_STAGE(sidx, ring0, ring1, 0)
        SRH_STAGE(sidx + 1, ring1, ring0, 1)
        rhp += 2 * RSTAGE;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // the tail's re-load of the last stage must not outlive the workgroup's LDS
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

#ifdef SRH_TUNING      // probe builds only (-Itools/probes): the persistent windowed experiment and the probe dispatcher
#include "attn_tuning.inc"
#endif

int launch_attention(const AttnParams& p_in, hipStream_t s) {
    AttnParams p = p_in;
#ifdef SRH_TUNING      // probe builds only: kernel selection of tools/probes/attn_win_probe (attn_tuning.inc)
    { int rc; if (attn_tuning_dispatch(p, s, &rc)) return rc; }
#else
    p.ablate = 0;
#endif
    const bool mfma_path = p.hd == HD && (p.win == 14 || (p.win == p.S && (p.S == 16 || p.S == 32 || p.S == 64)));
    if (!mfma_path && attention_hdx_supported(p)) return launch_attention_hdx(p, s);
    if (!mfma_path) {      // other head dims / windows (ViT-H at 512 / 1024 px)
        if ((p.hd != 64 && p.hd != 80) || !p.table_h || !p.table_w || p.win > 64) return -2;
        const int nw = (p.S + p.win - 1) / p.win, nqb = (p.win * p.win + 255) / 256;
        const int lds = (p.win > 32 ? GEN_KC / 2 : GEN_KC) * p.hd * 6 + 256 * 2 * p.win * 4;
        if (lds > 160 * 1024) return -2;
        static OncePerDevice generic_opt_in;
        if (!generic_opt_in.run([] {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_generic_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(attn_generic_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            }))
            return -3;
        const dim3 grid(p.B * nw * nw * p.heads * nqb);
        if (p.hd == 64) hipLaunchKernelGGL(attn_generic_kernel<8>, grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL(attn_generic_kernel<10>, grid, dim3(256), lds, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    if (p.win == p.S) {
        const int grid = p.B * p.heads * (p.S * p.S / 128);
        if (p.S == 32) hipLaunchKernelGGL((attn_global_kernel<32, 3>), dim3(grid), dim3(256), 0, s, p);
        else if (p.S == 16) hipLaunchKernelGGL((attn_global_kernel<16, 2>), dim3(grid), dim3(256), 0, s, p);
        else if (p.S == 64) hipLaunchKernelGGL((attn_global_kernel<64, 2>), dim3(grid), dim3(256), 0, s, p);   // 65 KiB of static LDS: two workgroups per CU
        else return -2;
    } else if (p.win == 14) {
        static OncePerDevice window_opt_in;
        if (!window_opt_in.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_window_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS) == hipSuccess; }))
            return -3;
        const int nw = (p.S + 13) / 14;
        const int n_items = p.B * nw * nw * p.heads;
        hipLaunchKernelGGL(attn_window_kernel, dim3(n_items), dim3(256), WIN_LDS, s, p);
    } else {
        return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

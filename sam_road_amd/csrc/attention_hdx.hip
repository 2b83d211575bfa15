// MFMA attention for head dims other than 64 on small windows (ViT-H: head dim 80; 14x14 windows and the 16x16 global
// window of 256-pixel tiles, i.e. BASELINE configs[4] toponet_vith_256.yaml).  Same semantics and the same transposed
// one-lane-per-query formulation as attention.hip (S^T = K Q^T, O^T = V^T P^T with v_mfma_f32_32x32x16_f16, rel_w as the
// C operand of the first S MFMA, rel_h folded into the row max / exp addend, pad tokens are real keys with k = b_k,
// v = b_v, pad queries are skipped), generalised over the number of 16-wide k steps (KS = HD / 16 = 5) and of 32-row
// O^T tiles (NDT = ceil(HD / 32) = 3, rows HD..95 are zero rows of V^T).  All keys of the window (<= 256) are staged in
// LDS once per (image, head, window): K rows at their natural 160-byte stride with the 16-byte chunk index XORed with bit 3
// of the row (rows r and r + 8 start in the same banks; the XOR moves one of them by four banks, which makes the ds_read_b128
// fragment reads of every 16-lane service group conflict-free), V^T tiles in attention.hip's key-permuted layout (4-key x
// 8-dim blocks transposed with v_perm_b32) holding only the HD real rows.  Round 3 trimmed the LDS image from 91 to 77.5 KiB
// (no K pad, no zero rows, a 15-float rel-pos row) so that TWO workgroups fit a CU: at B = 8 the launch is 512 workgroups,
// i.e. two rounds of one workgroup per CU with nothing to overlap a workgroup's staging phase became one round of two.
// This replaces the f32 VALU fallback (attn_generic_kernel), which took 7.3 of the 14.4 ms of a ViT-H step.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace srh {
namespace {

template <int WIN> struct GeomX;
template <> struct GeomX<14> { static constexpr int KPT = 28, RPT = 2, NT = 7; };
template <> struct GeomX<16> { static constexpr int KPT = 32, RPT = 2, NT = 8; };

// local MFMA row i of a key tile -> (row-in-tile, col) of the window
template <int WIN> __device__ __forceinline__ void hx_tile_rc(int i, int& r, int& c) {
    if (WIN == 16) { r = i >> 4; c = i & 15; }
    else { r = i >= 14; c = i - 14 * r; }
}
// V^T slot of local key i: the key permutation that makes exp(S^T) registers the P^T B-fragment (attention.hip vt_slot)
__device__ __forceinline__ int hx_vt_slot(int i) {
    const int half = (i >> 2) & 1, reg = (i & 3) + 4 * (i >> 3);
    return ((reg >> 3) * 2 + half) * 8 + (reg & 7);
}

template <int KS, int NDT>
struct QStateX {
    f16x8 q[KS];
    f32x16 relw;
    float m, l;
    f32x16 o[NDT];
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// the pieces of a key tile's softmax (attention.hip tile_max / tile_exp / update_max, for WIN 14 / 16 and NDT output tiles)
template <int WIN>
__device__ __forceinline__ float hx_tile_max(const f32x16& s, float rh0, float rh1, int half) {
    if (WIN == 16) {
        float ma = s[0], mb = s[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) { ma = fmaxf(ma, s[r]); mb = fmaxf(mb, s[8 + r]); }
        return fmaxf(ma + rh0, mb + rh1);
    } else {
        const float rhm = half ? rh1 : rh0;
        float ma = s[0], mb = s[8];
#pragma unroll
        for (int r = 1; r < 6; ++r) ma = fmaxf(ma, s[r]);
#pragma unroll
        for (int r = 9; r < 12; ++r) mb = fmaxf(mb, s[r]);
        const float mt = fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]));
        mb = fmaxf(mb, half ? -INFINITY : mt);                   // rows 28..31 are not keys
        return fmaxf(fmaxf(ma + rh0, mb + rh1), fmaxf(s[6], s[7]) + rhm);
    }
}

template <int WIN>
__device__ __forceinline__ float hx_tile_exp(const f32x16& s, float rh0, float rh1, float m_new, float c_exp, int half, f16x8 (&pb)[2]) {
    const float mc = -m_new * c_exp;
    const float mc0 = fmaf(rh0, c_exp, mc), mc1 = fmaf(rh1, c_exp, mc), mcm = WIN == 14 ? (half ? mc1 : mc0) : mc0;
    const float mct = WIN == 14 ? (half ? -INFINITY : mc1) : mc1;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float ad = WIN == 16 ? (r >= 8 ? mc1 : mc0) : (r < 6 ? mc0 : r < 8 ? mcm : r < 12 ? mc1 : mct);
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c_exp, ad));
        sum += pv;
        pb[r >> 3][r & 7] = (f16)pv;
    }
    return sum;
}

template <int KS, int NDT>
__device__ __forceinline__ float hx_update_max(QStateX<KS, NDT>& st, float mloc, float c_exp) {
    {   // max over both 32-lane halves: v_permlane32_swap_b32 (VALU) instead of a ds_bpermute round trip
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const float m_new = fmaxf(st.m, mloc);
    if (__any(m_new != st.m)) {
        const float alpha = __builtin_amdgcn_exp2f((st.m - m_new) * c_exp);
        st.l *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[dt][r] *= alpha;
        st.m = m_new;
    }
    return m_new;
}

template <int NDT>
__device__ __forceinline__ void hx_read_vfrag(f16x8 (&vf)[NDT][2], const char* vt_lds, int lane) {
    const int half = lane >> 5, row = lane & 31;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int c = (sx * 2 + half) ^ ((row >> 2) & 3);
            vf[dt][sx] = *reinterpret_cast<const f16x8*>(vt_lds + dt * 2048 + row * 64 + c * 16);
        }
}

// one 32-row key tile (see attention.hip attn_tile; identical arithmetic with KS k steps and NDT output tiles)
template <int WIN, int KS, int NDT>
__device__ __forceinline__ void hx_tile(QStateX<KS, NDT>& st, const f16x8 (&kf)[KS], const char* vt_lds, float rh0, float rh1,
                                        float c_exp, int lane) {
    const int half = lane >> 5;
    f32x16 s = mfma32(kf[0], st.q[0], st.relw);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) s = mfma32(kf[ks], st.q[ks], s);
    f16x8 vf[NDT][2];
    hx_read_vfrag<NDT>(vf, vt_lds, lane);
    __builtin_amdgcn_sched_barrier(0);
    const float m_new = hx_update_max<KS, NDT>(st, hx_tile_max<WIN>(s, rh0, rh1, half), c_exp);
    f16x8 pb[2];
    st.l += hx_tile_exp<WIN>(s, rh0, rh1, m_new, c_exp, half, pb);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) st.o[dt] = mfma32(vf[dt][sx], pb[sx], st.o[dt]);
}

// two key tiles at once (attention.hip attn_tile2): the two S^T chains — KS = 5 dependent MFMAs each at head dim 80 — interleave,
// one cross-half exchange / rescale decision per pair, the P.V MFMAs rotate over the NDT accumulators
template <int WIN, int KS, int NDT>
__device__ __forceinline__ void hx_tile2(QStateX<KS, NDT>& st, const f16x8 (&kfa)[KS], const f16x8 (&kfb)[KS], const char* vta,
                                         const char* vtb, float rh0a, float rh1a, float rh0b, float rh1b, float c_exp, int lane) {
    const int half = lane >> 5;
    f32x16 sa = mfma32(kfa[0], st.q[0], st.relw);
    f32x16 sb = mfma32(kfb[0], st.q[0], st.relw);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) { sa = mfma32(kfa[ks], st.q[ks], sa); sb = mfma32(kfb[ks], st.q[ks], sb); }
    f16x8 vfa[NDT][2], vfb[NDT][2];
    hx_read_vfrag<NDT>(vfa, vta, lane);
    hx_read_vfrag<NDT>(vfb, vtb, lane);
    __builtin_amdgcn_sched_barrier(0);
    const float m_new = hx_update_max<KS, NDT>(st, fmaxf(hx_tile_max<WIN>(sa, rh0a, rh1a, half), hx_tile_max<WIN>(sb, rh0b, rh1b, half)), c_exp);
    f16x8 pa[2], pb[2];
    const float suma = hx_tile_exp<WIN>(sa, rh0a, rh1a, m_new, c_exp, half, pa);
    const float sumb = hx_tile_exp<WIN>(sb, rh0b, rh1b, m_new, c_exp, half, pb);
    st.l += suma + sumb;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) st.o[dt] = mfma32(vfa[dt][sx], pa[sx], st.o[dt]);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) st.o[dt] = mfma32(vfb[dt][sx], pb[sx], st.o[dt]);
}

// grid = (slot, image, head), slot = (window, part of its query tiles) heaviest first (AttnParams::hx_order); 4 waves, ONE 32-query tile of
// the window's REAL tokens per wave: a window with more than four query tiles (the full 14 x 14 and the 16 x 16 global one) is TWO
// workgroups that both stage its keys — at B = 8 the full windows were 128 workgroups on 256 CUs, each wave walking two query tiles
// one after the other, while the edge windows' workgroups (28 and 4 real queries) had long finished.
// The keys go through LDS in TWO PHASES of at most four key tiles (the online softmax does not care): 48.5 KiB instead of 78.5, so THREE
// workgroups fit a CU and the 640 workgroups of a ViT-H launch at B = 8 (5 per image and head: 4 + 3 query tiles of the full window,
// the two edge windows, the corner) are ONE round on 768 slots instead of 1.25 on 512.  Every global load of a phase is in flight
// before its first LDS write, and the wave's q rows and rel-pos table fragments are
// requested first of all and consumed after the staging: the kernel was two chains of dependent loads (6.6 us of staging, 10.2 us of
// query phase; tools/probes/hdx_probe, profiles/r05_attention_hdx.txt).
template <int HD, int WIN>
__global__ __launch_bounds__(256, WIN == 14 ? 3 : 1) void attn_hdx_kernel(AttnParams p) {
    constexpr int KS = HD / 16, NDT = (HD + 31) / 32, NT = GeomX<WIN>::NT, KPT = GeomX<WIN>::KPT, RPT = GeomX<WIN>::RPT;
    constexpr int KROW = HD * 2, NCH = HD / 8, VTT = HD * 64;      // bytes per K row / per V^T key tile (HD rows x 32 keys x 2 B)
    constexpr int NTP = WIN == 14 ? 4 : NT;                         // key tiles per LDS phase (phase 0: tiles 0..NTP-1, phase 1: the rest); the global
                                                                    // window's workgroups are one per CU anyway and keep all their keys resident
    constexpr int LDS_K = NTP * 32 * KROW, LDS_VT = NTP * VTT + (32 * NDT - HD) * 64, RHS = WIN + 1;   // RHS: odd rel-pos row stride
    static_assert(HD % 16 == 0 && (NCH % 2) == 0, "head dim must be a multiple of 16");
    static_assert(NT <= 2 * NTP, "at most two phases");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const k_lds = smem;
    char* const vt_lds = smem + LDS_K;
    float* const rh_lds = reinterpret_cast<float*>(smem + LDS_K + LDS_VT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int S = p.S, D = p.heads * HD;
    const int nw = WIN == 14 ? (S + WIN - 1) / WIN : 1;
    int u = blockIdx.x;
    const int head = u % p.heads; u /= p.heads;
    const int b = u % p.B; u /= p.B;
    const int entry = (p.hx_order[u >> 2] >> ((u & 3) * 8)) & 0xff;
    const int widx = entry >> 1, part = entry & 1;
    const int wy = widx / nw, wx = widx % nw;
    const int nry = min(WIN, S - wy * WIN), nrx = min(WIN, S - wx * WIN);
    const int nreal = nry * nrx;
    const int ntq = (nreal + 31) / 32;

    // K rows and the transposed, key-permuted V^T tiles (pad positions: k = b_k, v = b_v; key rows KPT..31 are zero).
    // (V^T rows HD .. 32 * NDT - 1 do not exist: the last d tile's fragment reads run into the next key tile — or, for a phase's last
    // one, into the 1 KiB pad behind the array — and produce garbage O^T rows d >= HD, which are never stored; every O^T row
    // depends on its own V^T row only)
    auto key_src = [&](int t, int i) -> const f16* {             // qkv row of local key i of tile t (pad position: the bias row)
        int rr, cc;
        hx_tile_rc<WIN>(i, rr, cc);
        const int y = wy * WIN + t * RPT + rr, x = wx * WIN + cc;
        return (y < S && x < S) ? p.qkv + (((size_t)b * S + y) * S + x) * p.ld : p.bias_qkv;
    };
    constexpr int NKI = (NTP * 32 * NCH + 255) / 256, NVI = (NTP * 8 * NCH + 255) / 256;      // per-thread items of a full phase
    auto stage_load = [&](int t0, int nt, uint4 (&kreg)[NKI], uint4 (&vreg)[NVI][4]) {       // global -> registers, key tiles t0 .. t0 + nt - 1
#pragma unroll
        for (int j = 0; j < NKI; ++j) {
            const int item = tid + j * 256;
            const int c = item % NCH, i = (item / NCH) & 31, t = min(item / (NCH * 32), nt - 1);
            kreg[j] = make_uint4(0, 0, 0, 0);
            if (i < KPT && item < nt * 32 * NCH) kreg[j] = *reinterpret_cast<const uint4*>(key_src(t0 + t, i) + D + head * HD + c * 8);
        }
#pragma unroll
        for (int j = 0; j < NVI; ++j) {
            const int item = tid + j * 256;
            const int c = item % NCH, kq = (item / NCH) & 7, t = min(item / (NCH * 8), nt - 1);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int i = kq * 4 + jj;
                vreg[j][jj] = (i < KPT && item < nt * 8 * NCH) ? *reinterpret_cast<const uint4*>(key_src(t0 + t, i) + 2 * D + head * HD + c * 8)
                                                                : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto stage_store = [&](int nt, const uint4 (&kreg)[NKI], const uint4 (&vreg)[NVI][4]) {  // registers -> LDS (tile index local to the phase)
#pragma unroll
        for (int j = 0; j < NKI; ++j) {
            const int item = tid + j * 256;
            const int c = item % NCH, i = (item / NCH) & 31, t = item / (NCH * 32);
            if (item < nt * 32 * NCH) *reinterpret_cast<uint4*>(k_lds + (t * 32 + i) * KROW + (c ^ ((i >> 3) & 1)) * 16) = kreg[j];
        }
        // V: 4 keys x 8 dims per item -> 8 dims x 4 keys: every output word pairs the same fp16 of two keys = one v_perm_b32
#pragma unroll
        for (int j = 0; j < NVI; ++j) {
            const int item = tid + j * 256;
            if (item >= nt * 8 * NCH) continue;
            const int c = item % NCH, kq = (item / NCH) & 7, t = item / (NCH * 8);
            const int slot = hx_vt_slot(kq * 4);
            const int sc = slot >> 3, eo = slot & 7;             // eo is 0 or 4
            const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&vreg[j][0]);
            const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&vreg[j][1]);
            const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&vreg[j][2]);
            const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&vreg[j][3]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = c * 8 + e, dl = d & 31;
                const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                uint2 w;
                w.x = __builtin_amdgcn_perm(w1[e >> 1], w0[e >> 1], sel);
                w.y = __builtin_amdgcn_perm(w3[e >> 1], w2[e >> 1], sel);
                *reinterpret_cast<uint2*>(vt_lds + t * VTT + d * 64 + ((sc ^ ((dl >> 2) & 3)) * 16) + eo * 2) = w;
            }
        }
    };

    // ---- this wave's query tile (one per wave; part 1 of a window holds its query tiles 4..7)
    const int jt = part * 4 + wave;
    const bool has_q = jt < ntq;
    const int qi_raw = jt * 32 + (lane & 31);
    const bool valid = has_q && qi_raw < nreal;
    const int qi = min(qi_raw, nreal - 1);
    const int ry = qi / nrx, rx = qi % nrx;
    const size_t tok = ((size_t)b * S + wy * WIN + ry) * S + wx * WIN + rx;
    QStateX<KS, NDT> st;
    f16x8 tab[2][KS];
    const int jrow_t = min(lane & 31, 2 * WIN - 2);
    constexpr bool do_stage = true;
    {
        const f16* q = p.qkv + tok * p.ld + head * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st.q[ks] = *reinterpret_cast<const f16x8*>(q + (ks * 2 + half) * 8);
            tab[0][ks] = *reinterpret_cast<const f16x8*>(p.table_w + (size_t)jrow_t * HD + (ks * 2 + half) * 8);
            tab[1][ks] = *reinterpret_cast<const f16x8*>(p.table_h + (size_t)jrow_t * HD + (ks * 2 + half) * 8);
        }
    }
    uint4 kreg[NKI], vreg[NVI][4];
    if (do_stage) {
        stage_load(0, NTP, kreg, vreg);
        stage_store(NTP, kreg, vreg);
    }

    const float c_exp = p.scale * 1.4426950408889634f;
    const float inv_scale = 1.0f / p.scale;
    float* const rh = rh_lds + wave * 32 * RHS;
    if (has_q) {
        st.m = -INFINITY;
        st.l = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.o[dt][r] = 0.f;
        // fused decomposed rel-pos bias (attention.hip fused_relpos): table rows x Q via MFMA, scattered to the wave's
        // LDS table; the w part becomes the 16 tile-invariant per-lane values, the h part stays in LDS (2 per tile).
        // The wave's table is its own LDS region: this runs before the workgroup's staging barrier.
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int qc = pass == 0 ? rx : ry;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = mfma32(tab[pass][ks], st.q[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jrow = mfma32_row(r, lane);
                const int k = qc - jrow + WIN - 1;
                if (k >= 0 && k < WIN && jrow < 2 * WIN - 1) rh[(lane & 31) * RHS + k] = acc[r] * inv_scale;
            }
            __builtin_amdgcn_wave_barrier();
            if (pass == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int rr, cc;
                    hx_tile_rc<WIN>(mfma32_row(r, lane), rr, cc);
                    st.relw[r] = (cc < WIN) ? rh[(lane & 31) * RHS + cc] : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();                                             // phase 0's keys are in LDS
    auto read_kf = [&](f16x8 (&kf)[KS], int t) {                 // t: tile index inside the phase
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            kf[ks] = *reinterpret_cast<const f16x8*>(k_lds + (t * 32 + (lane & 31)) * KROW + (((ks * 2 + half) ^ ((lane >> 3) & 1)) * 16));
    };
    const float* rhq = rh + (lane & 31) * RHS;
    auto key_tiles = [&](int t0, int nt) {                        // key tiles t0 .. t0 + nt - 1 of the window = tiles 0 .. nt - 1 of the phase
        f16x8 kfa[KS];
        if constexpr (WIN == 16) {                               // one workgroup per CU: registers for two key tiles at once
            f16x8 kfb[KS];
#pragma unroll 1
            for (int t = 0; t + 1 < nt; t += 2) {
                read_kf(kfa, t);
                read_kf(kfb, t + 1);
                hx_tile2<WIN, KS, NDT>(st, kfa, kfb, vt_lds + t * VTT, vt_lds + (t + 1) * VTT, rhq[(t0 + t) * RPT], rhq[(t0 + t) * RPT + 1],
                                       rhq[(t0 + t + 1) * RPT], rhq[(t0 + t + 1) * RPT + 1], c_exp, lane);
            }
            if (nt & 1) {
                read_kf(kfa, nt - 1);
                hx_tile<WIN, KS, NDT>(st, kfa, vt_lds + (nt - 1) * VTT, rhq[(t0 + nt - 1) * RPT], rhq[(t0 + nt - 1) * RPT + 1], c_exp, lane);
            }
        } else {                                                 // three workgroups per CU (168 registers): one key tile at a time, the other
#pragma unroll 1                                                 // two resident workgroups' waves fill the gaps instead of a second S chain
            for (int t = 0; t < nt; ++t) {
                read_kf(kfa, t);
                hx_tile<WIN, KS, NDT>(st, kfa, vt_lds + t * VTT, rhq[(t0 + t) * RPT], rhq[(t0 + t) * RPT + 1], c_exp, lane);
            }
        }
    };
    if (has_q) key_tiles(0, NTP);
    if constexpr (NT > NTP) {
        if (do_stage) stage_load(NTP, NT - NTP, kreg, vreg);     // (168 registers at three workgroups per CU do not hold these across the key loop)
        __syncthreads();                                         // everyone is done with phase 0's keys
        if (do_stage) stage_store(NT - NTP, kreg, vreg);
        __syncthreads();
        if (has_q) key_tiles(NTP, NT - NTP);
    }
    if (!has_q) return;
    const auto lr = __builtin_amdgcn_permlane32_swap(__float_as_uint(st.l), __float_as_uint(st.l), false, false);
    const float inv = 1.0f / (__uint_as_float(lr[0]) + __uint_as_float(lr[1]));
    if (valid) {
        f16* o = p.out + tok * p.ldo + head * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d0 = dt * 32 + 8 * qd + 4 * half;
                if (d0 < HD) {
                    f16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (f16)(st.o[dt][qd * 4 + e] * inv);
                    *reinterpret_cast<f16x4*>(o + d0) = h;
                }
            }
    }
}

template <int HD, int WIN>
int hdx_launch(const AttnParams& p, hipStream_t s) {
    constexpr int NDT = (HD + 31) / 32, NT = GeomX<WIN>::NT;
    constexpr int NTP = WIN == 14 ? 4 : NT;                       // key tiles per LDS phase (attn_hdx_kernel)
    constexpr int lds = NTP * 32 * HD * 2 + NTP * HD * 64 + (32 * NDT - HD) * 64 + 4 * 32 * (WIN + 1) * 4;   // WIN 14: 48.5 KiB = three per CU
    static OncePerDevice opt_in;                                  // per template instantiation; the attribute is per device
    if (!opt_in.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_hdx_kernel<HD, WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess; }))
        return -3;
    const int nw = WIN == 14 ? (p.S + WIN - 1) / WIN : 1;
    // workgroup slots of one (image, head): every window's query tiles in parts of four (one per wave), heaviest part first
    int weight[64], entry[64], nslots = 0;
    for (int widx = 0; widx < nw * nw; ++widx) {
        const int nry = std::min(WIN, p.S - (widx / nw) * WIN), nrx = std::min(WIN, p.S - (widx % nw) * WIN);
        const int ntq = (nry * nrx + 31) / 32;
        for (int part = 0; part * 4 < ntq; ++part) {
            if (part > 1 || nslots >= 64) return -2;
            weight[nslots] = std::min(4, ntq - part * 4); entry[nslots] = widx << 1 | part; ++nslots;
        }
    }
    AttnParams q = p;
    for (int i = 0; i < 16; ++i) q.hx_order[i] = 0;
    for (int i = 0; i < nslots; ++i) {          // selection sort (stable: equal weights keep window order), nslots <= 64
        int best = i;
        for (int j = i + 1; j < nslots; ++j) if (weight[j] > weight[best]) best = j;
        const int w = weight[best], e = entry[best];
        for (int j = best; j > i; --j) { weight[j] = weight[j - 1]; entry[j] = entry[j - 1]; }
        weight[i] = w; entry[i] = e;
        q.hx_order[i >> 2] |= (unsigned)e << ((i & 3) * 8);
    }
    hipLaunchKernelGGL((attn_hdx_kernel<HD, WIN>), dim3(nslots * p.B * p.heads), dim3(256), lds, s, q);
    return SRH_CHECK_LAUNCH();
}
}  // namespace

bool attention_hdx_supported(const AttnParams& p) {
    return p.hd == 80 && p.table_h && p.table_w && ((p.win == 14 && p.S <= 5 * 14) || (p.win == 16 && p.S == 16));      // <= 64 workgroup slots per (image, head)
}

int launch_attention_hdx(const AttnParams& p, hipStream_t s) {
    if (!attention_hdx_supported(p)) return -2;
    return p.win == 14 ? hdx_launch<80, 14>(p, s) : hdx_launch<80, 16>(p, s);
}

}  // namespace srh

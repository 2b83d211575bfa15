// Attention probe: the windowed (LDS-DMA / ds_read_b64_tr_b16 / key split) and global kernels of attention.hip timed alone at the
// bench shape, with the kernel selection of a SRH_TUNING build (tools/probes/attn_tuning.inc: 9 the per-window kernel's phase counters,
// 13 global kernel at two workgroups / CU, 20 the asm global-attention experiment, 21..28 its ablations, 40..49 the persistent windowed
// experiment and ITS ablations).  The in-kernel ablation switches of the product kernels (no key loop, no staging, no rel-pos, round-4
// order: rounds 3-5) were removed in round 6; their measurements are profiles/r04_attention_*.txt and profiles/r05_attention_window.txt.
// Build: tools/probes/build_probes.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "../../sam_road_amd/csrc/common.hpp"
#include "../../sam_road_amd/csrc/kernels.hpp"
using namespace srh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float run(AttnParams p, int abl, int reps, hipStream_t st) {
    p.ablate = abl;
    for (int i = 0; i < 3; ++i) launch_attention(p, st);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_attention(p, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / reps);
    }
    return best;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, S = argc > 2 ? atoi(argv[2]) : 32, heads = 12, D = heads * 64;
    const size_t T = (size_t)B * S * S;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<f16> qkv(T * 3 * D), bias(3 * D), th(63 * 64), tw(63 * 64);
    for (auto& v : qkv) v = (f16)(nd(rng) * 1.5f);
    for (auto& v : bias) v = (f16)(nd(rng) * 0.5f);
    for (auto& v : th) v = (f16)(nd(rng) * 0.3f);
    for (auto& v : tw) v = (f16)(nd(rng) * 0.3f);
    f16 *dq, *db, *dh, *dw, *o0, *o1;
    CK(hipMalloc(&dq, qkv.size() * 2)); CK(hipMalloc(&db, bias.size() * 2)); CK(hipMalloc(&dh, th.size() * 2)); CK(hipMalloc(&dw, tw.size() * 2));
    CK(hipMalloc(&o0, T * D * 2)); CK(hipMalloc(&o1, T * D * 2));
    CK(hipMemcpy(dq, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), bias.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, th.data(), th.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, tw.data(), tw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(o0, 0, T * D * 2)); CK(hipMemset(o1, 0xff, T * D * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    AttnParams p;
    p.qkv = dq; p.ld = 3 * D; p.table_h = dh; p.table_w = dw; p.bias_qkv = db; p.ldo = D; p.B = B; p.S = S; p.heads = heads; p.hd = 64; p.win = 14;
    p.scale = 0.125f;
    void* scratch; CK(hipMalloc(&scratch, ATTN_SCRATCH_BYTES + 256 * 8 * 8 * 8)); CK(hipMemset(scratch, 0, ATTN_SCRATCH_BYTES + 256 * 8 * 8 * 8));
    p.scratch = scratch;
    // outputs: the shipped one-workgroup-per-window kernel (ablate 0) against the persistent experiment (ablate 40)
    p.out = o0; p.ablate = 0; if (launch_attention(p, st)) { printf("launch failed\n"); return 1; }
    p.out = o1; p.ablate = 40; if (launch_attention(p, st)) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    std::vector<f16> h0(T * D), h1(T * D);
    CK(hipMemcpy(h0.data(), o0, T * D * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, T * D * 2, hipMemcpyDeviceToHost));
    double md[3] = {0, 0, 0}; size_t nd_[3] = {0, 0, 0}, nn[3] = {0, 0, 0}; double ref_max = 0; size_t nanc = 0;
    for (size_t t = 0; t < T; ++t) {
        const int y = (t / S) % S, x = t % S;
        const int wy = y / 14, wx = x / 14;
        const int nry = std::min(14, S - wy * 14), nrx = std::min(14, S - wx * 14);
        const int cls = (nry * nrx > 128) ? 0 : (nry * nrx > 64 ? 1 : 2);
        for (int d = 0; d < D; ++d) {
            const float a = (float)h0[t * D + d], b = (float)h1[t * D + d];
            if (std::isnan(b)) ++nanc;
            md[cls] = std::max(md[cls], (double)fabsf(a - b)); nd_[cls] += (a != b); ++nn[cls];
            ref_max = std::max(ref_max, (double)fabsf(a));
        }
    }
    printf("B=%d S=%d  max|out|=%.3f  NaN=%zu\n", B, S, ref_max, nanc);
    const char* names[3] = {"5-7 query tiles (same split)", "3-4 query tiles (split here)", "1-2 query tiles (split in both)"};
    for (int c = 0; c < 3; ++c) printf("  %-32s max |persistent - per-window| = %.3e   values differing %zu of %zu\n", names[c], md[c], nd_[c], nn[c]);
    p.out = o1;
    for (int rep = 0; rep < 3; ++rep)
        printf("  windowed:  per-window kernel %6.1f us   persistent %6.1f   persistent: no key loop %6.1f   no rel-pos %6.1f\n",
               run(p, 0, 10, st), run(p, 40, 10, st), run(p, 41, 10, st), run(p, 43, 10, st));
    {   // phase ticks of the persistent experiment (ablate 49): per wave of every workgroup, averaged over the workgroups
        p.ablate = 49; launch_attention(p, st); CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hd(256 * 8 * 8);
        CK(hipMemcpy(hd.data(), (char*)scratch + ATTN_SCRATCH_BYTES, hd.size() * 8, hipMemcpyDeviceToHost));
        const char* ph[8] = {"tail(stores,ctl)", "own DMA wait", "barrier wait", "setup+relpos", "decode+issue+qload", "key loops", "merge", "exit"};
        for (int w = 0; w < 8; w += 7) {
            printf("  persistent kernel, wave %d, ticks per workgroup (avg over workgroups):", w);
            double tot = 0;
            for (int k = 0; k < 8; ++k) { double a = 0; int nz = 0; for (int b = 0; b < 256; ++b) { a += (double)hd[(b * 8 + w) * 8 + k]; nz += hd[(b * 8 + w) * 8 + k] != 0; } a /= 256; tot += a; printf("  %s %.0f", ph[k], a); }
            printf("  | total %.0f\n", tot);
        }
        // the per-window kernel (ablate 9 = its phase counters): sums over the workgroups that share a counter row (blockIdx & 255), i.e.
        // about the same 6.75 workgroups as one persistent workgroup's items
        CK(hipMemset((char*)scratch + ATTN_SCRATCH_BYTES, 0, 256 * 8 * 8 * 8));
        p.ablate = 9; launch_attention(p, st); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hd.data(), (char*)scratch + ATTN_SCRATCH_BYTES, hd.size() * 8, hipMemcpyDeviceToHost));
        const char* pk[8] = {"decode+issue+qload", "own DMA wait", "barrier", "setup+relpos", "key loops", "stores", "merge+stores", "-"};
        for (int w = 0; w < 4; w += 3) {
            printf("  per-window kernel, wave %d, ticks per 6.75 workgroups (avg):", w);
            double tot = 0;
            for (int k = 0; k < 7; ++k) { double a = 0; for (int b = 0; b < 256; ++b) a += (double)hd[(b * 8 + w) * 8 + k]; a /= 256; tot += a; printf("  %s %.0f", pk[k], a); }
            printf("  | total %.0f\n", tot);
        }
        p.ablate = 0;
    }
    if (argc > 3) return 0;         // windowed only
    AttnParams g = p; g.win = S;
    {   // global attention: the generated-asm kernel (attention_g64.hip) against the HIP kernel (ablate 9), bit for bit
        CK(hipMemset(o0, 0, T * D * 2)); CK(hipMemset(o1, 0xff, T * D * 2));
        g.out = o0; g.ablate = 0; if (launch_attention(g, st)) { printf("launch failed\n"); return 1; }
        g.out = o1; g.ablate = 20; if (launch_attention(g, st)) { printf("launch failed\n"); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h0.data(), o0, T * D * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, T * D * 2, hipMemcpyDeviceToHost));
        double mdg = 0; size_t ndg = 0, nang = 0;
        for (size_t i = 0; i < T * D; ++i) {
            const float a = (float)h0[i], b = (float)h1[i];
            if (std::isnan(b)) ++nang;
            mdg = std::max(mdg, (double)fabsf(a - b)); ndg += a != b;
        }
        printf("  global: asm vs HIP kernel: max |diff| = %.3e, values differing %zu of %zu, NaN %zu\n", mdg, ndg, T * D, nang);
        g.out = o1;
    }
    printf("  global asm ablations:  no softmax VALU %6.1f us   no MFMA %6.1f   no LDS reads %6.1f   no DMA / barrier %6.1f   MFMAs alone %6.1f   softmax VALU alone %6.1f\n",
           run(g, 21, 10, st), run(g, 22, 10, st), run(g, 23, 10, st), run(g, 24, 10, st), run(g, 25, 10, st), run(g, 26, 10, st));
    printf("  global asm:  S^T accumulators in AGPRs (wrong results) %6.1f us (MFMAs alone: %6.1f)   prologue + epilogue only (no key stages) %6.1f\n",
           run(g, 27, 10, st), run(g, 25, 10, st), run(g, 28, 10, st));
    for (int rep = 0; rep < 3; ++rep)
        printf("  global:    asm %6.1f us   HIP (LDS-DMA ring, 3 workgroups / CU) %6.1f   HIP (2 / CU) %6.1f\n", run(g, 20, 10, st), run(g, 0, 10, st), run(g, 13, 10, st));
    return 0;
}

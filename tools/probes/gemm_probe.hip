// GEMM variant probe: times srh::launch_gemm variants on the ViT-B layer shapes (interleaved rounds, random data)
// and checks each against a naive one-thread-per-output kernel.
//   tools/probes/build_probes.sh
//   tools/probes/gemm_probe [rounds] [variants: e.g. 0,40,41,42,43] [shapes: qkv,proj,fc1,fc2]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../sam_road_amd/csrc/kernels.hpp"
using namespace srh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void naive_gemm(const f16* A, const f16* W, const float* bias, const float* resid, int M, int N, int K, int act,
                           float* out, int lda, int ldw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * lda + k] * (float)W[(size_t)n * ldw + k];
    s += bias[n];
    if (act == 1) s = 0.5f * s * (1.0f + erff(s * 0.70710678118654752440f));
    if (resid) s += resid[(size_t)m * N + n];
    out[(size_t)m * N + n] = s;
}
__global__ void diff_kernel(const float* ref, const float* o32, const f16* o16, size_t n, float* maxabs, float* maxref) {
    float d = 0.f, r = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = o32 ? o32[i] : (float)o16[i];
        const float e = fabsf(v - ref[i]);
        d = fmaxf(d, (e == e) ? e : 1e30f);
        r = fmaxf(r, fabsf(ref[i]));
    }
    atomicMax((int*)maxabs, __float_as_int(d));
    atomicMax((int*)maxref, __float_as_int(r));
}

// a streaming kernel between GEMM launches (PROBE_SPOIL=1): the model never runs the same GEMM back to back — LayerNorm / attention
// launches in between change the cache contents, the instruction cache and the clock the next GEMM starts at
__global__ void spoil_kernel(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}

// blocked-16 layout helpers (GemmParams::a_blocked16 / out_blocked16): variants 97 (product body, output blocked) and 98 (product body, A operand blocked)
__global__ void to_blocked16(const f16* rm, f16* blk, int M, int N) {
    const size_t n = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), c = (int)(i % N);
        blk[((size_t)(m / 32) * (N / 16) + c / 16) * 512 + ((c % 16) / 8) * 256 + (m % 32) * 8 + (c % 8)] = rm[i];
    }
}
__global__ void from_blocked16(const f16* blk, f16* rm, int M, int N) {
    const size_t n = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), c = (int)(i % N);
        rm[i] = blk[((size_t)(m / 32) * (N / 16) + c / 16) * 512 + ((c % 16) / 8) * 256 + (m % 32) * 8 + (c % 8)];
    }
}

struct Shape { const char* name; int M, N, K, act; bool resid, f16out; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    std::vector<int> variants;
    { std::string v = argc > 2 ? argv[2] : "0,40"; size_t p = 0; while (p < v.size()) { size_t q = v.find(',', p); if (q == std::string::npos) q = v.size(); variants.push_back(atoi(v.substr(p, q - p).c_str())); p = q + 1; } }
    const std::string which = argc > 3 ? argv[3] : "qkv,proj,fc1,fc2";
    const int T = argc > 4 ? atoi(argv[4]) : 16384;
    const int D = 768;
    const Shape all[] = {{"qkv", T, 3 * D, D, 0, false, true}, {"proj", T, D, D, 0, true, false},
                         {"fc1", T, 4 * D, D, 1, false, true}, {"fc2", T, D, 4 * D, 0, true, false},
                         {"hproj", T, D, D, 0, false, true}, {"hfc2", T, D, 4 * D, 0, false, true},
                         // ViT-H at 256 px, B = 8 (BASELINE configs[4]): M = 2048 tokens, D = 1280 — the small-M regime of gemm_glds_kernel
                         {"vh_qkv", 2048, 3840, 1280, 0, false, true}, {"vh_proj", 2048, 1280, 1280, 0, true, false},
                         {"vh_fc1", 2048, 5120, 1280, 1, false, true}, {"vh_fc2", 2048, 1280, 5120, 0, true, false},
                         // the K slope of a one-round launch (240 tiles of 128 x 256): time per k-tile without launch / epilogue
                         // ViT-L at 256 px, B = 8: D = 1024
                         {"vl_qkv", 2048, 3072, 1024, 0, false, true}, {"vl_proj", 2048, 1024, 1024, 0, true, false},
                         {"vl_fc1", 2048, 4096, 1024, 1, false, true}, {"vl_fc2", 2048, 1024, 4096, 0, true, false},
                         {"kx1", 2048, 3840, 1280, 0, false, true}, {"kx2", 2048, 3840, 2560, 0, false, true}, {"kx4", 2048, 3840, 5120, 0, false, true}};
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (const Shape& s : all) {
        if (("," + which + ",").find(std::string(",") + s.name + ",") == std::string::npos) continue;
        const int pad = getenv("PROBE_LD_PAD") ? atoi(getenv("PROBE_LD_PAD")) : 0;      // extra elements per operand row
        const int LDA = s.K + pad, LDW = s.K + pad;
        const size_t nA = (size_t)s.M * LDA, nW = (size_t)s.N * LDW, nO = (size_t)s.M * s.N;
        std::vector<f16> hA(nA), hW(nW);
        std::vector<float> hb(s.N), hR(s.resid ? nO : 0);
        for (auto& v : hA) v = (f16)(0.5f * nd(rng));
        for (auto& v : hW) v = (f16)(0.05f * nd(rng));
        for (auto& v : hb) v = 0.1f * nd(rng);
        for (auto& v : hR) v = nd(rng);
        f16 *dA, *dW, *o16; float *db, *dR = nullptr, *o32, *ref, *dmax; unsigned long long* ddbg;
        CK(hipMalloc(&ddbg, 256 * 8)); CK(hipMemset(ddbg, 0, 256 * 8));
        CK(hipMalloc(&dA, nA * 2)); CK(hipMalloc(&dW, nW * 2)); CK(hipMalloc(&db, s.N * 4));
        CK(hipMalloc(&o16, nO * 2)); CK(hipMalloc(&o32, nO * 4)); CK(hipMalloc(&ref, nO * 4)); CK(hipMalloc(&dmax, 8));
        if (s.resid) { CK(hipMalloc(&dR, nO * 4)); CK(hipMemcpy(dR, hR.data(), nO * 4, hipMemcpyHostToDevice)); }
        CK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
        naive_gemm<<<dim3((s.N + 255) / 256, s.M), 256>>>(dA, dW, db, dR, s.M, s.N, s.K, s.act, ref, LDA, LDW);
        CK(hipDeviceSynchronize());
        f16 *dA_blk = nullptr, *o16_rm = nullptr;
        if (LDA == s.K) { CK(hipMalloc(&dA_blk, nA * 2)); to_blocked16<<<2048, 256>>>(dA, dA_blk, s.M, s.K); }
        CK(hipMalloc(&o16_rm, nO * 2));
        float* ws = nullptr;                                   // split-K workspace: 4 slices of f32 partials
        CK(hipMalloc(&ws, nO * 4 * 4));
        auto make = [&](int variant) {
            GemmParams g;
            g.A = dA; g.lda = LDA; g.W = dW; g.ldw = LDW; g.M = s.M; g.N = s.N; g.K = s.K; g.bias = db;
            g.resid = dR; g.ldr = s.N; g.act = s.act; g.variant = variant;
            if (s.f16out) { g.out_f16 = o16; g.ldc16 = s.N; } else { g.out_f32 = o32; g.ldc = s.N; }
            g.dbg = ddbg;
            // variant 0 = what the library would do: split-K where its heuristic asks for it; 31 = 128x160 tiles with split-K 4
            if (variant == 0) { GemmParams t = g; t.variant = 0; const int sk = gemm_splitk_factor(t); if (sk > 1) { g.splitk = sk; g.split_ws = ws; } }
            if (variant == 31) { g.splitk = 4; g.split_ws = ws; }
            if (variant == 33) { g.splitk = getenv("PROBE_SPLITK") ? atoi(getenv("PROBE_SPLITK")) : 3; g.split_ws = ws; }
            if (variant == 97) { g.variant = 70; g.out_blocked16 = 1; }
            if (variant > 70 && variant <= 70 + z192_var_count() && z192_var_out_blocked(variant - 70) && s.act == 1) g.out_blocked16 = 1;
            if (variant == 98) { g.variant = 70; g.a_blocked16 = 1; g.A = dA_blk; }
            return g;
        };
        std::vector<std::vector<float>> times(variants.size());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (size_t vi = 0; vi < variants.size(); ++vi) {      // correctness + warm-up
            CK(hipMemset(o16, 0xff, nO * 2)); CK(hipMemset(o32, 0xff, nO * 4)); CK(hipMemset(dmax, 0, 8));
            const GemmParams g = make(variants[vi]);
            const int rc = launch_gemm(g, 0);
            CK(hipDeviceSynchronize());
            const f16* o16_cmp = o16;
            if (g.out_blocked16) { from_blocked16<<<2048, 256>>>(o16, o16_rm, s.M, s.N); o16_cmp = o16_rm; }
            diff_kernel<<<1024, 256>>>(ref, s.f16out ? nullptr : o32, s.f16out ? o16_cmp : nullptr, nO, dmax, dmax + 1);
            float h[2]; CK(hipMemcpy(h, dmax, 8, hipMemcpyDeviceToHost));
            printf("  check %-5s variant %3d rc=%d  max|diff|=%.3e  (max|ref|=%.2f)\n", s.name, variants[vi], rc, h[0], h[1]);
            if (variants[vi] == 53 || variants[vi] == 55) {
                unsigned long long hd[32]; CK(hipMemcpy(hd, ddbg, 25 * 8, hipMemcpyDeviceToHost));
                const double n = (double)hd[24];
                const char* part[6] = {"issue(reads+dma)", "vmcnt wait", "store+lgkm wait", "barrier1", "mfma seg", "barrier2"};
                for (int g = 0; g < 2; ++g) for (int ph = 0; ph < 2; ++ph) {
                    printf("    group %d phase %d cycles/k-tile:", g, ph);
                    for (int q = 0; q < 6; ++q) printf("  %s %.0f", part[q], hd[(g * 2 + ph) * 6 + q] / n);
                    printf("\n");
                }
            }
        }
        static const int spoil = getenv("PROBE_SPOIL") ? atoi(getenv("PROBE_SPOIL")) : 0;
        static float4 *sp_a = nullptr, *sp_b = nullptr;
        const size_t sp_n = (size_t)(64 << 20) / 16;       // 64 MB read + 64 MB written
        if (spoil && !sp_a) { CK(hipMalloc(&sp_a, sp_n * 16)); CK(hipMalloc(&sp_b, sp_n * 16)); CK(hipMemset(sp_a, 0, sp_n * 16)); }
        for (int r = 0; r < rounds; ++r)
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                const GemmParams g = make(variants[vi]);
                launch_gemm(g, 0);
                if (!spoil) {
                    CK(hipEventRecord(e0, 0));
                    for (int it = 0; it < 10; ++it) launch_gemm(g, 0);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    times[vi].push_back(ms / 10);
                } else {            // one GEMM launch at a time, a streaming kernel before each; event pair around the GEMM only
                    float sum = 0.f;
                    for (int it = 0; it < 10; ++it) {
                        spoil_kernel<<<2048, 256>>>(sp_a, sp_b, sp_n);
                        if (spoil >= 2) {       // ... and other large kernels: the instruction cache no longer holds the timed kernel's code
                            for (int ov : {72, 73, 74, 75, 76}) { if (ov != variants[vi]) { const GemmParams og = make(ov); launch_gemm(og, 0); } }
                            spoil_kernel<<<2048, 256>>>(sp_a, sp_b, sp_n);
                        }
                        CK(hipEventRecord(e0, 0));
                        launch_gemm(g, 0);
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        sum += ms;
                    }
                    times[vi].push_back(sum / 10);
                }
            }
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            std::sort(times[vi].begin(), times[vi].end());
            const float med = times[vi][times[vi].size() / 2], mn = times[vi][0];
            const double fl = 2.0 * s.M * (double)s.N * s.K;
            printf("%-5s M=%d N=%d K=%d variant %3d: median %8.1f us (%7.1f TFLOP/s)   min %8.1f us (%7.1f)", s.name, s.M, s.N, s.K,
                   variants[vi], med * 1e3, fl / med / 1e9, mn * 1e3, fl / mn / 1e9);
            if (variants[vi] > 70 && variants[vi] <= 70 + z192_var_count()) {       // z192 probe variants record their shader-clock ticks per workgroup
                CK(hipMemset(ddbg, 0, 256 * 8));
                const GemmParams g = make(variants[vi]);
                hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
                CK(hipEventRecord(a0, 0)); launch_gemm(g, 0); CK(hipEventRecord(a1, 0)); CK(hipEventSynchronize(a1));
                float ms1; CK(hipEventElapsedTime(&ms1, a0, a1));
                unsigned long long hd[256]; CK(hipMemcpy(hd, ddbg, 256 * 8, hipMemcpyDeviceToHost));
                double avg = 0, mx = 0; for (int i = 0; i < 256; ++i) { avg += (double)hd[i]; if ((double)hd[i] > mx) mx = (double)hd[i]; } avg /= 256;
                printf("   ticks/WG avg %.0f max %.0f  (%.2f GHz if the launch is %.1f us)", avg, mx, mx / (ms1 * 1e-3) * 1e-9, ms1 * 1e3);
            }
            printf("\n");
        }
        for (void* q : {(void*)dA, (void*)dW, (void*)db, (void*)o16, (void*)o32, (void*)ref, (void*)dmax, (void*)dR, (void*)ws}) if (q) (void)hipFree(q);
    }
    return 0;
}

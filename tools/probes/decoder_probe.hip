// Fused map_decoder (decoder.hip decode_fused_kernel) timed alone: batch sweep (the per-workgroup weight prologue is the intercept, compute +
// stores the slope) with and without its output stores (scores == logits == nullptr skips them).  Random weights / inputs: timing only.
// Build: tools/probes/build_probes.sh.   tools/probes/decoder_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../../sam_road_amd/csrc/common.hpp"
#include "../../sam_road_amd/csrc/kernels.hpp"
using namespace srh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float run(DecodeFusedParams p, hipStream_t st) {
    const int reps = 20;
    for (int i = 0; i < 3; ++i) launch_decode_fused(p, st);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_decode_fused(p, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / reps);
    }
    return best;
}

int main() {
    const int S = 32, Bmax = 64;
    const size_t T = (size_t)Bmax * S * S;
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<f16> emb(T * 256), frags(336 * 512);
    std::vector<float> prm(768);          // 738 parameters in a 3-KiB block (the kernel stages three 1-KiB pieces)
    for (auto& v : emb) v = (f16)nd(rng);
    for (auto& v : frags) v = (f16)(nd(rng) * 0.05f);
    for (auto& v : prm) v = nd(rng) * 0.1f;
    for (int i = 128; i < 256; ++i) prm[i] = 1.f;
    f16 *de; char* df; float *dp, *ds, *dl;
    CK(hipMalloc(&de, emb.size() * 2)); CK(hipMalloc(&df, frags.size() * 2)); CK(hipMalloc(&dp, prm.size() * 4));
    CK(hipMalloc(&ds, (size_t)Bmax * 512 * 512 * 2 * 4)); CK(hipMalloc(&dl, (size_t)Bmax * 512 * 512 * 2 * 4));
    CK(hipMemcpy(de, emb.data(), emb.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(df, frags.data(), frags.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, prm.data(), prm.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    DecodeFusedParams p;
    p.emb16 = de; p.frags = df; p.prm = dp; p.S = S;
    for (int B : {1, 4, 8, 16, 32, 64}) {
        p.B = B;
        p.scores = ds; p.logits = nullptr; const float a = run(p, st);
        p.scores = ds; p.logits = dl; const float b = run(p, st);
        p.scores = nullptr; p.logits = nullptr; const float c = run(p, st);
        printf("decode_fused B = %2d (%6zu tokens):  scores %6.1f us   scores + logits %6.1f   no stores %6.1f\n", B, (size_t)B * S * S, a, b, c);
    }
    return 0;
}

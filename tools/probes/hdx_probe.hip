// attention_hdx.hip (ViT-H: head dim 80, 14 x 14 windows / the 16 x 16 global window of 256-px tiles) timed alone at the bench shape
// (B = 8, S = 16, 16 heads).  (Its in-kernel ablation switches — staging only, no staging, no key loop ... — were removed from the product
// source in round 6; what they measured is profiles/r05_attention_hdx.txt.)
// Build: tools/probes/build_probes.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../../sam_road_amd/csrc/common.hpp"
#include "../../sam_road_amd/csrc/kernels.hpp"
using namespace srh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float run(AttnParams p, int abl, hipStream_t st) {
    p.ablate = abl;
    const int reps = 50;
    for (int i = 0; i < 3; ++i) launch_attention_hdx(p, st);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_attention_hdx(p, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / reps);
    }
    return best;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, S = 16, heads = 16, HD = 80, D = heads * HD;
    const size_t T = (size_t)B * S * S;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<f16> qkv(T * 3 * D), bias(3 * D), th(31 * HD), tw(31 * HD);
    for (auto& v : qkv) v = (f16)(nd(rng) * 1.5f);
    for (auto& v : bias) v = (f16)(nd(rng) * 0.5f);
    for (auto& v : th) v = (f16)(nd(rng) * 0.3f);
    for (auto& v : tw) v = (f16)(nd(rng) * 0.3f);
    f16 *dq, *db, *dh, *dw, *o0;
    CK(hipMalloc(&dq, qkv.size() * 2)); CK(hipMalloc(&db, bias.size() * 2)); CK(hipMalloc(&dh, th.size() * 2)); CK(hipMalloc(&dw, tw.size() * 2));
    CK(hipMalloc(&o0, T * D * 2));
    CK(hipMemcpy(dq, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), bias.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, th.data(), th.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, tw.data(), tw.size() * 2, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    AttnParams p;
    p.qkv = dq; p.ld = 3 * D; p.table_h = dh; p.table_w = dw; p.bias_qkv = db; p.out = o0; p.ldo = D; p.B = B; p.S = S; p.heads = heads; p.hd = HD;
    p.scale = 1.0f / sqrtf((float)HD);
    for (int win : {14, 16}) {
        p.win = win;
        if (launch_attention_hdx(p, st)) { printf("launch failed\n"); return 1; }
        CK(hipStreamSynchronize(st));
        for (int r = 0; r < 2; ++r)
            printf("hdx win %d (B = %d):  kernel %6.1f us\n", win, B, run(p, 0, st));
    }
    return 0;
}

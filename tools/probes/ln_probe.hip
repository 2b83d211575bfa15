// Standalone LayerNorm bandwidth probe (hipcc --offload-arch=gfx950 -O3 ln_probe.hip -o ln_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum_shfl(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// DPP row reductions + readlane for cross-row
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    // now every lane of a 16-lane row holds the row sum; combine the 4 rows
    const float r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const float r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return (r0 + r1) + (r2 + r3);
}

template <int MODE>
__global__ __launch_bounds__(256) void ln(const float* x, const float* g, const float* b, f16* out, int M) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * 768;
    float v[12];
    for (int c = 0; c < 3; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(xr + (c * 64 + lane) * 4);
        v[c * 4] = t.x; v[c * 4 + 1] = t.y; v[c * 4 + 2] = t.z; v[c * 4 + 3] = t.w;
    }
    float mean = 0.f, rstd = 1.f;
    if (MODE != 1) {
        float s = 0.f;
        for (int e = 0; e < 12; ++e) s += v[e];
        mean = (MODE == 2 ? wave_sum_dpp(s) : wave_sum_shfl(s)) / 768.f;
        float q = 0.f;
        for (int e = 0; e < 12; ++e) { v[e] -= mean; q += v[e] * v[e]; }
        rstd = rsqrtf((MODE == 2 ? wave_sum_dpp(q) : wave_sum_shfl(q)) / 768.f + 1e-6f);
    }
    for (int c = 0; c < 3; ++c) {
        const int off = (c * 64 + lane) * 4;
        const float4 gg = *reinterpret_cast<const float4*>(g + off), bb = *reinterpret_cast<const float4*>(b + off);
        f16x4 h = {(f16)(v[c * 4] * rstd * gg.x + bb.x), (f16)(v[c * 4 + 1] * rstd * gg.y + bb.y),
                   (f16)(v[c * 4 + 2] * rstd * gg.z + bb.z), (f16)(v[c * 4 + 3] * rstd * gg.w + bb.w)};
        *reinterpret_cast<f16x4*>(out + (size_t)row * 768 + off) = h;
    }
}

// MODE 3: flat conversion, grid-stride, 16 B per lane contiguous across the whole block
__global__ __launch_bounds__(256) void conv(const float* x, f16* out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 t = reinterpret_cast<const float4*>(x)[i];
        f16x4 h = {(f16)t.x, (f16)t.y, (f16)t.z, (f16)t.w};
        reinterpret_cast<f16x4*>(out)[i] = h;
    }
}

template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}

int main() {
    const int M = 16384, D = 768;
    float *x, *g, *b; f16* o;
    hipMalloc(&x, (size_t)M * D * 4); hipMalloc(&g, D * 4); hipMalloc(&b, D * 4); hipMalloc(&o, (size_t)M * D * 2);
    std::vector<float> h((size_t)M * D);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(g, h.data(), D * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data() + D, D * 4, hipMemcpyHostToDevice);
    const double bytes = (double)M * D * 6;
    float t;
    t = timeit([&] { hipLaunchKernelGGL(ln<0>, dim3(M / 4), dim3(256), 0, 0, x, g, b, o, M); });
    printf("ln shfl      : %7.1f us %7.1f GB/s\n", t, bytes / t / 1e3);
    t = timeit([&] { hipLaunchKernelGGL(ln<1>, dim3(M / 4), dim3(256), 0, 0, x, g, b, o, M); });
    printf("ln no-reduce : %7.1f us %7.1f GB/s\n", t, bytes / t / 1e3);
    t = timeit([&] { hipLaunchKernelGGL(ln<2>, dim3(M / 4), dim3(256), 0, 0, x, g, b, o, M); });
    printf("ln dpp       : %7.1f us %7.1f GB/s\n", t, bytes / t / 1e3);
    t = timeit([&] { hipLaunchKernelGGL(conv, dim3(2048), dim3(256), 0, 0, x, o, (long)M * D / 4); });
    printf("flat convert : %7.1f us %7.1f GB/s\n", t, bytes / t / 1e3);
    return 0;
}

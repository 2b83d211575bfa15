// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the sam_road hot path.
// Wave = 64 lanes everywhere; MFMA fragments are the gfx950 double-K f16 forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srh {

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// v_mfma_f32_32x32x16_f16: A row = lane&31, B col = lane&31, both hold k = (lane>>5)*8 + j.
// C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)   (MI355X guide §3)
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16: A row = lane&15, B col = lane&15, k = (lane>>4)*8 + j.
// C/D: col = lane&15, row = (lane>>4)*4 + reg
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma32_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// exact-erf GELU without erff:  gelu(x) = max(x,0) - |x| * Phi(-|x|)  with  Phi(-a) = 2^(-r(a)),  r a degree-5
// polynomial fitted (tools/fit_gelu.py) so that |a*2^-r(a) - a*Phi(-a)| <= 1.6e-6 for every a >= 0 — three
// orders of magnitude below the fp16 rounding of the stored activation.  5 FMA + 1 v_exp_f32 + 2 VALU per
// element (A&S 7.1.26 erf needed ~16 VALU + rcp + exp: it made the fc1 epilogue VALU-bound).
__device__ __forceinline__ float gelu_fast(float x) {
    const float a = fabsf(x);
    float r = 0.00048291164585022967f;
    r = fmaf(r, a, -0.0071898452371611365f);
    r = fmaf(r, a, 0.05218537922649359f);
    r = fmaf(r, a, 0.4595148493607732f);
    r = fmaf(r, a, 1.1510354141727006f);
    r = fmaf(r, a, 1.0f);
    const float e = __builtin_amdgcn_exp2f(-r);
    return fmaf(-a, e, fmaxf(x, 0.f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// XOR swizzle of 16-byte chunks inside a 128-byte LDS row (8 chunks): conflict-free
// ds_read_b128 for the 16-lane service groups when lanes read distinct rows mod 16.
__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

}  // namespace srh

#define SRH_CHECK_LAUNCH() (hipGetLastError())

// f16 x f16 -> f32 MFMA GEMM with fused epilogues for the SAM ViT encoder / decoder / TopoNet
// linear layers:  OUT[M,N] = act(A[M,K] * W[N,K]^T + bias) (+ residual | + pos-embed).
//
// Replaces the ATen call sites K2/K4/K7/K8/K9/K10/K13 of SURVEY.md §2.1 (nn.Linear / 1x1 and
// 3x3 Conv2d / ConvTranspose2d-as-GEMM in the un-vendored SAM fork and reference model.py:283-295).
//
// Design (gfx950): 128x128x64 block tile, 256 threads = 4 waves in a 2(N) x 2(M) grid, each wave
// 64x64 via 2x2 v_mfma_f32_32x32x16_f16 tiles (64 accumulator registers).  The MFMA is issued
// "transposed" (A operand = weight rows, B operand = activation rows) so that every lane ends up
// with 4 CONSECUTIVE output columns of one output row: the epilogue then does 16-byte bias /
// residual loads and 8/16-byte stores instead of 2-byte scatters.  Operands are staged
// global -> registers -> LDS (double buffered, loads of tile k+1 issued before the MFMAs of
// tile k) with a 16-byte-chunk XOR swizzle so the ds_read_b128 fragment reads are conflict-free.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage
constexpr int GEMM_LDS = 2 * 2 * TILE_BYTES;     // 64 KiB

// Source row for the activation operand.  AMODE 0: plain row m.  AMODE 1: implicit 3x3 conv over
// an [B,S,S,C] channels-last grid (zero padding 1): k-tile kt addresses tap = (kt*BK)/C.
template <int AMODE>
__device__ __forceinline__ const f16* a_src(const GemmParams& p, int m, int kt, bool& zero) {
    zero = false;
    if (AMODE == 0) {
        return p.A + (size_t)m * p.lda + kt * BK;
    } else {
        const int S = p.conv_S, C = p.conv_C;
        const int k0 = kt * BK;
        const int tap = k0 / C, c0 = k0 - tap * C;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int x = m % S, y = (m / S) % S;
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= S || xx < 0 || xx >= S) { zero = true; return p.A; }
        return p.A + (size_t)(m + dy * S + dx) * p.lda + c0;
    }
}

template <int AMODE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int tiles_n = p.N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / BK;

    // staging assignment: 4 chunks of A and 4 of W per thread per k-tile
    uint4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + GEMM_THREADS * i;
            const int row = id >> 3, c = id & 7;
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            bool zero;
            const f16* src = a_src<AMODE>(p, m, kt, zero);
            ra[i] = zero ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(src + c * 8);
            rw[i] = *reinterpret_cast<const uint4*>(p.W + (size_t)(n0 + row) * p.ldw + kt * BK + c * 8);
        }
    };
    auto store_tile = [&](int stage) {
        char* sa = smem + stage * 2 * TILE_BYTES;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + GEMM_THREADS * i;
            const int row = id >> 3, c = id & 7;
            const int off = row * 128 + swz8(row, c) * 16;
            *reinterpret_cast<uint4*>(sa + off) = ra[i];
            *reinterpret_cast<uint4*>(sw + off) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* sa = smem + stage * 2 * TILE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 fw[2], fx[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rw_ = wn * 64 + i * 32 + frow;
                fw[i] = *reinterpret_cast<const f16x8*>(sw + rw_ * 128 + swz8(rw_, ks * 2 + fhalf) * 16);
                const int rx_ = wm * 64 + i * 32 + frow;
                fx[i] = *reinterpret_cast<const f16x8*>(sa + rx_ * 128 + swz8(rx_, ks * 2 + fhalf) * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(stage ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds, per (i,j,q), 4 consecutive columns n..n+3 of row m
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * fhalf;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.pos) {
                    const float4 b = *reinterpret_cast<const float4*>(p.pos + (size_t)(m % p.pos_rows) * p.N + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (p.resid) {
                    const float4 b = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (p.out_f32)
                    *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                if (p.out_f16) {
                    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h;
                }
            }
        }
    }
}

int launch_gemm(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0) return 0;
    if (p.N % BN != 0 || p.K % BK != 0) return -2;
    if (p.conv_S > 0 && (p.conv_C % BK != 0)) return -2;
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    if (p.conv_S > 0)
        hipLaunchKernelGGL(gemm_kernel<1>, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS, stream, p);
    else
        hipLaunchKernelGGL(gemm_kernel<0>, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

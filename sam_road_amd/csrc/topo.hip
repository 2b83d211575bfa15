// BilinearSampler + TopoNet glue kernels.
//
// Reference: model.py:29-58 (F.grid_sample bilinear, align_corners=False, zero padding, on
// [B,256,h,w]) and model.py:88-148 (TopoNet: gather src/tgt point features + offset, 3 post-LN
// nn.TransformerEncoderLayer over 16-pair sequences with key-padding mask, all-invalid rows flipped
// to all-valid, Linear(128->1), sigmoid).  The linear layers run on gemm.hip; this file holds the
// gathers and the 16-token attention.  Embeddings are kept channels-last so a point's 256 channels
// are one contiguous 1 KiB row per bilinear corner (coalesced float4 per lane) instead of the
// reference layout's 256 strided scalars.
#include "common.hpp"
#include "kernels.hpp"

namespace srh {

__device__ __forceinline__ float point_coord(const void* pts, int i64, size_t idx) {
    return i64 ? (float)reinterpret_cast<const long long*>(pts)[idx] : reinterpret_cast<const float*>(pts)[idx];
}

// one wave per point, lane = 4 channels
__global__ __launch_bounds__(256) void sample_kernel(SampleParams p) {
    const int lane = threadIdx.x & 63;
    const long pt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= (long)p.B * p.N) return;
    const int b = p.point_tile ? min(max(p.point_tile[pt], 0), max(p.n_tiles, 1) - 1) : (int)(pt / p.N);   // a bad tile index must not read out of bounds
    const float px = point_coord(p.points, p.points_i64, pt * 2), py = point_coord(p.points, p.points_i64, pt * 2 + 1);
    // model.py:47 then ATen grid_sampler_unnormalize (align_corners=False)
    const float gx = (px / p.patch) * 2.0f - 1.0f, gy = (py / p.patch) * 2.0f - 1.0f;
    const float ix = ((gx + 1.f) * p.w - 1.f) / 2.f, iy = ((gy + 1.f) * p.h - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy);
    const float wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const float* base = p.emb + (size_t)b * p.h * p.w * p.C;
    for (int c = lane * 4; c < p.C; c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        auto tap = [&](int yy, int xx, float wgt) {
            if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
                const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * p.w + xx) * p.C + c);
                acc.x += v.x * wgt; acc.y += v.y * wgt; acc.z += v.z * wgt; acc.w += v.w * wgt;
            }
        };
        tap(y0, x0, wnw); tap(y0, x1, wne); tap(y1, x0, wsw); tap(y1, x1, wse);
        if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + pt * p.C + c) = acc;
        if (p.out_f16) {
            f16x4 h = {(f16)acc.x, (f16)acc.y, (f16)acc.z, (f16)acc.w};
            *reinterpret_cast<f16x4*>(p.out_f16 + pt * p.C + c) = h;
        }
    }
}

int launch_sample(const SampleParams& p, hipStream_t s) {
    const long n = (long)p.B * p.N;
    if (n <= 0) return 0;
    if (p.C % 4) return -2;
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// pair row = [src feat 128 | tgt feat 128 | dx dy | zero pad]   (model.py:104-116)
__global__ __launch_bounds__(256) void pair_gather_kernel(PairGatherParams p) {
    const int sub = threadIdx.x & 31;                       // 32 threads per pair row
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const long rows = (long)p.B * p.Ns * p.Kp;
    if (row >= rows) return;
    const int b = row / ((long)p.Ns * p.Kp);
    long src, tgt;
    if (p.pairs_i64) {
        src = reinterpret_cast<const long long*>(p.pairs)[row * 2];
        tgt = reinterpret_cast<const long long*>(p.pairs)[row * 2 + 1];
    } else {
        src = reinterpret_cast<const int*>(p.pairs)[row * 2];
        tgt = reinterpret_cast<const int*>(p.pairs)[row * 2 + 1];
    }
    src -= p.index_base; tgt -= p.index_base;
    // python-style negative index wrap, then clamp (out-of-range is an error in the reference)
    if (src < 0) src += p.N;
    if (tgt < 0) tgt += p.N;
    src = min(max(src, 0L), (long)p.N - 1);
    tgt = min(max(tgt, 0L), (long)p.N - 1);
    f16* out = p.out + row * p.ld;
    const long which = sub < 16 ? src : tgt;
    const uint4 v = *reinterpret_cast<const uint4*>(p.pf + ((size_t)b * p.N + which) * 128 + (sub & 15) * 8);
    *reinterpret_cast<uint4*>(out + sub * 8) = v;
    // tail: offset + zero padding, 8 halfs per thread
    const int tail_chunks = (p.ld - 256) / 8;
    if (sub < tail_chunks) {
        f16x8 t;
        for (int e = 0; e < 8; ++e) t[e] = (f16)0.f;
        if (sub == 0 && !p.zero_offset) {
            const size_t ps = ((size_t)b * p.N + src) * 2, pt = ((size_t)b * p.N + tgt) * 2;
            if (p.points_i64) {
                const long long* q = reinterpret_cast<const long long*>(p.points);
                t[0] = (f16)(float)(q[pt] - q[ps]);
                t[1] = (f16)(float)(q[pt + 1] - q[ps + 1]);
            } else {
                const float* q = reinterpret_cast<const float*>(p.points);
                t[0] = (f16)(q[pt] - q[ps]);
                t[1] = (f16)(q[pt + 1] - q[ps + 1]);
            }
        }
        *reinterpret_cast<f16x8*>(out + 256 + sub * 8) = t;
    }
}

int launch_pair_gather(const PairGatherParams& p, hipStream_t s) {
    const long rows = (long)p.B * p.Ns * p.Kp;
    if (rows <= 0) return 0;
    if (p.ld < 264 || p.ld % 8 || (p.ld - 256) / 8 > 32) return -2;
    hipLaunchKernelGGL(pair_gather_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace srh

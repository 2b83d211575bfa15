#!/usr/bin/env python
"""Per-kernel micro-benchmarks at the BASELINE config-2 shapes (B=16 tiles of 512^2, ViT-B) through the
op-level C ABI.  Prints achieved TFLOP/s (or GB/s) per op.  Usage: python tools/opbench.py [gemm attn ln]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_road_amd import _lib  # noqa: E402


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    which = sys.argv[1:] or ["gemm", "attn", "ln"]
    ctx = _lib.Context.get(0)
    lib, h = ctx.lib, ctx.handle
    T, D = 16 * 1024, 768
    if "gemm" in which:
        for name, M, N, K, act, resid, f16out in [("qkv", T, 3 * D, D, 0, False, True), ("proj", T, D, D, 0, True, False),
                                                 ("fc1", T, 4 * D, D, 1, False, True), ("fc2", T, D, 4 * D, 0, True, False),
                                                 ("dec2", 16 * T, 128, 64, 1, False, True), ("topo", 65536, 384, 128, 0, False, True)]:
            A = (torch.randn(M, K, device="cuda") * 0.5).half()
            W = (torch.randn(N, K, device="cuda") * 0.05).half()
            bias = torch.randn(N, device="cuda")
            R = torch.randn(M, N, device="cuda") if resid else None
            o32 = torch.empty(M, N, device="cuda") if not f16out else None
            o16 = torch.empty(M, N, device="cuda", dtype=torch.half) if f16out else None
            ms = timeit(lambda: ctx.check(lib.srh_op_gemm(h, p(A), p(W), p(bias), p(R), M, N, K, act, p(o32), p(o16), None), "gemm"))
            print(f"gemm {name:5s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s")
    if "attn" in which:
        B, S, heads = 16, 32, 12
        qkv = (torch.randn(B * S * S, 3 * D, device="cuda") * 1.0).half()
        bias = (torch.randn(3 * D, device="cuda") * 0.5).half()
        out = torch.empty(B * S * S, D, device="cuda", dtype=torch.half)
        for win in (14, 32):
            rh = (torch.randn(2 * win - 1, 64, device="cuda") * 0.3).half()
            rw = (torch.randn(2 * win - 1, 64, device="cuda") * 0.3).half()
            ms = timeit(lambda: ctx.check(lib.srh_op_attention(h, p(qkv), p(rh), p(rw), p(bias), B, S, heads, win, p(out), None), "attn"))
            fl = 4.0 * B * heads * 1024 * (196 if win == 14 else 1024) * 64
            print(f"attention win={win}: {ms*1e3:8.1f} us (relpos+attn)  {fl/ms/1e9:8.1f} TFLOP/s")
    if "ln" in which:
        x = torch.randn(T, D, device="cuda")
        g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
        o16 = torch.empty(T, D, device="cuda", dtype=torch.half)
        ms = timeit(lambda: ctx.check(lib.srh_op_layernorm(h, p(x), p(g), p(b), 1e-6, T, D, 0, None, p(o16), None), "ln"))
        print(f"layernorm {T}x{D}: {ms*1e3:8.1f} us  {T*D*6/ms/1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()


def calib():
    """Calibrate achievable HBM bandwidth with torch's own streaming kernels."""
    x = torch.randn(16384, 768, device="cuda")
    big = torch.randn(64 * 1024 * 1024, device="cuda")   # 256 MB
    ms = timeit(lambda: x.half())
    print(f"torch f32->f16 16384x768: {ms*1e3:7.1f} us  {16384*768*6/ms/1e6:8.1f} GB/s")
    ms = timeit(lambda: big.clone())
    print(f"torch clone 256MB: {ms*1e3:7.1f} us  {big.numel()*8/ms/1e6:8.1f} GB/s")
    ms = timeit(lambda: x.clone())
    print(f"torch clone 50MB: {ms*1e3:7.1f} us  {x.numel()*8/ms/1e6:8.1f} GB/s")
    ms = timeit(lambda: torch.nn.functional.layer_norm(x, (768,)))
    print(f"torch layer_norm f32 16384x768: {ms*1e3:7.1f} us  {x.numel()*8/ms/1e6:8.1f} GB/s")


if "calib" in sys.argv:
    calib()

"""Yardstick only (not used by the product path): what does the vendor library (hipBLASLt through torch.matmul) reach on the
encoder's GEMM shapes on this GPU?  fp16 in, fp16 out, fp32 accumulate; bias fused via addmm where the library allows."""
import torch, time
dev = "cuda"
shapes = {"qkv": (16384, 2304, 768), "proj": (16384, 768, 768), "fc1": (16384, 3072, 768), "fc2": (16384, 768, 3072)}
for name, (M, N, K) in shapes.items():
    a = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev).half()
    for label, fn in (("matmul", lambda: a @ w.t()), ("addmm(bias)", lambda: torch.addmm(b, a, w.t()))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"{name:5s} M={M} N={N} K={K} {label:12s} {best*1e3:8.1f} us  {2*M*N*K/best/1e9:8.1f} TFLOP/s", flush=True)

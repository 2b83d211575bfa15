"""Time srh_op_attention alone (global 32x32 window and 14x14 windows at the bench shape B=16, S=32, 12 heads) under the
kernel's ablation switches (SRH_ATTN_ABL: 0 full, 1 no key loop, 2 no K/V staging, 3 no rel-pos).  The switch is read once
per process, so the script re-runs itself per value.  Run on the GPU box: python tools/attn_probe.py"""
import ctypes as C, os, subprocess, sys

def one():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sam_road_amd import _lib
    ctx = _lib.Context.get(0)
    B, S, heads = 16, 32, 12
    D = heads * 64
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B * S * S, 3 * D, generator=g) * 1.5).half().cuda()
    bias = (torch.randn(3 * D, generator=g) * 0.5).half().cuda()
    out = torch.zeros((B * S * S, D), device="cuda", dtype=torch.half)
    p = lambda t: C.c_void_p(t.data_ptr())
    res = []
    for win in (32, 14):
        rh = (torch.randn(2 * win - 1, 64, generator=g) * 0.3).half().cuda()
        rw = (torch.randn(2 * win - 1, 64, generator=g) * 0.3).half().cuda()
        call = lambda: ctx.check(ctx.lib.srh_op_attention(ctx.handle, p(qkv), p(rh), p(rw), p(bias), B, S, heads, win, p(out), None), "attn")
        for _ in range(3): call()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.default_stream())
            for _ in range(10): call()
            e1.record(torch.cuda.default_stream()); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res.append(f"win={win}: {best * 1e3:7.1f} us")
    print(f"SRH_ATTN_ABL={os.environ.get('SRH_ATTN_ABL', '0')} {os.environ.get('ATTN_PROBE_TAG', '')}  " + "   ".join(res), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for a in (sys.argv[1:] or ["0", "1", "2", "3"]):
            subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, SRH_ATTN_ABL=a), check=False)

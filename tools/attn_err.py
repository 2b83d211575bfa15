"""Error statistics of srh_op_attention against the fp32 reference used by tests/test_gpu_ops.py (larger sample)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import ref_sam_attention
from sam_road_amd import _lib
ctx = _lib.Context.get(0)
p = lambda t: C.c_void_p(t.data_ptr())
for S, win in [(32, 14), (32, 32), (16, 14), (16, 16)]:
    B, heads = 4, 6
    D = heads * 64
    g = torch.Generator().manual_seed(S * 100 + win + 7)
    qkv = (torch.randn(B * S * S, 3 * D, generator=g) * 1.5).half()
    bias = (torch.randn(3 * D, generator=g) * 0.5).half()
    rel_h = (torch.randn(2 * win - 1, 64, generator=g) * 0.3).half()
    rel_w = (torch.randn(2 * win - 1, 64, generator=g) * 0.3).half()
    ref = ref_sam_attention(qkv, rel_h, rel_w, bias, B, S, heads, win)
    out = torch.zeros((B * S * S, D), device="cuda", dtype=torch.half)
    dq, dh, dw, db = qkv.cuda(), rel_h.cuda(), rel_w.cuda(), bias.cuda()
    ctx.check(ctx.lib.srh_op_attention(ctx.handle, p(dq), p(dh), p(dw), p(db), B, S, heads, win, p(out), None), "attn")
    torch.cuda.synchronize()
    err = (out.cpu().float() - ref).abs()
    print(f"S={S} win={win}: max {err.max().item():.3e} mean {err.mean().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e}", flush=True)

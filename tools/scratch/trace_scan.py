#!/usr/bin/env python
"""Scratch: per-pass-1 (delimited by patch_im2col launches in groups of 4 batches) busy time, span and per-class kernel time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows), key=lambda t: t[0])
# find scene_normalise kernels as scene delimiters
idx = [i for i, k in enumerate(ks) if "scene_norm_kernel" in k[2]]
print("scenes:", len(idx))
prev = 0
def cls(n):
    for key in ("gemm_q192_kernel<0, 1", "gemm_q192", "layernorm", "attn_window", "attn_global", "gemm", "scene_add", "patch_im2col", "decode", "topo", "sample"):
        if key in n: return key
    return "other"
for si, i in enumerate(idx):
    # walk back to the first patch_im2col of this scene's pass 1 (4 batches -> 4 im2col kernels)
    seg = ks[prev:i + 1]
    p = [j for j, k in enumerate(seg) if "patch_im2col" in k[2]]
    if not p: prev = i + 1; continue
    seg = seg[p[0]:]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[j + 1][0] - seg[j][1]) for j in range(len(seg) - 1)), reverse=True)
    per = collections.defaultdict(float)
    for s, e, n in seg: per[cls(n)] += (e - s) / 1e6
    print(f"scene {si:2d}: kernels {len(seg):4d} span {span/1e6:6.1f} ms busy {busy/1e6:6.1f} ms  top gaps us {[round(g/1e3) for g in gaps[:4]]}  " +
          " ".join(f"{k}={v:.1f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:6]))
    prev = i + 1

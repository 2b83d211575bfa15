"""sha256 of the hot path's outputs (mask scores, image embedding) of every bench workload on seeded weights and inputs — to compare two
BUILDS of this repo bit for bit (run it in each tree on the same GPU box):  python tools/emb_digest.py [workload ...]
Used in round 5 to check that the ping-pong GEMM and the LayerNorm-folded split-K reduce change no bit of the ViT-H / ViT-L / ViT-B
outputs against the previous build (same k-order per accumulator, same slice boundaries, same order of the slice sums)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    names = sys.argv[1:] or ["encdec", "vith256", "vitl256"]
    dev = torch.device("cuda:0")
    for name in names:
        wl = name
        if name == "vitl256":            # not a bench workload: ViT-L at 256 px, B = 8 (the other small-M architecture)
            bench.WORKLOADS["vitl256"] = dict(bench.WORKLOADS["vith256"], version="vit_l")
        net, sd, cfg, step, B, P, WL = bench.build_workload(wl, 0, 0, dev, False)
        ms, emb = step()
        torch.cuda.synchronize()
        h = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{name}: B={B} P={P} mask_scores {h(ms)} emb {h(emb)}  (emb mean |.| {emb.float().abs().mean().item():.6f})")
        del net


if __name__ == "__main__":
    main()

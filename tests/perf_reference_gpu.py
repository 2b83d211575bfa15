#!/usr/bin/env python
"""Times the ORACLE (the reference's eager PyTorch op sequence) on the MI355X with stock PyTorch-ROCm:
this is "the reference PyTorch path" the north star's >=4x target is quoted against.  Not a pytest
file (no test_ prefix); lives under tests/ because only tests may import oracle/.
    python tests/perf_reference_gpu.py [--batch 16]"""
import argparse
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.samroad import AttrDict, SAMRoadOracle  # noqa: E402
from oracle.synth import synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    cfg = AttrDict(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal")
    net = SAMRoadOracle(cfg).eval()
    net.load_state_dict(synth_state_dict(net, 1234))
    net.cuda()
    x = (torch.rand((args.batch, 512, 512, 3)) * 255).round().cuda()
    for name, ctx in (("eager fp32", torch.autocast("cuda", enabled=False)),
                      ("autocast bf16", torch.autocast("cuda", dtype=torch.bfloat16)),
                      ("autocast fp16", torch.autocast("cuda", dtype=torch.float16))):
        with ctx:
            for _ in range(2):
                net.infer_masks_and_img_features(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                net.infer_masks_and_img_features(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
        print(f"reference-on-GPU ({name}, B={args.batch}): {dt*1e3:.2f} ms/step  {args.batch/dt:.1f} tiles/s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Scratch experiment: why does pass 1 stretch from 75 to ~105 ms on the device when the host works in parallel (infer_imgs)?
Variants: host stages as they are / worker threads limited to 1 / uploads on the compute stream / host stages replaced by a sleep."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_road_amd import Config, SAMRoad
import sam_road_amd.inferencer as inf

dev = torch.device("cuda", 0)
cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", DATASET="cityscale",
             INFER_BATCH_SIZE=64, SAMPLE_MARGIN=64, INFER_PATCHES_PER_EDGE=16, ITSC_THRESHOLD=0.248,
             ROAD_THRESHOLD=0.364, TOPO_THRESHOLD=0.499, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64,
             MAX_NEIGHBOR_QUERIES=16)
import warnings; warnings.simplefilter("ignore")
net = SAMRoad(cfg)
g = torch.Generator().manual_seed(1234)
sd = {}
for k, v in net.state_dict().items():
    sd[k] = (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if (v.dim() == 1 and k.endswith("weight")) else 0.02 * torch.randn(v.shape, generator=g)
sd["map_decoder.7.weight"] = 16.0 * torch.randn(sd["map_decoder.7.weight"].shape, generator=g)
sd["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], -2.2)
net.load_state_dict(sd, strict=True); net.eval().to(dev)
rng = np.random.default_rng(0)
img = np.kron(rng.integers(0, 256, size=(256, 256, 3)).astype(np.float32), np.ones((8, 8, 1), np.float32)).astype(np.uint8)

def cpu_snapshot():
    per = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
            per[tid] = (int(f[11]) + int(f[12])) * 10.0      # utime + stime, ms (CLK_TCK 100)
        except OSError:
            pass
    thr = 0
    try:
        for l in open("/sys/fs/cgroup/cpu.stat"):
            if l.startswith("throttled_usec"): thr = int(l.split()[1]) / 1e3
    except OSError:
        pass
    return per, thr

def run(tag, n=16):
    list(inf.infer_imgs(net, (img for _ in range(3)), cfg))
    torch.cuda.synchronize(); c0, th0 = cpu_snapshot(); t0 = time.perf_counter()
    list(inf.infer_imgs(net, (img for _ in range(n)), cfg))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c1, th1 = cpu_snapshot()
    d = sorted(((c1[k] - c0.get(k, 0.0)) for k in c1), reverse=True)
    ms_ = torch.cuda.memory_stats()
    print("   device mallocs so far", ms_.get("num_device_alloc"), "frees", ms_.get("num_device_free"), "reserved MB", ms_.get("reserved_bytes.all.current", 0) >> 20,
          "active MB", ms_.get("active_bytes.all.current", 0) >> 20)
    print(f"{tag}: {wall / n * 1e3:.1f} ms/scene | wall {wall*1e3:.0f} ms, cpu {sum(d):.0f} ms over {sum(x > 0 for x in d)} busy threads (top: {[int(x) for x in d[:6]]}), "
          f"throttled {th1 - th0:.0f} ms, threads alive {len(c1)}", flush=True)

print("torch threads", torch.get_num_threads())
for rep in range(6):
    run(f"as is, default torch threads, repetition {rep}", n=12)


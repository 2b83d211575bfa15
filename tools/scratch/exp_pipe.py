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

def run(tag, n=16):
    list(inf.infer_imgs(net, (img for _ in range(3)), cfg))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    list(inf.infer_imgs(net, (img for _ in range(n)), cfg))
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.1f} ms/scene", flush=True)

run("as is")
real_cpu_count = os.cpu_count
os.cpu_count = lambda: 2
torch.set_num_threads(1)
run("1 worker thread in srh_pass2_fill, torch 1 thread")
os.cpu_count = real_cpu_count
orig = inf._Lane.upload_staged
def on_main(self, stage):
    dst = torch.empty(stage.shape, dtype=stage.dtype, device=self.device)
    dst.copy_(stage, non_blocking=True)
    return dst
inf._Lane.upload_staged = on_main
run("uploads on the compute stream")
inf._Lane.upload_staged = orig
# host stages replaced by sleeps of the same length: does CPU load matter?
egp, bq = inf.extract_graph_points, inf.build_all_patch_queries
cache = {}
def egp_s(*a, **k):
    if "p" not in cache: cache["p"] = egp(*a, **k)
    time.sleep(0.012); return cache["p"]
def bq_s(*a, **k):
    if "q" not in cache: cache["q"] = bq(*a, **k)
    time.sleep(0.018); return cache["q"]
inf.extract_graph_points, inf.build_all_patch_queries = egp_s, bq_s
run("points + queries replaced by sleeps (cached results)")
inf.extract_graph_points, inf.build_all_patch_queries = egp, bq
print("cpu_count", real_cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu.stat"):
    if os.path.exists(f): print(f, open(f).read().strip().replace("\n", " | "))

#!/usr/bin/env python
"""Scratch: where do the slow first seconds of the pipelined scene loop come from?  Per-scene DEVICE time of pass 1 (events) in
(A) a bare back-to-back pass-1 loop from cold, (B) the pipelined loop with host stages replaced by sleeps, (C) the real loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_road_amd import Config, SAMRoad
import sam_road_amd.inferencer as inf
from sam_road_amd.hostcpu import usable_cpus
torch.set_num_threads(usable_cpus())
order = sys.argv[1] if len(sys.argv) > 1 else "ABC"
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
from sam_road_amd.tiling import get_patch_info_one_img
infos = get_patch_info_one_img(0, 2048, 64, 512, 16)
xy = torch.as_tensor(np.array([[p[1][0], p[1][1]] for p in infos], dtype=np.int32)).to(dev)
scene = torch.as_tensor(img).to(dev)
res0 = inf.infer_one_img(net, img, cfg)     # one serial scene: everything allocated / packed

def phase_a(n=36):
    evs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); kp, road, emb = net.scene_pass1(scene, xy, 64); net.scene_normalise(kp, road, xy); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    print(f"A bare pass-1 loop: {(time.perf_counter()-t0)/n*1e3:.1f} ms/scene; device ms per scene:", " ".join(f"{a.elapsed_time(b):.0f}" for a, b in evs), flush=True)

# device-side pass-1 time per scene inside infer_imgs: wrap scene_pass1 / scene_normalise with events
marks = []
orig_p1, orig_norm = net.scene_pass1, net.scene_normalise
def p1(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append([e, None]); return orig_p1(*a, **k)
def norm(*a, **k):
    r = orig_norm(*a, **k); e = torch.cuda.Event(enable_timing=True); e.record(); marks[-1][1] = e; return r

def piped(tag, n=36):
    net.scene_pass1, net.scene_normalise = p1, norm
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    list(inf.infer_imgs(net, (img for _ in range(n)), cfg))
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter()-t0)/n*1e3:.1f} ms/scene; device pass-1 ms per scene:", " ".join(f"{a.elapsed_time(b):.0f}" for a, b in marks), flush=True)
    net.scene_pass1, net.scene_normalise = orig_p1, orig_norm

egp, bq = inf.extract_graph_points, inf.build_all_patch_queries
cache = {}
def egp_s(*a, **k):
    if "p" not in cache: cache["p"] = egp(*a, **k)
    time.sleep(0.012); return cache["p"]
def bq_s(*a, **k):
    if "q" not in cache: cache["q"] = bq(*a, **k)
    time.sleep(0.018); return cache["q"]
for ph in order:
    if ph == "A":
        phase_a()
    elif ph == "B":
        inf.extract_graph_points, inf.build_all_patch_queries = egp_s, bq_s
        piped("B pipelined, points + queries replaced by sleeps")
        inf.extract_graph_points, inf.build_all_patch_queries = egp, bq
    elif ph == "C":
        piped("C pipelined, real host stages")
    elif ph == "S":
        time.sleep(2.0); print("slept 2 s", flush=True)

#!/usr/bin/env python
"""Regression aid: infer_one_img on the scene of tools/dump_scene.py must reproduce the nodes and the edge LIST (order included) saved
in tools/_scene_dump.npz (written by an earlier build).  GPU needed."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_road_amd import Config, SAMRoad
from sam_road_amd import inferencer as I
d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_scene_dump.npz"))
cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", DATASET="cityscale",
             INFER_BATCH_SIZE=64, SAMPLE_MARGIN=64, INFER_PATCHES_PER_EDGE=16, ITSC_THRESHOLD=0.248, ROAD_THRESHOLD=0.364, TOPO_THRESHOLD=0.499,
             ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
net = SAMRoad(cfg); g = torch.Generator().manual_seed(1234); sd = {}
for k, v in net.state_dict().items():
    sd[k] = (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if (v.dim() == 1 and k.endswith("weight")) else 0.02 * torch.randn(v.shape, generator=g)
sd["map_decoder.7.weight"] = 16.0 * torch.randn(sd["map_decoder.7.weight"].shape, generator=g)
sd["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], -2.2)
net.load_state_dict(sd); net.eval().to("cuda")
rng = np.random.default_rng(0)
coarse = rng.integers(0, 256, size=(256, 256, 3)).astype(np.float32)
img = np.kron(coarse, np.ones((8, 8, 1), np.float32)).astype(np.uint8)
nodes, edges, kp, road = I.infer_one_img(net, img, cfg)
assert np.array_equal(kp, d["kp"]) and np.array_equal(road, d["road"]), "masks differ"
assert np.array_equal(nodes, d["nodes"]), "nodes differ"
assert np.array_equal(edges, d["edges"]), "edge list differs"
got = list(I.infer_imgs(net, iter([img, img]), cfg))
assert all(np.array_equal(r[1], d["edges"]) for r in got), "pipelined edge list differs"
print("scene reproduces the saved result:", len(nodes), "nodes", len(edges), "edges (serial and pipelined)")

#!/usr/bin/env python
"""Development aid: run the scene_bench scene once on the GPU and save the host-stage inputs (u8 masks, graph points, TopoNet
scores per batch) to gpurun_out/scene_dump.npz, so that the host stages can be profiled and rewritten on a machine without a GPU."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_road_amd import Config, SAMRoad
from sam_road_amd import inferencer as I
from sam_road_amd.graph_points import extract_graph_points

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
img, infos, all_xy = I._scene_plan(img, cfg)
dev = torch.device("cuda")
xy = torch.as_tensor(all_xy).to(dev); scene = torch.as_tensor(img).to(dev)
kp_c, road_c, emb = net.scene_pass1(scene, xy, 64)
kp_u8, road_u8 = net.scene_normalise(kp_c, road_c, xy)
kp, road = kp_u8.cpu().numpy(), road_u8.cpu().numpy()
gp = extract_graph_points(kp, road, cfg)
K = 16
fq = I.build_all_patch_queries(gp, infos, 0, len(infos), cfg, flat=True)
plan, pts_h, pairs_h, valid_h = I._pack_pass2_batches(fq, 0, len(infos), 64, K, sort_tiles=False)      # batches of consecutive tiles
pts_d, pairs_d, valid_d = (torch.from_numpy(x).to(dev) for x in (pts_h, pairs_h, valid_h))
launched = I._launch_pass2_batches(net, emb, plan, pts_d, pairs_d, valid_d, K, 0)
out = {"kp": kp, "road": road, "graph_points": gp, "offsets": fq.offsets, "ids": fq.ids, "knn": fq.knn, "tied": fq.tied,
       "plan": np.array([(int(t[0]), int(t[-1]) + 1, n_max, base) for t, n_max, base in plan], dtype=np.int64)}
for i, (tiles, sc) in enumerate(launched):
    out[f"scores{i}"] = sc.cpu().numpy()
nodes, edges, _, _ = I.infer_one_img(net, img, cfg)
out["nodes"], out["edges"] = nodes, edges
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/scene_dump.npz", **out)
print({k: (v.shape, str(v.dtype)) for k, v in out.items()})

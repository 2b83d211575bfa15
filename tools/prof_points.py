import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sam_road_amd import Config, SAMRoad
from sam_road_amd import graph_points as gp
from sam_road_amd.inferencer import _scene_plan
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
img, infos, all_xy = _scene_plan(img, cfg)
xy = torch.as_tensor(all_xy).cuda(); scene = torch.as_tensor(img).cuda()
kp_c, road_c, emb = net.scene_pass1(scene, xy, 64)
kp_u8, road_u8 = net.scene_normalise(kp_c, road_c, xy)
kp, road = kp_u8.cpu().numpy(), road_u8.cpu().numpy()
def T(f, n=8):
    f(); t = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t) / n * 1e3, r
print("extract_graph_points", T(lambda: gp.extract_graph_points(kp, road, cfg))[0])
for name, m, thr, rad in (("kp", kp, cfg.ITSC_THRESHOLD * 255, 8), ("road", road, cfg.ROAD_THRESHOLD * 255, 16)):
    t, (c, s) = T(lambda: gp.points_and_scores_from_mask(m, thr)); print(name, "candidates", t, c.shape)
    t, o = T(lambda: np.argsort(s)[::-1]); print(name, "argsort u8", t)
    t, _ = T(lambda: c[o, :]); print(name, "gather", t)
    t, k = T(lambda: gp.nms_points(c, s, rad)); print(name, "nms_points total", t, k.shape)
    if name == "kp": k0 = k
    else: k1 = k
cand = np.concatenate([k0, k1], 0); prio = np.concatenate([np.ones(len(k0)), np.zeros(len(k1))])
t, o = T(lambda: np.argsort(prio)[::-1]); print("argsort prio", t, len(prio))
t, r = T(lambda: gp.nms_points(cand, prio, 16)); print("final nms_points", t, r.shape)

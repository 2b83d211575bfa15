"""Diagnostic for tests/test_gpu_scene.py::test_infer_one_img_end_to_end: distribution of (HIP - oracle) mean edge scores."""
import os, sys, warnings
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_scene import CFG, SCENE
from oracle import scene as oscene
from oracle.samroad import AttrDict, SAMRoadOracle
from oracle.synth import synth_scene, synth_state_dict
from sam_road_amd import Config, SAMRoad
from sam_road_amd.graph_points import extract_graph_points
from sam_road_amd.inferencer import edge_votes
from sam_road_amd.tiling import get_patch_info_one_img
warnings.simplefilter("ignore")
oracle = SAMRoadOracle(AttrDict(CFG)).eval()
sd = synth_state_dict(oracle, 77); sd["map_decoder.7.bias"] = torch.tensor([-0.3, 0.2])
oracle.load_state_dict(sd, strict=True)
net = SAMRoad(Config(CFG)); net.load_state_dict(sd, strict=True); net.eval().to("cuda")
for seed in (6, 7, 8):
    img = synth_scene(SCENE, seed=seed)
    cfg = dict(CFG)
    infos, feats, kp_r, road_r = oscene.infer_pass1(oracle, img, AttrDict(cfg))
    cfg["ITSC_THRESHOLD"] = float(np.percentile(kp_r[kp_r > 0], 99.5)) / 255.0
    cfg["ROAD_THRESHOLD"] = float(np.percentile(road_r[road_r > 0], 98.0)) / 255.0
    c = Config(cfg)
    xy = torch.tensor([[p[1][0], p[1][1]] for p in infos], dtype=torch.int32).cuda()
    kp_c, road_c, emb = net.scene_pass1(torch.as_tensor(img).cuda(), xy, c.INFER_BATCH_SIZE)
    kp, road = [t.cpu().numpy() for t in net.scene_normalise(kp_c, road_c, xy)]
    pts = extract_graph_points(kp, road, c)
    edges_r, sums_r, cnts_r = oscene.infer_pass2(oracle, feats, pts, infos, AttrDict(cfg))
    uk, sums, cnts, _ = edge_votes(net, emb, pts, infos, 0, len(infos), c, torch.device("cuda"))
    n = pts.shape[0]
    hip = {(int(k // n), int(k % n)): s / m for k, s, m in zip(uk, sums, cnts)}
    d = np.array([hip[e] - sums_r[e] / cnts_r[e] for e in sums_r])
    worst = sorted(sums_r, key=lambda e: -abs(hip[e] - sums_r[e] / cnts_r[e]))[:3]
    print(f"seed {seed}: {len(d)} edges  |diff| max {np.abs(d).max():.4f}  p99 {np.percentile(np.abs(d), 99):.4f}  mean {np.abs(d).mean():.5f}  "
          + "  ".join(f"{e}: hip {hip[e]:.4f} ref {sums_r[e] / cnts_r[e]:.4f}" for e in worst), flush=True)

#!/usr/bin/env python
"""Host stages of one scene (mask -> points, pass-2 queries, collate, votes, accumulation, edge list) timed on the saved inputs of
tools/dump_scene.py — no GPU needed.  usage: python tools/prof_host_stages.py [gpurun_out/scene_dump.npz]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_road_amd import Config
from sam_road_amd import inferencer as I
from sam_road_amd.graph_points import extract_graph_points
from sam_road_amd.tiling import get_patch_info_one_img

_here = os.path.dirname(os.path.abspath(__file__))
d = np.load(sys.argv[1] if len(sys.argv) > 1 else (os.path.join(_here, "_scene_dump.npz") if os.path.exists(os.path.join(_here, "_scene_dump.npz"))
                                                    else "gpurun_out/scene_dump.npz"))
cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", DATASET="cityscale",
             INFER_BATCH_SIZE=64, SAMPLE_MARGIN=64, INFER_PATCHES_PER_EDGE=16, ITSC_THRESHOLD=0.248, ROAD_THRESHOLD=0.364, TOPO_THRESHOLD=0.499,
             ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
infos = get_patch_info_one_img(0, 2048, 64, 512, 16)
kp, road = d["kp"], d["road"]
N = int(os.environ.get("N", 10))
COLD = os.environ.get("COLD") == "1"          # between calls: sweep 256 MB and sleep, as a scene's 70 ms of pass 1 does to the caches
_junk = np.zeros(32 << 20, np.float64) if COLD else None
def T(name, f):
    f(); ts = []
    for _ in range(N):
        if COLD:
            _junk.__iadd__(1.0); time.sleep(0.05)
        t = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"{name:34s} median {np.median(ts):7.2f} ms   min {min(ts):7.2f}")
    return r
gp = T("extract_graph_points", lambda: extract_graph_points(kp, road, cfg))
assert np.array_equal(gp, d["graph_points"])
fq = T("build_all_patch_queries", lambda: I.build_all_patch_queries(gp, infos, 0, 256, cfg, flat=True))
assert np.array_equal(fq.knn, d["knn"]) and np.array_equal(fq.ids, d["ids"])
plan, pts_h, pairs_h, valid_h = T("pack_pass2_batches", lambda: I._pack_pass2_batches(fq, 0, 256, 64, 16))
batches = [(int(off), int(end), d[f"scores{i}"]) for i, (off, end, n_max, base) in enumerate(d["plan"])]
k, s = T("votes_from_scores", lambda: I._votes_from_scores(fq, 0, batches, gp.shape[0], 16))
uk, sums, cnts, first = T("accumulate_votes", lambda: I._accumulate_votes(k, s))
e = T("votes_to_edges", lambda: I.votes_to_edges(uk, sums, cnts, first, gp.shape[0], cfg.TOPO_THRESHOLD))
assert np.array_equal(e, d["edges"]), "edge list differs from the GPU run's"
print("votes", k.shape[0], "unique", uk.shape[0], "edges", e.shape[0])
r2 = T("vote_sums (fused)", lambda: I._vote_sums(fq, 0, batches, gp.shape[0], 16))
for a, b, n in zip(r2, (uk, sums, cnts, first), ("keys", "sums", "counts", "first")):
    assert a.shape == b.shape and np.array_equal(a, b), n
print("fused vote sums == votes + accumulate, bit for bit")
from sam_road_amd import graph_points as G
for name, m, thr in (("kp", kp, cfg.ITSC_THRESHOLD * 255), ("road", road, cfg.ROAD_THRESHOLD * 255)):
    c, sc = T(f"  {name}: mask candidates", lambda: G.points_and_scores_from_mask(m, thr))
    o = T(f"  {name}: argsort u8 ({len(sc)})", lambda: np.argsort(sc)[::-1])
    T(f"  {name}: gather points[order]", lambda: c[o, :])
    if name == "kp": k0 = c[o, :]
    else: k1 = c[o, :]
cand = np.concatenate([k0, k1], 0); prio = np.concatenate([np.ones(len(k0)), np.zeros(len(k1))])
o = T("  final: argsort prio", lambda: np.argsort(prio)[::-1])
T("  final: concat + gather", lambda: np.concatenate([k0, k1], 0)[o, :])
T("  final: nms_points total", lambda: G.nms_points(cand, prio, 16))
from sam_road_amd.hostcpu import usable_cpus
print("usable cpus", usable_cpus(), "worker_threads", I.worker_threads())
_wt, _ft = I.worker_threads, I.fill_threads
for nt in (1, 2, 4, 8, 16):
    I.worker_threads = I.fill_threads = lambda: nt
    T(f"vote_sums threads={nt}", lambda: I._vote_sums(fq, 0, batches, gp.shape[0], 16))
    T(f"build_all_patch_queries threads={nt}", lambda: I.build_all_patch_queries(gp, infos, 0, 256, cfg, flat=True))
I.worker_threads, I.fill_threads = _wt, _ft

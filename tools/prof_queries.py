"""Host-only timing of the pass-2 query builder (sam_road_amd.inferencer.build_all_patch_queries) on a synthetic CityScale-
sized point set (4400 NMS-spaced integer points, 256 tiles), with the library call timed separately."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sam_road_amd import Config, _lib
import sam_road_amd.inferencer as inf
from sam_road_amd.tiling import get_patch_info_one_img
from sam_road_amd.graph_points import nms_points
rng = np.random.default_rng(0)
cand = rng.integers(64, 1984, size=(60000, 2))
pts = nms_points(cand, np.zeros(len(cand)), 16)[:4400].astype(np.int64)
cfg = Config(NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
infos = get_patch_info_one_img(0, 2048, 64, 512, 16)
for _ in range(2): inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg)
t = time.perf_counter()
for _ in range(5): inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg)
print("build_all_patch_queries: %.1f ms  (cpu_count %d)" % ((time.perf_counter() - t) / 5 * 1e3, os.cpu_count()))
lib = _lib.load()
boxes = np.ascontiguousarray([[*i[1], *i[2]] for i in infos], dtype=np.int32)
counts = np.zeros(len(infos), dtype=np.int64)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
lib.srh_pass2_count(vp(pts), len(pts), vp(boxes), len(infos), vp(counts))
off = np.zeros(len(infos) + 1, dtype=np.int64); np.cumsum(counts, out=off[1:]); tot = int(off[-1])
ids = np.zeros(tot, np.int64); knn = np.zeros((tot, 16), np.int32); amb = np.zeros(tot, np.uint8)
for nt in (1, 4, 16):
    t = time.perf_counter()
    for _ in range(5): lib.srh_pass2_fill(vp(pts), len(pts), vp(boxes), len(infos), 16, 64, vp(off), vp(ids), vp(knn), vp(amb), nt)
    print("srh_pass2_fill %2d threads: %.1f ms   rows %d ambiguous %d" % (nt, (time.perf_counter() - t) / 5 * 1e3, tot, int(amb.sum())))

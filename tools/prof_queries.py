#!/usr/bin/env python
"""Where build_all_patch_queries' time goes on the box it runs on: srh_pass2_count / srh_pass2_fill separately, by thread count
(CityScale-like: 4400 NMS-spaced points, 256 tiles).  python tools/prof_queries.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_road_amd import _lib
from sam_road_amd.hostcpu import usable_cpus, worker_threads
from sam_road_amd.tiling import get_patch_info_one_img

rng = np.random.default_rng(0)
cand = rng.integers(64, 1984, size=(60000, 2)); keep = []; grid = {}
for p in cand:
    c = (p[0] // 16, p[1] // 16); ok = True
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for q in grid.get((c[0] + dx, c[1] + dy), []):
                if (q[0] - p[0]) ** 2 + (q[1] - p[1]) ** 2 < 256: ok = False
    if ok: grid.setdefault(c, []).append(p); keep.append(p)
    if len(keep) >= 4400: break
pts = np.ascontiguousarray(np.array(keep), dtype=np.int64)
infos = get_patch_info_one_img(0, 2048, 64, 512, 16)
boxes = np.ascontiguousarray([[*i[1], *i[2]] for i in infos], dtype=np.int32)
lib = _lib.load(); vp = lambda a: a.ctypes.data_as(C.c_void_p)
counts = np.zeros(256, np.int64)
def T(f, n=10):
    f(); t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e3
print("usable cpus", usable_cpus(), "worker_threads", worker_threads())
print("srh_pass2_count ms", T(lambda: lib.srh_pass2_count(vp(pts), len(pts), vp(boxes), 256, vp(counts))))
offsets = np.zeros(257, np.int64); np.cumsum(counts, out=offsets[1:]); total = int(offsets[-1])
ids = np.zeros(total, np.int64); knn = np.zeros((total, 16), np.int32); amb = np.zeros(total, np.uint8)
for nt in (1, 2, 4, 8, 16):
    print("srh_pass2_fill threads", nt, "ms", T(lambda: lib.srh_pass2_fill(vp(pts), len(pts), vp(boxes), 256, 16, 64, vp(offsets), vp(ids), vp(knn), vp(amb), None, nt)))
print("np.zeros of the outputs ms", T(lambda: (np.zeros(total, np.int64), np.zeros((total, 16), np.int32), np.zeros(total, np.uint8))))

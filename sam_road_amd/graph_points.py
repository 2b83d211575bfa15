"""Mask -> graph points (host side, between the two GPU passes): threshold + greedy radius NMS.

Mirrors reference graph_extraction.py:24-28,130-139 and graph_utils.py:572-591.  This step is a
"next" row of SURVEY.md §8(f) (rank 2): it stays on the host as in the reference; the greedy NMS is
order-dependent, so the candidate order (descending score, np.argsort tie order) is kept identical.
"""
import ctypes as C

import numpy as np

from . import _lib


def points_and_scores_from_mask(mask, threshold, n_threads=None):
    """graph_extraction.py:24-28: (x, y) of the pixels above the threshold in np.where order, and their scores.  u8 masks
    (the fused scene masks) are scanned by the library's host code (srh_mask_candidates: bands of rows on worker threads, 64-byte
    chunks tested with a byte-max instead of numpy's bool image + nonzero + gather, 14 -> 0.4 ms per 2048^2 mask); anything else
    takes the reference's numpy path."""
    if mask.dtype == np.uint8 and mask.ndim == 2 and mask.flags.c_contiguous:
        from .hostcpu import worker_threads
        lib = _lib.load()
        n = C.c_int64(0)
        H, W = mask.shape
        mp = mask.ctypes.data_as(C.c_void_p)
        nt = worker_threads() if n_threads is None else int(n_threads)
        cap = max(1024, (H * W) // 16)                      # one call unless more than 1/16 of the pixels pass
        while True:
            xy = np.empty((cap, 2), dtype=np.int64)
            sc = np.empty(cap, dtype=np.uint8)
            rc = lib.srh_mask_candidates(mp, H, W, float(threshold), xy.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                         cap, C.byref(n), nt)
            if rc == 0:
                return xy[:n.value], sc[:n.value]
            if n.value <= cap:
                raise _lib.SrhError("srh_mask_candidates failed")
            cap = n.value
    sel = mask > threshold
    rc = np.column_stack(np.where(sel))
    return rc[:, ::-1], mask[sel]          # (x, y), scores


def nms_points(points, scores, radius, return_indices=False):
    """Greedy radius suppression in descending score order; a score > 1.0 is always kept.
    The candidate order is numpy's (np.argsort tie order, as in the reference); the suppression loop itself — a
    Python loop over a KDTree in the reference, ~1 s per CityScale scene — runs in the library's host code
    (srh_nms_points_host, csrc/host_geom.hip: same algorithm on a uniform grid, exact integer distances)."""
    order = np.argsort(scores)[::-1]
    pts, sc = points[order, :], scores[order]
    n = int(order.shape[0])
    kept = np.ones(n, dtype=np.uint8)
    if n:
        if not np.issubdtype(pts.dtype, np.integer) or float(radius) != int(radius):
            raise TypeError("nms_points: integer pixel coordinates and an integer radius are expected "
                            "(np.where of the u8 masks, *_NMS_RADIUS of the YAMLs)")
        xy = np.ascontiguousarray(pts, dtype=np.int32)
        force = np.ascontiguousarray(sc > 1.0, dtype=np.uint8)
        if force.all():
            # every candidate is force-kept (graph_utils.py:586 `kept[nbr] = sc[nbr] > 1.0`): with the u8 mask scores of the first
            # two calls of extract_graph_points that is ALWAYS the case — the reference's loop then suppresses nothing and the call
            # only re-orders the candidates.  No neighbour search (it was 3/4 of mask -> points).
            return (pts, order) if return_indices else pts
        rc = _lib.load().srh_nms_points_host(xy.ctypes.data_as(C.c_void_p), force.ctypes.data_as(C.c_void_p), n,
                                             int(radius), kept.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise _lib.SrhError(f"srh_nms_points_host failed ({rc})")
    kept = kept.astype(bool)
    if return_indices:
        return pts[kept], order[kept]
    return pts[kept]


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(1, thread_name_prefix="srh-points")
    return _POOL


def _mask_points(mask, threshold, radius):
    cand, sc = points_and_scores_from_mask(mask, threshold)
    return nms_points(cand, sc, radius)


def _mask_candidates_ordered(mask, threshold, n_threads):
    """(candidates, their visiting order) of one mask when nms_points would keep all of them: np.argsort(scores)[::-1] is
    numpy's (the reference's tie order), the gather is left to srh_nms_merge_points."""
    cand, sc = points_and_scores_from_mask(mask, threshold, n_threads)
    return cand, np.ascontiguousarray(np.argsort(sc)[::-1], dtype=np.int64)


def extract_graph_points(keypoint_mask, road_mask, config):
    """graph_extraction.py:130-139.  The two masks are independent until the final merge: the road mask is processed on a worker
    thread while this thread does the keypoint mask (the library calls and numpy's argsort release the GIL).
    With u8 masks, thresholds >= 1 and integer radii (every shipped config) each mask's own nms_points call keeps all its
    candidates — their scores exceed 1.0, graph_utils.py:586 — and only orders them; then the whole step is: two threaded mask
    scans, numpy's two argsorts (kept for their tie order), numpy's argsort of the priorities, and ONE library call that gathers
    the candidates in visiting order, suppresses and compacts (srh_nms_merge_points).  Everything else takes the general path."""
    u8 = all(m.dtype == np.uint8 and m.ndim == 2 and m.flags.c_contiguous for m in (keypoint_mask, road_mask))
    thr_k, thr_r = config.ITSC_THRESHOLD * 255, config.ROAD_THRESHOLD * 255
    if u8 and thr_k >= 1.0 and thr_r >= 1.0 and float(config.ROAD_NMS_RADIUS) == int(config.ROAD_NMS_RADIUS) \
            and float(config.ITSC_NMS_RADIUS) == int(config.ITSC_NMS_RADIUS):
        import os
        import time
        from .hostcpu import worker_threads
        prof = os.environ.get("SRH_PROFILE_HOST") == "1"
        t_sec = [time.perf_counter()]
        def lap(name):
            if prof:
                t_sec.append(time.perf_counter())
                print(f"[points] {name}: {(t_sec[-1] - t_sec[-2]) * 1e3:.2f} ms", flush=True)
        nt = max(1, worker_threads() // 2)
        fut = _pool().submit(_mask_candidates_ordered, road_mask, thr_r, nt)
        xy_a, ord_a = _mask_candidates_ordered(keypoint_mask, thr_k, nt)
        na = xy_a.shape[0]
        lap("keypoint mask: scan + argsort")
        xy_b, ord_b = fut.result()
        nb = xy_b.shape[0]
        lap("wait for the road mask's")
        prio = np.concatenate([np.ones(na), np.zeros(nb)], axis=0)       # intersections first
        order = np.ascontiguousarray(np.argsort(prio)[::-1], dtype=np.int64)
        lap("argsort of the priorities")
        out = np.empty((na + nb, 2), dtype=np.int64)
        n = C.c_int64(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        xy_a, xy_b = np.ascontiguousarray(xy_a), np.ascontiguousarray(xy_b)
        rc = _lib.load().srh_nms_merge_points(vp(xy_a), vp(ord_a), na, vp(xy_b), vp(ord_b), nb, vp(order), int(config.ROAD_NMS_RADIUS),
                                              vp(out), C.byref(n))
        if rc != 0:
            raise _lib.SrhError(f"srh_nms_merge_points failed ({rc})")
        lap("gather + suppress + compact (library)")
        return out[:n.value].copy()
    fut = _pool().submit(_mask_points, road_mask, thr_r, config.ROAD_NMS_RADIUS)
    kp0 = _mask_points(keypoint_mask, thr_k, config.ITSC_NMS_RADIUS)
    kp1 = fut.result()
    cand = np.concatenate([kp0, kp1], axis=0)
    prio = np.concatenate([np.ones(kp0.shape[0]), np.zeros(kp1.shape[0])], axis=0)  # intersections first
    return nms_points(cand, prio, config.ROAD_NMS_RADIUS)

"""Scene-level tiled inference (reference inferencer.py:61-234) on the HIP path.

    infer_one_img(net, img, config) -> (pred_nodes[N,2] (row, col), pred_edges[E,2],
                                        keypoint_mask u8[H,W], road_mask u8[H,W])

Pass 1 (tile batcher, model, mask fusion, normalise) runs entirely on the GPU behind
SAMRoad.scene_pass1 / scene_normalise: the u8 scene is uploaded ONCE (the reference ships 4x the
bytes as f32 tiles, inferencer.py:56,94) and tiles are cropped on the device.  The step between the
passes (mask -> points) and the pass-2 query builder / edge vote stay on the host as in the
reference (SURVEY.md §8f "next" rows), with the python triple loop replaced by numpy.

With torch.distributed initialised (one process per GPU) the tile list is split into contiguous
chunks per rank; canvases are summed on rank 0, points broadcast, edge votes gathered
(sam_road_amd/distributed.py).  Only rank 0 returns the graph; other ranks return None.
"""
import numpy as np
import scipy.spatial
import torch

from . import distributed as D
from .graph_points import extract_graph_points
from .hostcpu import fill_threads, usable_cpus, worker_threads
from .tiling import get_patch_info_one_img, shard_tiles


def build_patch_queries(graph_points, x0, y0, x1, y1, config):
    """inferencer.py:148-176 for one tile: closed-box point query (rtree.intersection semantics for
    degenerate boxes), kNN(k+1) within NEIGHBOR_RADIUS, self removed, missing neighbour -> source."""
    gx, gy = graph_points[:, 0], graph_points[:, 1]
    ids = np.nonzero((gx >= x0) & (gx <= x1) & (gy >= y0) & (gy <= y1))[0]
    n, k = len(ids), int(config.MAX_NEIGHBOR_QUERIES)
    pts = graph_points[ids, :] - np.array([[x0, y0]], dtype=graph_points.dtype)
    if n == 0:
        return ids, pts.reshape(0, 2), np.zeros((0, k, 2), np.int64), np.zeros((0, k), bool)
    tree = scipy.spatial.KDTree(pts)          # the reference's class (inferencer.py:156): leafsize 10, which decides ties
    _, knn = tree.query(pts, k=k + 1, distance_upper_bound=config.NEIGHBOR_RADIUS)
    knn = knn[:, 1:]
    src = np.tile(np.arange(n)[:, None], (1, k))
    valid = knn < n
    tgt = np.where(valid, knn, src)
    return ids, pts, np.stack([src, tgt], -1), valid


class _FlatQueries:
    """Pass-2 queries of tiles [lo, hi) as flat arrays (the library's layout): offsets [n_tiles+1] rows per tile, ids [total]
    global point index of every row, local [total,2] tile-local (x, y), knn [total,K] int32 tile-local target or -1."""

    def __init__(self, offsets, ids, local, knn, tied=None):
        self.offsets, self.ids, self.local, self.knn = offsets, ids, local, knn
        self.tied = tied          # u8 [total]: rows decided by the kd-tree restatement (tie at the cut-off / coincident point)
        self.n_tiles = offsets.shape[0] - 1

    def tile(self, t):
        """(ids, points, pairs[n,K,2], valid[n,K]) of tile t in the reference's per-tile form (inferencer.py:148-176)."""
        a, b = int(self.offsets[t]), int(self.offsets[t + 1])
        knn = self.knn[a:b]
        valid = knn >= 0
        src = np.arange(b - a, dtype=np.int64)[:, None]
        pairs = np.stack([np.broadcast_to(src, knn.shape), np.where(valid, knn, src)], axis=-1)
        return self.ids[a:b], self.local[a:b], pairs, valid


_KDTREE_OK = [None]      # None: not checked yet; True / False: the library's kd-tree restatement agrees with the installed scipy


def _kdtree_selfcheck():
    """csrc/kdtree_emul.hpp restates how scipy's kd-tree (validated against scipy 1.15) breaks ties at the k-th neighbour.  Another
    scipy build could break them differently with no signal, so the first call compares the two on a small lattice where almost
    every row is tied (one tile, 13 x 11 points 8 px apart plus duplicates); on a mismatch the per-tile scipy path answers from
    then on — the reference's own call, slower — and a warning names the installed version."""
    if _KDTREE_OK[0] is not None:
        return _KDTREE_OK[0]
    _KDTREE_OK[0] = True                                        # re-entrancy guard: the check itself goes through the library path
    xs, ys = np.meshgrid(np.arange(13) * 8 + 3, np.arange(11) * 8 + 5)
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int64)
    pts = np.concatenate([pts, pts[[5, 5, 40, 77]]], 0)         # coincident points as well
    cfg = type("C", (), dict(MAX_NEIGHBOR_QUERIES=16, NEIGHBOR_RADIUS=64))()
    infos = [(0, (0, 0), (127, 127))]
    try:
        want = build_patch_queries(pts, 0, 0, 127, 127, cfg)
        got = build_all_patch_queries(pts, infos, 0, 1, cfg)[0]
        # the SET of neighbours of every source point must be scipy's (which points fall on the kept side of a tie); the order inside a
        # group of equidistant neighbours is heap-internal in scipy and (distance, index) in the library — the edge vote does not see it
        nbr = lambda q: np.sort(np.where(q[3], q[2][..., 1], -1), axis=1)
        ok = np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]) and np.array_equal(nbr(want), nbr(got))
    except Exception:
        ok = False
    if not ok:
        import warnings
        warnings.warn(f"the kd-tree tie-breaking restated in libsamroad_hip (validated against scipy 1.15) differs from the installed scipy "
                      f"{scipy.__version__}: pass-2 queries fall back to the per-tile scipy path", RuntimeWarning)
    _KDTREE_OK[0] = ok
    return ok


def build_all_patch_queries(graph_points, infos, lo, hi, config, flat=False):
    """build_patch_queries for tiles [lo, hi) in ONE call into the library's host code (srh_pass2_count / srh_pass2_fill,
    csrc/host_geom.hip: closed-box filter + exact integer kNN per tile, worker threads).  Source points whose scipy result is
    not determined by distances alone (tie at the k-th neighbour, coincident points) are answered by the library's restatement
    of scipy's kd-tree on their tile's points (ids ascending; csrc/kdtree_emul.hpp), so those rows equal the reference's call
    element for element; elsewhere the order inside a group of equidistant neighbours is (distance, index) where scipy's is
    heap-internal.  Returns a list of per-tile tuples, or the flat form."""
    import ctypes as C
    import os
    from . import _lib
    k, r = int(config.MAX_NEIGHBOR_QUERIES), config.NEIGHBOR_RADIUS
    n_tiles = hi - lo
    if n_tiles <= 0:
        return None if flat else []
    if float(r) != int(r) or not np.issubdtype(graph_points.dtype, np.integer) or not _kdtree_selfcheck():
        if flat:
            return None
        return [build_patch_queries(graph_points, *infos[t][1], *infos[t][2], config) for t in range(lo, hi)]
    import time
    prof = os.environ.get("SRH_PROFILE_HOST") == "1"
    t_sec = [time.perf_counter()]
    def lap(name):
        if prof:
            t_sec.append(time.perf_counter())
            print(f"[queries] {name}: {(t_sec[-1] - t_sec[-2]) * 1e3:.1f} ms", flush=True)
    lib = _lib.load()
    pts = np.ascontiguousarray(graph_points, dtype=np.int64)
    boxes = np.ascontiguousarray([[*infos[t][1], *infos[t][2]] for t in range(lo, hi)], dtype=np.int32)
    counts = np.zeros(n_tiles, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if lib.srh_pass2_count(vp(pts), pts.shape[0], vp(boxes), n_tiles, vp(counts)) != 0:
        raise _lib.SrhError("srh_pass2_count failed")
    offsets = np.zeros(n_tiles + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    total = int(offsets[-1])
    ids = np.empty(total, dtype=np.int64)                     # srh_pass2_fill writes every element of the four arrays
    knn = np.empty((total, k), dtype=np.int32)
    amb = np.empty(total, dtype=np.uint8)
    local = np.empty((total, 2), dtype=np.int64)
    if lib.srh_pass2_fill(vp(pts), pts.shape[0], vp(boxes), n_tiles, k, int(r), vp(offsets), vp(ids), vp(knn), vp(amb), vp(local),
                          fill_threads()) != 0:
        raise _lib.SrhError("srh_pass2_fill failed")
    lap("count + fill (library)")
    # (source points whose answer is not determined by distances alone — a tie at the K-th neighbour, a coincident point — were
    # decided inside the library the way the reference's scipy kd-tree decides them, csrc/kdtree_emul.hpp; round 2 re-queried
    # scipy per tile here: 11.5 ms per CityScale scene)
    fq = _FlatQueries(offsets, ids, local, knn, amb)
    if flat:
        return fq
    return [fq.tile(t) for t in range(n_tiles)]


def _collate(xs):
    """Zero-pad along axis 0 to the longest item and stack (graph_collate_fn-style padding, inferencer.py:179-185)."""
    length = max(x.shape[0] for x in xs)
    out = np.zeros((len(xs), length) + xs[0].shape[1:], dtype=xs[0].dtype)
    for i, x in enumerate(xs):
        out[i, :x.shape[0]] = x
    return out


PASS2_SORT_TILES = True       # False: batches of consecutive tiles, as the reference forms them (tools / tests set it; no environment switch)


def _sort_pass2_tiles():
    return PASS2_SORT_TILES


def _pass2_plan(fq, bs, sort_tiles=None):
    """Which tiles share a TopoNet batch: [(tiles int64 [nb] — indices into fq —, n_max, base_row)]; a batch is padded to its longest
    tile (graph_collate_fn-style, inferencer.py:179-185) and owns rows [base, base + nb * n_max) of the staging buffers.
    The reference batches consecutive tiles, so one dense tile makes 63 others carry its padding: a CityScale-like scene has 48 k
    query rows and 121 k padded ones.  TopoNet treats every row on its own (the sampler reads the row's tile, the transformer runs
    over the row's K neighbours), so WHICH tiles share a launch cannot change a score: tiles are grouped by ascending row count
    instead — 68 k padded rows, a little over half the device time and half the upload / download bytes of pass 2 — and the votes
    are still read in tile order (the reference's visiting order).  Empty tiles join no batch."""
    counts = np.diff(fq.offsets)
    if _sort_pass2_tiles() if sort_tiles is None else sort_tiles:
        idx = np.flatnonzero(counts > 0)
        idx = idx[np.argsort(counts[idx], kind="stable")]
        groups = [idx[i:i + bs] for i in range(0, len(idx), bs)]
    else:
        groups = [np.arange(i, min(i + bs, fq.n_tiles)) for i in range(0, fq.n_tiles, bs)]
    plan, rows_total = [], 0
    for tiles in groups:
        n_max = int(counts[tiles].max()) if len(tiles) else 0
        if n_max:
            plan.append((tiles.astype(np.int64), n_max, rows_total))
            rows_total += len(tiles) * n_max
    return plan, rows_total


def _contiguous(tiles):
    return len(tiles) > 0 and int(tiles[-1]) - int(tiles[0]) == len(tiles) - 1 and bool((np.diff(tiles) == 1).all())


def _pack_pass2_batches(fq, lo, hi, bs, K, alloc=None, sort_tiles=None):
    """Padded collate (inferencer.py:179-185) of every batch of _pass2_plan into ONE host buffer per kind: returns
    (plan, points f32 [rows,2], pairs i32 [rows,K,2], valid u8 [rows,K]).  Indices travel as int32 and the integer pixel coordinates
    as float32 (exact; srh_toponet accepts both, model.py:47's division promotes anyway).  `alloc(name, shape, dtype)` supplies the
    arrays (page-locked ones in the pipelined path).  (lo, hi: the tile range fq was built for; fq indexes tiles from 0.)"""
    if alloc is None:
        alloc = lambda name, shape, dtype: np.zeros(shape, dtype)
    plan, rows_total = _pass2_plan(fq, bs, sort_tiles)
    pts_h = alloc("points", (max(rows_total, 1), 2), np.float32)
    pairs_h = alloc("pairs", (max(rows_total, 1), K, 2), np.int32)
    valid_h = alloc("valid", (max(rows_total, 1), K), np.uint8)
    from . import _lib
    lib = _lib.load()
    local = np.ascontiguousarray(fq.local, dtype=np.int64)
    offsets = np.ascontiguousarray(fq.offsets, dtype=np.int64)
    knn = np.ascontiguousarray(fq.knn, dtype=np.int32)
    a_off, a_loc, a_knn = offsets.ctypes.data, local.ctypes.data, knn.ctypes.data
    a_pts, a_pairs, a_valid = pts_h.ctypes.data, pairs_h.ctypes.data, valid_h.ctypes.data
    for tiles, n_max, base in plan:
        # library host code (srh_pass2_pack): the numpy scatter of ~90k rows x 16 slots took 4 ms per CityScale scene.  One call per
        # run of consecutive tiles (raw addresses: a ctypes cast per argument would cost more than a tile's packing)
        runs = [(0, len(tiles))] if _contiguous(tiles) else [(j, j + 1) for j in range(len(tiles))]
        for j0, j1 in runs:
            row = base + j0 * n_max
            if lib.srh_pass2_pack(a_off + int(tiles[j0]) * 8, a_loc, a_knn, j1 - j0, n_max, K, a_pts + row * 8, a_pairs + row * K * 8,
                                  a_valid + row * K) != 0:
                raise _lib.SrhError("srh_pass2_pack failed")
    return plan, pts_h, pairs_h, valid_h


def _launch_pass2_batches(net, emb, plan, pts_d, pairs_d, valid_d, K, lo=0):
    """Sampler + TopoNet (inferencer.py:187-207) for every planned batch; nothing is fetched.  Returns [(tiles, scores)] with scores
    [nb, n_max, K] on the device (NaN -> -100 as the reference does before its range check); emb[t] = embeddings of fq's tile t."""
    out = []
    for tiles, n_max, base in plan:
        nb, sl = len(tiles), slice(base, base + len(tiles) * n_max)
        if _contiguous(tiles):
            e = emb[int(tiles[0]):int(tiles[-1]) + 1]
        else:
            # emb is an NCHW view of channels-last memory ([B,h,w,256] is what the library wrote and what srh_toponet reads): gather in
            # the STORAGE layout — index_select on the view would write an NCHW-contiguous copy that _topo then copies back
            e = emb.permute(0, 2, 3, 1).index_select(0, torch.as_tensor(tiles, device=emb.device)).permute(0, 3, 1, 2)
        scores = net.infer_toponet(e, pts_d[sl].view(nb, n_max, 2), pairs_d[sl].view(nb, n_max, K, 2), valid_d[sl].view(nb, n_max, K))
        out.append((tiles, torch.where(torch.isnan(scores), -100.0, scores).squeeze(-1)))
    return out


def _cfg_switch(v, default):
    """One parser for the boolean switches a YAML / override may spell several ways (PASS2_RAGGED, TILE_SHARD_PIPELINE): a missing key
    (the Config object's empty node) or None is the default; 'false' / 'no' / 'off' / '0' / '' are False; anything else bool(v)."""
    if v is None or (not isinstance(v, (bool, int, float, str)) and not v):     # absent key: addict-style empty node
        return default
    if isinstance(v, str):
        return v.strip().lower() not in ("false", "no", "off", "0", "")
    return bool(v)


def _poll_finite(net, device):
    """The library's non-finite sentinel (SAMRoad.check_finite, no synchronisation): an fp16 overflow in the encoder raises here instead
    of producing silently wrong masks.  The CPU stand-in models of the gloo tests have no such method."""
    chk = getattr(net, "check_finite", None)
    if chk is not None:
        chk(device, synchronize=False)


def _ragged_pass2(net, config):
    """Pass 2 without padding (srh_toponet_ragged: one launch for all query rows of the scene's tiles) unless the config switches it
    off (PASS2_RAGGED: False) or the model object has no such entry point (the CPU stand-in of the gloo tests)."""
    return _cfg_switch(config.PASS2_RAGGED, True) and hasattr(net, "infer_toponet_ragged")


def _pack_pass2_ragged(fq, K, alloc=None):
    """Unpadded collate of ALL query rows of fq's tiles (srh_pass2_pack_ragged): (points f32 [R,2], point_tile i32 [R], pairs i32
    [R,K,2], valid u8 [R,K]); rows keep their position in the flat query arrays, so tile t's scores are rows offsets[t] .. offsets[t+1]."""
    if alloc is None:
        alloc = lambda name, shape, dtype: np.zeros(shape, dtype)
    from . import _lib
    lib = _lib.load()
    offsets = np.ascontiguousarray(fq.offsets, dtype=np.int64)
    R = int(offsets[-1] - offsets[0])
    pts_h = alloc("points", (max(R, 1), 2), np.float32)
    tile_h = alloc("point_tile", (max(R, 1),), np.int32)
    pairs_h = alloc("pairs", (max(R, 1), K, 2), np.int32)
    valid_h = alloc("valid", (max(R, 1), K), np.uint8)
    local = np.ascontiguousarray(fq.local, dtype=np.int64)
    knn = np.ascontiguousarray(fq.knn, dtype=np.int32)
    if lib.srh_pass2_pack_ragged(offsets.ctypes.data, local.ctypes.data, knn.ctypes.data, fq.n_tiles, K, pts_h.ctypes.data,
                                 pairs_h.ctypes.data, valid_h.ctypes.data, tile_h.ctypes.data) != 0:
        raise _lib.SrhError("srh_pass2_pack_ragged failed")
    return R, pts_h, tile_h, pairs_h, valid_h


def _ragged_offsets(fq):
    """Row offsets of fq's tiles counted from 0 (int64 [n_tiles + 1]): what srh_toponet_ragged chunks the scene by."""
    return np.ascontiguousarray(np.asarray(fq.offsets, dtype=np.int64) - int(fq.offsets[0]))


def _ragged_batches(fq, scores_flat):
    """The per-tile view _tile_slots / _vote_sums take, of one flat score array [R, K]: tile t = rows offsets[t] .. offsets[t+1]."""
    off = np.asarray(fq.offsets, dtype=np.int64) - int(fq.offsets[0])
    sc = np.ascontiguousarray(scores_flat, dtype=np.float32)
    return [(np.array([t], dtype=np.int64), sc[off[t]:off[t + 1]][None]) for t in range(fq.n_tiles) if off[t + 1] > off[t]]


def _tile_slots(fq, lo, batches, K):
    """Per-tile view of the score batches: [(tile, address of its f32 [n_max, K] block, n_max)] in ascending tile order, plus the
    arrays that keep the memory alive.  batches = [(tiles, scores [nb,n_max,K])] or, for consecutive tiles, [(off, end, scores)]
    with absolute tile numbers (off - lo indexes fq)."""
    keep, slots = [], []
    for b in batches:
        if len(b) == 3:
            tiles, sc = np.arange(b[0] - lo, b[1] - lo, dtype=np.int64), b[2]
        else:
            tiles, sc = b
        sc = np.ascontiguousarray(sc, dtype=np.float32)
        if sc.ndim != 3 or sc.shape[0] != len(tiles) or sc.shape[2] != K:
            raise ValueError("score batch of the wrong shape")
        keep.append(sc)
        stride = sc.shape[1] * K * 4
        slots.extend((int(t), sc.ctypes.data + j * stride, sc.shape[1]) for j, t in enumerate(tiles))
    slots.sort(key=lambda x: x[0])
    return slots, keep


def _votes_from_scores(fq, lo, batches, n_pts, K):
    """inferencer.py:209-221's visiting order (tile, source point, neighbour slot) as flat (key, score) vote arrays; batches as in
    _tile_slots, on the host (srh_pass2_votes, csrc/host_geom.hip; it also enforces the reference's 0 <= score <= 1 assertion)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    slots, keep = _tile_slots(fq, lo, batches, K)
    cap = int((fq.knn >= 0).sum())
    k = np.empty(cap, np.int64)
    s = np.empty(cap, np.float64)
    cnt_c = C.c_int64(0)
    offsets = np.ascontiguousarray(fq.offsets, dtype=np.int64)
    a_off = offsets.ctypes.data
    # runs of consecutive tiles that sit one after the other in one score array go through one call
    i = 0
    while i < len(slots):
        t, addr, n_max = slots[i]
        j = i + 1
        while j < len(slots) and slots[j][0] == slots[j - 1][0] + 1 and slots[j][2] == n_max and slots[j][1] == slots[j - 1][1] + n_max * K * 4:
            j += 1
        rc = lib.srh_pass2_votes(addr, j - i, n_max, K, a_off + t * 8, vp(fq.ids), vp(fq.knn), n_pts, vp(k), vp(s), cap, C.byref(cnt_c))
        if rc != 0:
            raise AssertionError("edge score outside [0, 1] (reference inferencer.py:219) or inconsistent query arrays")
        i = j
    del keep
    return k[:cnt_c.value], s[:cnt_c.value]


def _vote_sums(fq, lo, batches, n_pts, K):
    """_votes_from_scores + _accumulate_votes without the ~700k intermediate votes of a CityScale scene: the library groups the
    query rows by source point and adds every point's votes into a table of its few dozen targets (srh_pass2_vote_sums,
    csrc/host_geom.hip; visiting order per key kept, so the float64 sums, counts and first-vote positions are the same, bit for
    bit — tests/test_host_logic.py).  batches as in _tile_slots (every tile is handed over as a "batch" of one: where its scores
    lie does not matter)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    slots, keep = _tile_slots(fq, lo, batches, K)
    nb = len(slots)
    ptrs = (C.c_void_p * max(nb, 1))(*[a for _, a, _ in slots])
    tile0 = np.array([t for t, _, _ in slots], dtype=np.int32)
    cnt = np.ones(nb, dtype=np.int32)
    n_max = np.array([m for _, _, m in slots], dtype=np.int64)
    cap = int(fq.knn.size)                     # >= the number of distinct edges; the pages beyond them are never touched
    uk, sums, cnts, first = np.empty(cap, np.int64), np.empty(cap, np.float64), np.empty(cap, np.float64), np.empty(cap, np.int64)
    nu = C.c_int64(0)
    rc = lib.srh_pass2_vote_sums(ptrs, vp(tile0), vp(cnt), vp(n_max), nb, K, vp(fq.offsets), fq.n_tiles, vp(fq.ids), vp(fq.knn),
                                 n_pts, vp(uk), vp(sums), vp(cnts), vp(first), cap, C.byref(nu), worker_threads())
    del keep
    if rc != 0:
        raise AssertionError("edge score outside [0, 1] (reference inferencer.py:219) or inconsistent query arrays")
    return uk[:nu.value], sums[:nu.value], cnts[:nu.value], first[:nu.value]


def _accumulate_votes(k, s):
    """The reference's dict accumulation (float64 sums in visiting order) as a stable radix sort by key + one sequential pass
    in the library's host code (np.unique + np.bincount did the same in 11 ms per CityScale scene)."""
    import ctypes as C
    from . import _lib
    if k.shape[0] == 0:
        return np.zeros(0, np.int64), np.zeros(0), np.zeros(0), np.zeros(0, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    uk, sums, cnts, first = np.empty_like(k), np.empty_like(s), np.empty_like(s), np.empty_like(k)
    nu = C.c_int64(0)
    if _lib.load().srh_edge_vote_accumulate_mt(vp(k), vp(s), k.shape[0], vp(uk), vp(sums), vp(cnts), vp(first), C.byref(nu),
                                               worker_threads()) != 0:
        raise _lib.SrhError("srh_edge_vote_accumulate failed")
    return uk[:nu.value], sums[:nu.value], cnts[:nu.value], first[:nu.value]


def edge_votes(net, emb, graph_points, infos, lo, hi, config, device, raw=False):
    """Pass 2 over tiles [lo, hi) whose embeddings are emb[0 : hi-lo] (inferencer.py:135-221): returns the
    unique directed edge keys (src * n_points + tgt) with their score sums, counts and first-vote positions.  The sums are
    accumulated in float64 in the reference's order (tile, point, neighbour slot), so they are bit-identical to its dict loop
    for the tiles of THIS call (a multi-rank merge of such results: see distributed.gather_edge_votes).  raw=True returns the
    votes themselves, (keys int64, scores float64) in visiting order, for an exact merge on one rank."""
    import os
    import time
    prof = os.environ.get("SRH_PROFILE_HOST") == "1"      # tuning aid: print the wall time of each section
    t_sec = [time.perf_counter()]
    def lap(name):
        if prof:
            t_sec.append(time.perf_counter())
            print(f"[edge_votes] {name}: {(t_sec[-1] - t_sec[-2]) * 1e3:.1f} ms", flush=True)
    bs = int(config.INFER_BATCH_SIZE)
    n_pts = graph_points.shape[0]
    K = int(config.MAX_NEIGHBOR_QUERIES)
    empty = (np.zeros(0, np.int64), np.zeros(0)) if raw else (np.zeros(0, np.int64), np.zeros(0), np.zeros(0), np.zeros(0, np.int64))
    fq = build_all_patch_queries(graph_points, infos, lo, hi, config, flat=True)
    lap("build_all_patch_queries")
    if fq is None:
        # non-integer coordinates / radius: the reference's per-tile scipy path, votes gathered in numpy
        if hi - lo <= 0:
            return empty
        all_q = [build_patch_queries(graph_points, *infos[t][1], *infos[t][2], config) for t in range(lo, hi)]
    # launch every batch before fetching any scores.  All batches' padded arrays are built into ONE host buffer per kind and
    # uploaded with one blocking copy each: per-batch non_blocking uploads of pageable numpy memory went through torch's
    # pinned-staging allocator and stalled 30-40 ms in some scenes (profiles/r02_scene_stages.txt).
    launched = []
    if fq is not None and _ragged_pass2(net, config):
        # every query row of this call's tiles in ONE unpadded launch (68 k padded rows -> 48 k real ones on a CityScale scene)
        R, pts_h, tile_h, pairs_h, valid_h = _pack_pass2_ragged(fq, K)
        if R == 0:
            return empty
        scores = net.infer_toponet_ragged(emb, *(torch.from_numpy(x[:R]).to(device) for x in (pts_h, tile_h, pairs_h, valid_h)),
                                          tile_offsets=_ragged_offsets(fq))
        scores = torch.where(torch.isnan(scores), -100.0, scores)
        lap("collate + H2D + launch (ragged)")
        host_scores = _ragged_batches(fq, scores.cpu().numpy())
        if not raw:
            out = _vote_sums(fq, lo, host_scores, n_pts, K)
            lap("score fetch + vote sums")
            return out
        return _votes_from_scores(fq, lo, host_scores, n_pts, K)          # raw: the votes themselves, in visiting order
    if fq is not None:
        plan, pts_h, pairs_h, valid_h = _pack_pass2_batches(fq, lo, hi, bs, K)
        pts_d, pairs_d, valid_d = (torch.from_numpy(x).to(device) for x in (pts_h, pairs_h, valid_h))
        launched = [(tiles, None, None, sc) for tiles, sc in _launch_pass2_batches(net, emb, plan, pts_d, pairs_d, valid_d, K)]
    else:
        for off in range(lo, hi, bs):
            end = min(off + bs, hi)
            qs = all_q[off - lo:end - lo]
            if max(q[1].shape[0] for q in qs) == 0:
                continue
            pts = _collate([q[1].astype(np.float32) for q in qs])
            pairs = _collate([q[2].astype(np.int32) for q in qs])
            valid = _collate([q[3] for q in qs])
            scores = net.infer_toponet(emb[off - lo:end - lo], torch.as_tensor(pts).to(device), torch.as_tensor(pairs).to(device),
                                       torch.as_tensor(valid).to(device))
            launched.append((off, end, qs, torch.where(torch.isnan(scores), -100.0, scores).squeeze(-1)))
    lap("collate + H2D + launch")
    if not launched:
        return empty
    if fq is not None:
        host_scores = [(tiles, sc.cpu().numpy()) for tiles, _, _, sc in launched]
        if not raw:
            out = _vote_sums(fq, lo, host_scores, n_pts, K)
            lap("score fetch + vote sums")
            return out
        k, s = _votes_from_scores(fq, lo, host_scores, n_pts, K)
    else:
        keys_l, score_l = [], []
        for off, end, qs, scores_dev in launched:
            scores = scores_dev.cpu().numpy()
            for b, (ids, _, prs, vld) in enumerate(qs):
                n = len(ids)
                if n == 0:
                    continue
                sc = scores[b, :n][vld]
                assert ((sc >= 0.0) & (sc <= 1.0)).all()
                keys_l.append(ids[prs[:, :, 0]][vld].astype(np.int64) * n_pts + ids[prs[:, :, 1]][vld].astype(np.int64))
                score_l.append(sc.astype(np.float64))
        if not keys_l:
            return empty
        k = np.ascontiguousarray(np.concatenate(keys_l), dtype=np.int64)
        s = np.ascontiguousarray(np.concatenate(score_l), dtype=np.float64)
    lap("score fetch + keys")
    if raw:
        return k, s
    if k.shape[0] == 0:
        return empty
    out = _accumulate_votes(k, s)
    lap("accumulate")
    return out


def votes_to_edges(uk, sums, cnts, first, n_pts, threshold):
    """inferencer.py:224-228: mean directed score > TOPO_THRESHOLD, as an [E,2] array in the INSERTION order of the
    reference's dict (= order of each key's first vote; `first` from edge_votes / gather_edge_votes).  That order follows the
    per-tile point order, which in the reference is whatever rtree.intersection yields; here tiles list their points by
    ascending global index (the edge SET does not depend on it — pinned by tests/test_refrun_golden.py).  Among EQUIDISTANT
    neighbours of one source point the slot order is scipy-heap-internal in the reference and distance-then-index here."""
    n = int(uk.shape[0])
    arrs = [np.ascontiguousarray(a, dtype=d) for a, d in ((uk, np.int64), (sums, np.float64), (cnts, np.float64), (first, np.int64))]
    if n == 0 or n_pts <= 0 or any(a.shape != (n,) for a in arrs):
        keep = (sums / np.maximum(cnts, 1.0)) > threshold
        k = uk[keep][np.argsort(first[keep], kind="stable")]
        return np.stack([k // n_pts, k % n_pts], axis=1).reshape(-1, 2)
    import ctypes as C
    from . import _lib
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    out = np.empty((n, 2), dtype=np.int64)
    ne = C.c_int64(0)
    # library host code (srh_votes_to_edges): the kept edges dropped into a table indexed by first-vote position, read back in order
    rc = _lib.load().srh_votes_to_edges(*(vp(a) for a in arrs), n, int(n_pts), float(threshold), vp(out), C.byref(ne))
    if rc != 0:
        raise _lib.SrhError(f"srh_votes_to_edges failed ({rc})")
    return out[:ne.value].copy()


def _numpy_hugepages(enabled):
    """numpy's switch for madvise(MADV_HUGEPAGE) on array allocations of 4 MiB and more (NUMPY_MADVISE_HUGEPAGE); returns the
    previous state, None if this numpy has no such switch."""
    try:
        from numpy._core.multiarray import _set_madvise_hugepage
    except ImportError:
        try:
            from numpy.core.multiarray import _set_madvise_hugepage
        except ImportError:
            return None
    return bool(_set_madvise_hugepage(bool(enabled)))


def _host_quiet():
    """Context manager around the host stages of a scene.  Two process-wide settings are switched off inside and restored after:
    * the cyclic garbage collector — cheap insurance against a generation-2 sweep landing inside a scene; reference counting still
      frees every array as it goes.  (It was introduced for +20-30 ms scenes that turned out to be the second item.)
    * numpy's transparent-huge-page madvise for large arrays.  Every first touch / split of such a 2 MiB mapping raises an MMU
      notifier, and the amdgpu driver answers by taking the process's GPU queues off the hardware for 20-30 ms: with the host
      stages running beside pass 1 (infer_imgs) ONE kernel per scene was frozen for that long during the first ~30 scenes of a
      process — pass 1 took 105 instead of 75 ms on the device — until the heap had warmed up (profiles/r02_scene_pipeline.txt,
      rocprofv3 kernel trace); with plain 4 KiB pages the stalls are gone from the first scene on."""
    import contextlib
    import gc

    @contextlib.contextmanager
    def cm():
        was_gc = gc.isenabled()
        gc.disable()
        was_huge = _numpy_hugepages(False)
        try:
            yield
        finally:
            if was_huge:
                _numpy_hugepages(True)
            if was_gc:
                gc.enable()
    return cm()


def infer_one_img(net, img, config, device=None):
    """reference inferencer.py:61-234: (pred_nodes (row, col), pred_edges, keypoint_mask u8, road_mask u8) of one scene (with the
    garbage collector and numpy's huge-page madvise paused for the duration of the call, see _host_quiet)."""
    with _host_quiet():
        return _infer_one_img(net, img, config, device)


def _scene_plan(img, config):
    """Validated scene + its tile list (inferencer.py:63-76): (img u8 [S,S,3], infos, tile origins int32 [n,2] (x0, y0))."""
    img = np.asarray(img)
    # the reference uses img.shape[0] for both axes (inferencer.py:63,67) and casts whatever it gets to f32; a non-square or
    # non-u8 scene would silently produce garbage here (row stride = S on the device), so it is refused instead
    if img.ndim != 3 or img.shape[2] != 3 or img.shape[0] != img.shape[1] or img.dtype != np.uint8:
        raise ValueError(f"infer_one_img expects a square HxWx3 uint8 scene, got {img.dtype} {tuple(img.shape)}")
    image_size = img.shape[0]
    if image_size < int(config.PATCH_SIZE) + 2 * int(config.SAMPLE_MARGIN or 0):
        raise ValueError(f"scene {image_size} px is smaller than PATCH_SIZE + 2 * SAMPLE_MARGIN")
    infos = get_patch_info_one_img(0, image_size, config.SAMPLE_MARGIN, config.PATCH_SIZE,
                                   config.INFER_PATCHES_PER_EDGE)
    all_xy = np.array([[p[1][0], p[1][1]] for p in infos], dtype=np.int32)
    assert all_xy.min() >= 0 and all_xy.max() + int(config.PATCH_SIZE) <= image_size
    return img, infos, all_xy


def _infer_one_img(net, img, config, device=None):
    device = torch.device(device) if device is not None else next(net.parameters()).device
    img, infos, all_xy = _scene_plan(img, config)
    image_size = img.shape[0]
    bs = int(config.INFER_BATCH_SIZE)
    world = torch.distributed.get_world_size() if D.is_distributed() else 1
    rank = torch.distributed.get_rank() if D.is_distributed() else 0
    lo, hi = shard_tiles(len(infos), world, rank)

    import os
    import time
    prof = os.environ.get("SRH_PROFILE_HOST") == "1"      # tuning aid: wall time of each stage (synchronises the device)
    t_sec = [time.perf_counter()]
    def lap(name):
        if prof:
            torch.cuda.synchronize(device)
            t_sec.append(time.perf_counter())
            print(f"[infer_one_img] {name}: {(t_sec[-1] - t_sec[-2]) * 1e3:.1f} ms", flush=True)

    # ---- pass 1 (GPU): crop -> encoder -> decoder -> fused canvases; embeddings stay resident
    scene = torch.as_tensor(np.ascontiguousarray(img), dtype=torch.uint8).to(device)
    xy_dev = torch.as_tensor(all_xy).to(device)
    lap("scene upload")
    kp_c, road_c, emb = net.scene_pass1(scene, xy_dev[lo:hi], bs)      # an empty shard (world > n_tiles) returns zero canvases
    lap("pass 1 (GPU)")
    D.reduce_canvases(kp_c, road_c, dst=0, bands=D.tile_bands(all_xy, int(config.PATCH_SIZE), world) if world > 1 else None)
    graph_points = None
    kp_mask = road_mask = None
    if rank == 0:
        kp_u8, road_u8 = net.scene_normalise(kp_c, road_c, xy_dev)
        kp_mask, road_mask = kp_u8.cpu().numpy(), road_u8.cpu().numpy()
        _poll_finite(net, device)                      # the masks are on the host, so every LayerNorm pass of pass 1 has reported
        lap("normalise + mask D2H")
        graph_points = extract_graph_points(kp_mask, road_mask, config)
        lap("extract_graph_points")
    graph_points = D.broadcast_points(graph_points, src=0, device=device if world > 1 else None)
    if graph_points.shape[0] == 0:
        if rank != 0:
            return None
        return graph_points, np.zeros((0, 2), dtype=np.int32), kp_mask, road_mask

    # ---- pass 2: per-tile queries (host) -> sampler + TopoNet (GPU) -> directed edge votes
    n_pts = graph_points.shape[0]
    if world > 1 and config.EXACT_VOTE_MERGE:
        # exact multi-rank merge (extension key, default off): every raw vote goes to rank 0 in the one-process visiting order
        k_raw, s_raw = edge_votes(net, emb, graph_points, infos, lo, hi, config, device, raw=True)
        k_raw, s_raw = D.gather_raw_votes(k_raw, s_raw, dst=0, device=device)
        if rank != 0:
            return None
        uk, sums, cnts, first = _accumulate_votes(k_raw, s_raw)
    else:
        uk, sums, cnts, first = edge_votes(net, emb, graph_points, infos, lo, hi, config, device)
        lap("edge_votes")
        uk, sums, cnts, first = D.gather_edge_votes(uk, sums, cnts, n_pts, dst=0, device=device if world > 1 else None,
                                                    first=first)
    if rank != 0:
        return None
    pred_edges = votes_to_edges(uk, sums, cnts, first, n_pts, config.TOPO_THRESHOLD)
    pred_nodes = graph_points[:, ::-1]  # (row, col)
    lap("threshold + edge list")
    return pred_nodes, pred_edges, kp_mask, road_mask


class _StagingPool:
    """Page-locked host staging buffers of one in-flight scene, grown on demand and reused (a hipHostMalloc costs
    milliseconds).  On a CPU device (the gloo / stand-in tests) plain tensors take their place."""

    def __init__(self, device):
        self._pin = device.type == "cuda"
        self._t = {}

    def get(self, name, shape, dtype):
        n = int(np.prod(shape))
        t = self._t.get(name)
        if t is None or t.dtype != dtype or t.numel() < n:
            t = torch.empty(max(n + n // 4, 1), dtype=dtype, pin_memory=self._pin)
            self._t[name] = t
        return t[:n].view(tuple(shape))


class _Lane:
    """Stream plumbing of infer_imgs: uploads go through page-locked staging on a side stream, the compute stream waits on an
    event, downloads are asynchronous copies into page-locked buffers followed by an event — no call blocks the host until the
    results are actually needed.  (A pageable-memory hipMemcpyAsync is stream-ordered AND host-blocking: issued behind a scene's
    pass 1 it would park the host for the whole pass.)  Degenerates to synchronous copies on a CPU device."""

    def __init__(self, device):
        self.device = device
        self.cuda = device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device) if self.cuda else None

    def upload(self, pool, name, arr):
        """numpy array -> device tensor, stream-ordered before everything launched on the compute stream afterwards."""
        t_dtype = torch.from_numpy(arr[:0].reshape(0)).dtype
        stage = pool.get("up_" + name, arr.shape, t_dtype)
        stage.numpy()[...] = arr
        return self.upload_staged(stage)

    def upload_staged(self, stage):
        if not self.cuda:
            return stage.clone()
        main = torch.cuda.current_stream(self.device)
        dst = torch.empty(stage.shape, dtype=stage.dtype, device=self.device)
        self.copy_stream.wait_stream(main)           # dst may be a recycled block still in use by queued compute work
        with torch.cuda.stream(self.copy_stream):
            dst.copy_(stage, non_blocking=True)
        dst.record_stream(self.copy_stream)
        main.wait_stream(self.copy_stream)
        return dst

    def download(self, pool, name, tensors):
        """Device tensors -> page-locked host tensors (asynchronous) + the event that says they have landed."""
        outs = []
        for i, t in enumerate(tensors):
            h = pool.get(f"down_{name}{i}", t.shape, t.dtype)
            h.copy_(t, non_blocking=self.cuda)
            outs.append(h)
        ev = None
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        return outs, ev


class _SceneJob:
    pass


def infer_imgs(net, imgs, config, device=None, tile_sharded=None, pipelined=None):
    """infer_one_img over a sequence of scenes, as a generator of the same tuples in the same order — software-pipelined on
    one GPU: while the device runs pass 1 of scene i+1, the host does scene i's mask -> points -> pass-2 queries; scene i's
    TopoNet batches are queued behind that pass 1 and its edge vote runs while scene i+2 is on the device.  One compute stream
    (the library context is single-stream), one copy stream, events instead of device-wide synchronisation; every scene's
    canvases / embeddings are its own tensors, so nothing of the context is double-buffered.  The results are those of
    infer_one_img bit for bit (same kernels in the same order per scene).  The reference's loop (inferencer.py:289-349) is
    strictly serial; this is what the CLI uses.  tile_sharded (default: torch.distributed is initialised) selects the other
    multi-GPU mode instead — every scene's tiles split over the ranks; with tile_sharded=False each rank runs its own scenes through
    the pipeline and no collective is issued.  The tile-sharded mode has two loops: the SERIAL one (default) — infer_one_img scene by
    scene, its three exchange steps strictly in sequence — and the PIPELINED one (_infer_imgs_tile_sharded; `pipelined=True`, config
    key TILE_SHARD_PIPELINE, CLI `--shard tiles-pipelined`), which interleaves the band reduce of scene i+1 with the point broadcast
    and vote gather of scene i.  The pipelined loop has only ever run on gloo / CPU (no multi-GPU box was available to the builder:
    DESIGN.md §6), so it stays opt-in until an RCCL run exists; both give the same results (tests/test_distributed_cpu.py)."""
    if D.is_distributed() if tile_sharded is None else tile_sharded:
        if pipelined is None:
            pipelined = _cfg_switch(config.TILE_SHARD_PIPELINE, False)
        if pipelined:
            yield from _infer_imgs_tile_sharded(net, imgs, config, device)
        else:
            for img in imgs:
                yield infer_one_img(net, img, config, device=device)
        return
    device = torch.device(device) if device is not None else next(net.parameters()).device
    lane = _Lane(device)
    pools = [_StagingPool(device), _StagingPool(device)]
    bs, K = int(config.INFER_BATCH_SIZE), int(config.MAX_NEIGHBOR_QUERIES)
    import os
    import time
    prof = os.environ.get("SRH_PROFILE_HOST") == "1"      # tuning aid: host wall time of each step (no device synchronisation)
    t_sec = [time.perf_counter()]
    def lap(name):
        if prof:
            t_sec.append(time.perf_counter())
            print(f"[infer_imgs] {name}: {(t_sec[-1] - t_sec[-2]) * 1e3:.1f} ms", flush=True)

    def launch_pass1(img, pool):                       # G1: upload, pass 1, normalise, masks on their way to the host
        job = _SceneJob()
        img, job.infos, all_xy = _scene_plan(img, config)
        job.pool, job.n_tiles = pool, len(job.infos)
        scene = lane.upload(pool, "scene", img)
        xy_dev = lane.upload(pool, "xy", all_xy)
        lap("stage + queue scene upload")
        if prof and lane.cuda:
            job.t = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            job.t[0].record()
        kp_c, road_c, job.emb = net.scene_pass1(scene, xy_dev, bs)
        lap("queue pass 1")
        kp_u8, road_u8 = net.scene_normalise(kp_c, road_c, xy_dev)
        if prof and lane.cuda:
            job.t[1].record()
        job.masks, job.e1 = lane.download(pool, "mask", [kp_u8, road_u8])
        lap("queue normalise + mask download")
        return job

    def points_and_pass2(job):                         # H1 + G2: points, queries, TopoNet launches, scores on their way back
        if job.e1 is not None:
            job.e1.synchronize()
        job.kp_mask, job.road_mask = (m.numpy().copy() for m in job.masks)
        job.graph_points = extract_graph_points(job.kp_mask, job.road_mask, config)
        lap("extract_graph_points")
        job.fq = job.plan = job.votes = None
        if job.graph_points.shape[0] == 0:
            return
        job.fq = build_all_patch_queries(job.graph_points, job.infos, 0, job.n_tiles, config, flat=True)
        lap("build_all_patch_queries")
        if job.fq is None:                             # non-integer radius: the serial per-tile path
            job.votes = edge_votes(net, job.emb, job.graph_points, job.infos, 0, job.n_tiles, config, device)
            return
        stage = {}
        def alloc(name, shape, dtype):
            stage[name] = job.pool.get("up_" + name, shape, torch.from_numpy(np.zeros(0, dtype)).dtype)
            return stage[name].numpy()                 # srh_pass2_pack writes every row, padding included
        if _ragged_pass2(net, config):                 # ONE unpadded launch for the scene's query rows (srh_toponet_ragged)
            R = _pack_pass2_ragged(job.fq, K, alloc)[0]
            job.plan = "ragged" if R else []
            if not R:
                return
            pts_d, tile_d, pairs_d, valid_d = (lane.upload_staged(stage[n][:R]) for n in ("points", "point_tile", "pairs", "valid"))
            if prof and lane.cuda:
                job.t[2].record()
            sc = net.infer_toponet_ragged(job.emb, pts_d, tile_d, pairs_d, valid_d, tile_offsets=_ragged_offsets(job.fq))
            sc = torch.where(torch.isnan(sc), -100.0, sc)
            if prof and lane.cuda:
                job.t[3].record()
            job.scores, job.e2 = lane.download(job.pool, "score", [sc])
            if prof and lane.cuda:
                job.t[4].record()
            job.emb = None
            lap("pack + queue pass 2 (ragged)")
            return
        job.plan = _pack_pass2_batches(job.fq, 0, job.n_tiles, bs, K, alloc)[0]
        if not job.plan:
            return
        pts_d, pairs_d, valid_d = (lane.upload_staged(stage[n]) for n in ("points", "pairs", "valid"))
        if prof and lane.cuda:
            job.t[2].record()
        launched = _launch_pass2_batches(net, job.emb, job.plan, pts_d, pairs_d, valid_d, K, 0)
        if prof and lane.cuda:
            job.t[3].record()
        job.scores, job.e2 = lane.download(job.pool, "score", [sc for _, sc in launched])
        if prof and lane.cuda:
            job.t[4].record()
        job.emb = None
        lap("pack + queue pass 2")

    def finish(job):                                   # H2: votes -> edges
        nodes = job.graph_points[:, ::-1]              # (row, col)
        no_edges = np.zeros((0, 2), dtype=np.int32)
        if job.graph_points.shape[0] == 0:
            return job.graph_points, no_edges, job.kp_mask, job.road_mask
        n_pts = job.graph_points.shape[0]
        if job.votes is None:
            if not job.plan:
                return nodes, no_edges.astype(np.int64), job.kp_mask, job.road_mask
            if job.e2 is not None:
                job.e2.synchronize()
            lap("wait for pass-2 scores")
            if prof and lane.cuda:
                t = job.t
                print(f"[infer_imgs] device: pass 1 {t[0].elapsed_time(t[1]):.1f} ms, mask download -> pass 2 start {t[1].elapsed_time(t[2]):.1f} ms, "
                      f"pass 2 {t[2].elapsed_time(t[3]):.1f} ms, score download {t[3].elapsed_time(t[4]):.1f} ms", flush=True)
            batches = _ragged_batches(job.fq, job.scores[0].numpy()) if job.plan == "ragged" else \
                [(tiles, sc.numpy()) for (tiles, _, _), sc in zip(job.plan, job.scores)]
            job.votes = _vote_sums(job.fq, 0, batches, n_pts, K)
        edges = votes_to_edges(*job.votes, n_pts, config.TOPO_THRESHOLD)
        lap("votes -> edges")
        return nodes, edges, job.kp_mask, job.road_mask

    it = iter(imgs)
    img = next(it, None)
    if img is None:
        return
    with _host_quiet():
        cur = launch_pass1(img, pools[0])
    prev, i = None, 0
    while cur is not None:
        with _host_quiet():
            lap("(consumer)")
            if cur.e1 is not None:
                cur.e1.synchronize()                   # scene i's masks are on the host: the device is free for scene i+1
                _poll_finite(net, device)
            lap("wait for pass-1 masks")
            img = next(it, None)
            nxt = launch_pass1(img, pools[(i + 1) % 2]) if img is not None else None
            res = finish(prev) if prev is not None else None
        if prev is not None:
            yield res
        with _host_quiet():
            points_and_pass2(cur)
        prev, cur, i = cur, nxt, i + 1
    with _host_quiet():
        res = finish(prev)
    yield res


def _infer_imgs_tile_sharded(net, imgs, config, device=None, stats=None):
    """Tile-sharded scenes (BASELINE configs[3]: ONE scene's tiles over the ranks of a node), software-pipelined like infer_imgs:
    every rank queues pass 1 of scene i+1 on its GPU BEFORE the host stages of scene i, so that rank 0's serial section (mask ->
    graph points, reference graph_extraction.py:130-139 / graph_utils.py:572-591, and the final vote merge) runs while all GPUs —
    its own included — are busy with the next scene's pass 1.  Per scene and rank the device then sees pass 1 / world + pass 2 /
    world back to back; the exchange steps are the three of infer_one_img (banded canvas reduce, point broadcast, vote gather) in
    the same order on every rank:   stage1(i+1): reduce_canvases(i+1)   |   stage2(i): broadcast_points(i), gather_*_votes(i).
    Results per scene are those of infer_one_img under the same world size (same kernels, same summation orders); rank 0 yields
    the tuples, the other ranks yield None.  `stats` (a dict) collects per-stage wall times and the bytes of every exchange."""
    import time
    device = torch.device(device) if device is not None else next(net.parameters()).device
    world = torch.distributed.get_world_size() if D.is_distributed() else 1
    rank = torch.distributed.get_rank() if D.is_distributed() else 0
    bs = int(config.INFER_BATCH_SIZE)
    cuda = device.type == "cuda"
    stats = stats if stats is not None else {}
    for k in ("pass1_queue_ms", "points_host_ms", "pass2_ms", "merge_host_ms", "canvas_bytes", "points_bytes", "votes_bytes", "scenes"):
        stats.setdefault(k, 0.0)

    def stage1(img):
        job = _SceneJob()
        t0 = time.perf_counter()
        job.img, job.infos, job.all_xy = _scene_plan(img, config)
        job.lo, job.hi = shard_tiles(len(job.infos), world, rank)
        scene = torch.as_tensor(np.ascontiguousarray(job.img), dtype=torch.uint8).to(device, non_blocking=cuda)
        job.xy_dev = torch.as_tensor(job.all_xy).to(device)
        kp_c, road_c, job.emb = net.scene_pass1(scene, job.xy_dev[job.lo:job.hi], bs)
        bands = D.tile_bands(job.all_xy, int(config.PATCH_SIZE), world) if world > 1 else None
        D.reduce_canvases(kp_c, road_c, dst=0, bands=bands)
        if bands is not None:
            # this rank's own share: the band it ships to rank 0 (rank 0: what it receives), so that per-rank statistics are per rank
            x0, x1 = bands[rank]
            stats["canvas_bytes"] += D.canvas_bytes(bands, job.img.shape[0]) if rank == 0 else 2 * 4 * job.img.shape[0] * max(0, x1 - x0)
        job.masks = job.e1 = None
        if rank == 0:
            kp_u8, road_u8 = net.scene_normalise(kp_c, road_c, job.xy_dev)
            if cuda:       # asynchronous download behind the scene's own kernels: the host does not wait here
                job.masks = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (kp_u8, road_u8)]
                for h, t in zip(job.masks, (kp_u8, road_u8)):
                    h.copy_(t, non_blocking=True)
                job.e1 = torch.cuda.Event()
                job.e1.record(torch.cuda.current_stream(device))
            else:
                job.masks = [kp_u8, road_u8]
        stats["pass1_queue_ms"] += 1e3 * (time.perf_counter() - t0)
        return job

    def stage2(job):
        t0 = time.perf_counter()
        graph_points = kp_mask = road_mask = None
        if rank == 0:
            if job.e1 is not None:
                job.e1.synchronize()
            kp_mask, road_mask = (np.array(m.numpy() if isinstance(m, torch.Tensor) else m) for m in job.masks)
            graph_points = extract_graph_points(kp_mask, road_mask, config)
        t1 = time.perf_counter()
        stats["points_host_ms"] += 1e3 * (t1 - t0)
        graph_points = D.broadcast_points(graph_points, src=0, device=device if world > 1 else None)
        stats["points_bytes"] += 16 * graph_points.shape[0] * ((world - 1) if rank == 0 else 1)     # rank 0: sent to every peer; others: received
        if graph_points.shape[0] == 0:
            return None if rank != 0 else (graph_points, np.zeros((0, 2), dtype=np.int32), kp_mask, road_mask)
        n_pts = graph_points.shape[0]
        if world > 1 and config.EXACT_VOTE_MERGE:
            k_raw, s_raw = edge_votes(net, job.emb, graph_points, job.infos, job.lo, job.hi, config, device, raw=True)
            t2 = time.perf_counter()
            stats["votes_bytes"] += 16 * k_raw.shape[0]
            k_raw, s_raw = D.gather_raw_votes(k_raw, s_raw, dst=0, device=device)
            if rank != 0:
                stats["pass2_ms"] += 1e3 * (t2 - t1)
                return None
            uk, sums, cnts, first = _accumulate_votes(k_raw, s_raw)
        else:
            uk, sums, cnts, first = edge_votes(net, job.emb, graph_points, job.infos, job.lo, job.hi, config, device)
            t2 = time.perf_counter()
            stats["votes_bytes"] += 32 * uk.shape[0]
            uk, sums, cnts, first = D.gather_edge_votes(uk, sums, cnts, n_pts, dst=0, device=device if world > 1 else None, first=first)
        stats["pass2_ms"] += 1e3 * (t2 - t1)
        job.emb = None
        if rank != 0:
            return None
        t3 = time.perf_counter()
        pred_edges = votes_to_edges(uk, sums, cnts, first, n_pts, config.TOPO_THRESHOLD)
        stats["merge_host_ms"] += 1e3 * (time.perf_counter() - t3)
        return graph_points[:, ::-1], pred_edges, kp_mask, road_mask

    it = iter(imgs)
    img = next(it, None)
    if img is None:
        return
    with _host_quiet():
        cur = stage1(img)
    while cur is not None:
        with _host_quiet():
            img = next(it, None)
            nxt = stage1(img) if img is not None else None        # the next scene's pass 1 is on the device before this scene's host work
            res = stage2(cur)
        stats["scenes"] += 1
        yield res
        cur = nxt


def get_img_paths(root_dir, image_indices):
    """inferencer.py:38-44."""
    import os
    return [os.path.join(root_dir, f"region_{ind}_sat.png") for ind in image_indices]


def read_rgb_img(path):
    """dataset.py:16-19 (cv2.imread + BGR->RGB): [H,W,3] uint8 RGB.  PIL decodes the same 8-bit PNGs."""
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB")))


def cityscale_data_partition():
    """dataset.py:21-38: (train, validation, test) region indices of the 180 CityScale regions."""
    train = [x for x in range(180) if x % 10 < 8]
    test = [x for x in range(180) if x % 10 == 9 or x % 20 == 8]
    val = [x for x in range(180) if x % 20 == 18]
    return train, val, test


def spacenet_data_partition():
    """dataset.py:41-53: reads ./spacenet/data_split.json relative to the working directory, as the reference does."""
    import json
    with open("./spacenet/data_split.json", "r") as jf:
        d = json.load(jf)
    return d["train"], d["validation"], d["test"]


def create_output_dir_and_save_config(output_dir_prefix, config, specified_dir=None):
    """utils.py:11-29."""
    import os
    import yaml
    from datetime import datetime
    out = specified_dir if specified_dir else f"{output_dir_prefix}_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "config.yaml"), "w") as f:
        yaml.dump(config.to_dict(), f)
    return out


def _build_net(config, checkpoint, device):
    """inferencer.py:246-254.  Under torchrun only rank 0 reads the checkpoint: it packs the weights once and the packed arena
    is broadcast device-to-device over RCCL (SAMRoad.share_packed_weights); the other ranks never touch the file."""
    from .model import SAMRoad
    net = SAMRoad(config)
    shared = D.is_distributed() and device.type == "cuda"
    if not shared or torch.distributed.get_rank() == 0:
        ckpt = torch.load(checkpoint, map_location="cpu")
        print(f"##### Loading Trained CKPT {checkpoint} #####")
        net.load_state_dict(ckpt["state_dict"], strict=True)
    net.eval()
    net.to(device)
    if shared:
        net.share_packed_weights(src=0)
    return net


def main(argv=None):
    """Drop-in for `python inferencer.py --config ... --checkpoint ... [--output_dir ...] [--device cuda]` (reference
    inferencer.py:24-35,239-349), run from a sam_road checkout: enumerates the test split of config.DATASET
    (./cityscale/20cities/region_{}_sat.png or ./spacenet/RGB_1.0_meter/{}__rgb.png), runs the scenes through infer_imgs, writes
    save/<output_dir>/{config.yaml, mask/{id}_road.png, mask/{id}_itsc.png, graph/{id}.p, inference_time.txt} with the
    reference's formats (8-bit grayscale PNG masks, sat2graph pickle, SpaceNet (400 - r, c) flip).  Not reproduced: the cv2
    `viz/` renderings and the ground-truth pickle the reference loads but only uses in commented-out code (visualisation,
    SURVEY §2 #17).  Extras: `--images a.png b.npy ...` runs explicit scene files instead of the dataset split; under torchrun
    (one process per GPU) the scenes are dealt round-robin to the ranks (`--shard scenes`, default) or every scene's tiles are split
    over the ranks (`--shard tiles`; `tiles-pipelined` for the software-pipelined loop), all ranks writing into the one output directory."""
    import argparse
    import os
    import pickle
    import time
    from PIL import Image
    from .config import load_config
    from .formats import convert_to_sat2graph_format
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default=None, help="checkpoint of the model to test.")
    ap.add_argument("--config", default=None, help="model config.")
    ap.add_argument("--output_dir", default=None, help="Name of the output dir, if not specified will use timestamp")
    ap.add_argument("--device", default="cuda", help="device to use (an MI355X: there is no CPU path)")
    ap.add_argument("--images", nargs="*", default=None, help="(extension) explicit scene images instead of the dataset split")
    ap.add_argument("--shard", choices=("scenes", "tiles", "tiles-pipelined"), default="scenes",
                    help="(extension, multi-GPU runs under torchrun) scenes: every rank takes whole scenes, no data-path collective "
                         "(throughput); tiles: the tiles of every scene are split over the ranks, scene by scene (latency of one scene); "
                         "tiles-pipelined: the same with scene i+1's pass 1 queued before scene i's host stages (opt-in: exercised on gloo only)")
    args = ap.parse_args(argv)
    config = load_config(args.config)
    device = torch.device("cuda") if args.device == "cuda" else torch.device(args.device)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // max(1, int(os.environ.get("WORLD_SIZE", "1"))))))   # this rank's share of the container's CPU quota (hostcpu.py)
    _numpy_hugepages(False)                      # for the whole run: image decoding and output encoding allocate beside the GPU too (_host_quiet)
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dist = torch.distributed
    if world > 1:                                # one process per GPU (torchrun); the reference is single-process (inferencer.py:243)
        if device.type == "cuda":
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if device.type == "cuda":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group("gloo")
    by_scene = world > 1 and args.shard == "scenes"
    net = _build_net(config, args.checkpoint, device)

    if args.images is not None:
        jobs = [(os.path.splitext(os.path.basename(p))[0], p) for p in args.images]
    elif config.DATASET == "cityscale":
        _, _, test_img_indices = cityscale_data_partition()
        jobs = [(i, "./cityscale/20cities/region_{}_sat.png".format(i)) for i in test_img_indices]
    elif config.DATASET == "spacenet":
        _, _, test_img_indices = spacenet_data_partition()
        jobs = [(i, "./spacenet/RGB_1.0_meter/{}__rgb.png".format(i)) for i in test_img_indices]
    else:
        raise ValueError(f"config.DATASET must be 'cityscale' or 'spacenet' (got {config.DATASET!r}), or pass --images")
    if by_scene:
        jobs = jobs[rank::world]                 # independent scenes: round-robin over the ranks, nothing to exchange

    output_dir_prefix = "./save/infer_"
    if world > 1:                                # one directory for all ranks: rank 0 creates it (and its timestamp), the others wait
        name = [None]
        if rank == 0:
            name[0] = create_output_dir_and_save_config(output_dir_prefix, config,
                                                        specified_dir=f"./save/{args.output_dir}" if args.output_dir else None)
        dist.broadcast_object_list(name, src=0)
        output_dir = name[0]
    elif args.output_dir:
        output_dir = create_output_dir_and_save_config(output_dir_prefix, config, specified_dir=f"./save/{args.output_dir}")
    else:
        output_dir = create_output_dir_and_save_config(output_dir_prefix, config)

    from concurrent.futures import ThreadPoolExecutor

    def scenes(depth=3):                         # decode ahead on worker threads: a 2048^2 RGB PNG takes ~100 ms to decode, more than
        load = lambda path: np.load(path) if str(path).endswith(".npy") else read_rgb_img(path)   # the GPU needs for the scene
        with ThreadPoolExecutor(depth) as ex:
            futs = {}
            for j in range(len(jobs)):
                for k in range(j, min(j + depth, len(jobs))):
                    if k not in futs:
                        futs[k] = ex.submit(load, jobs[k][1])
                yield futs.pop(j).result()

    mask_save_dir, graph_save_dir = os.path.join(output_dir, "mask"), os.path.join(output_dir, "graph")

    os.makedirs(mask_save_dir, exist_ok=True)
    os.makedirs(graph_save_dir, exist_ok=True)

    def write_png(mask, name):
        # zlib level 1 = cv2.imwrite's default for PNG (the reference, inferencer.py:303-304); PIL's default level 6 takes ~50 ms per
        # 2048^2 mask — two masks per scene would make the encoder, not the GPU, the bottleneck of the loop.  Same pixels either way.
        Image.fromarray(mask).save(os.path.join(mask_save_dir, name), compress_level=1)

    def write_graph(img_id, pred_nodes, pred_edges):                             # inferencer.py:330-343
        if config.DATASET == "spacenet":
            pred_nodes = np.stack([400 - pred_nodes[:, 0], pred_nodes[:, 1]], axis=1)   # inferencer.py:332-334
        with open(os.path.join(graph_save_dir, f"{img_id}.p"), "wb") as f:
            pickle.dump(convert_to_sat2graph_format(pred_nodes, pred_edges), f)
        print(f"Done for {img_id}.")

    # the reference times infer_one_img per image (inferencer.py:292-296); the scenes are software-pipelined here (infer_imgs), so
    # the time reported is what the loop spends waiting for results — its sum over the images is the wall time of inference.  PNG
    # encoding and pickling (tens of ms per 2048^2 scene) run on two writer threads so that the loop goes straight back to the GPU.
    total_inference_seconds = 0.0
    results = infer_imgs(net, scenes(), config, device=device, tile_sharded=world > 1 and not by_scene,
                         pipelined=True if args.shard == "tiles-pipelined" else None)
    with ThreadPoolExecutor(2) as writer:
        pending = []
        for img_id, path in jobs:
            print(f"Processing {img_id}")
            start_seconds = time.time()
            res = next(results)
            total_inference_seconds += time.time() - start_seconds
            if res is None:                      # non-zero rank of a tile-sharded run
                continue
            pred_nodes, pred_edges, itsc_mask, road_mask = res
            pending += [writer.submit(write_png, road_mask, f"{img_id}_road.png"), writer.submit(write_png, itsc_mask, f"{img_id}_itsc.png"),
                        writer.submit(write_graph, img_id, pred_nodes, pred_edges)]
        for f in pending:
            f.result()

    if world > 1:                                # the slowest rank is the wall time of the run
        t = torch.tensor([total_inference_seconds], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total_inference_seconds = float(t.item())
    time_txt = f"Inference completed for {args.config} in {total_inference_seconds} seconds."
    print(time_txt)
    if rank == 0:
        with open(os.path.join(output_dir, "inference_time.txt"), "w") as f:
            f.write(time_txt)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()

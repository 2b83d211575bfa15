"""CPU tests of the host-side mirror of the reference interface: config semantics, tile grid, mask->points,
pass-2 query builder, output formats, SAMRoad state_dict layout, and that the C-ABI library loads and exports
every symbol include/samroad_hip.h declares (no compute calls without a GPU)."""
import os
import re
import warnings

import numpy as np
import pytest
import torch

from sam_road_amd import Config, SAMRoad, get_patch_info_one_img, load_config
from sam_road_amd import _lib
from sam_road_amd.formats import convert_from_sat2graph_format, convert_to_sat2graph_format
from sam_road_amd.graph_points import extract_graph_points, nms_points
from sam_road_amd.inferencer import build_patch_queries
from sam_road_amd.tiling import shard_tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_missing_key_is_falsy(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text("SAM_VERSION: 'vit_h'\nPATCH_SIZE: 256\n")
    cfg = load_config(str(p))
    assert cfg.SAM_VERSION == "vit_h" and cfg.PATCH_SIZE == 256
    assert not cfg.NO_SAM and not cfg.USE_SAM_DECODER          # reference utils.py:6-9 (addict) semantics
    assert cfg.TOPONET_VERSION != "no_transformer"


def test_patch_info_matches_reference_golden(golden_dir):
    g = np.load(f"{golden_dir}/patch_info.npz")
    i = 0
    while f"case{i}" in g:
        info = get_patch_info_one_img(0, *[int(v) for v in g[f"case{i}"]])
        got = np.array([[p[1][0], p[1][1], p[2][0], p[2][1]] for p in info])
        np.testing.assert_array_equal(got, g[f"xy{i}"])
        i += 1
    # CityScale: 256 tiles, origins 64..1472, x outer / y inner
    info = get_patch_info_one_img(0, 2048, 64, 512, 16)
    assert len(info) == 256 and info[0][1] == (64, 64) and info[1][1] == (64, 158) and info[-1][1] == (1472, 1472)


def test_nms_points_matches_reference_golden(golden_dir):
    g = np.load(f"{golden_dir}/nms_points.npz")
    for i in range(4):
        np.testing.assert_array_equal(nms_points(g[f"pts{i}"], g[f"sc{i}"], int(g[f"r{i}"])), g[f"kept{i}"])
    np.testing.assert_array_equal(nms_points(g["pts_p"], g["sc_p"], 16), g["kept_p"])
    assert nms_points(np.zeros((0, 2), np.int64), np.zeros((0,)), 8).shape == (0, 2)   # empty input


@pytest.mark.parametrize("seed,n,radius,forced_frac", [(0, 3000, 8, 0.0), (1, 3000, 16, 0.05), (2, 500, 3, 0.5),
                                                       (3, 2000, 16, 1.0), (4, 1, 8, 0.0), (5, 4000, 1, 0.0)])
def test_nms_points_host_code_matches_oracle_loop(seed, n, radius, forced_frac):
    """srh_nms_points_host (grid, C++) against the oracle's literal KDTree loop (reference graph_utils.py:572-591) on
    dense integer candidates with heavy score ties, forced (> 1.0) scores, duplicates and boundary distances."""
    from oracle import scene as oscene
    rng = np.random.default_rng(seed)
    pts = rng.integers(0, 120, size=(n, 2)).astype(np.int64)          # dense: many neighbours, duplicate points
    sc = rng.integers(0, 4, size=n).astype(np.float64) / 4.0             # ties; all <= 1.0
    sc[rng.random(n) < forced_frac] = 200.0                              # forced candidates (u8-mask-like scores)
    np.testing.assert_array_equal(nms_points(pts, sc, radius), oscene.nms_points(pts, sc, radius))


def test_extract_graph_points_and_queries():
    rng = np.random.default_rng(0)
    kp = (rng.random((256, 256)) ** 8 * 255).astype(np.uint8)
    road = (rng.random((256, 256)) ** 4 * 255).astype(np.uint8)
    cfg = Config(ITSC_THRESHOLD=0.5, ROAD_THRESHOLD=0.6, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16,
                 NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
    pts = extract_graph_points(kp, road, cfg)
    from oracle import scene as oscene
    from oracle.samroad import AttrDict
    np.testing.assert_array_equal(pts, oscene.extract_graph_points(kp, road, AttrDict(cfg)))
    d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1) + np.eye(len(pts)) * 1e9
    assert d.min() > 8 - 1e-6
    ids, p, pairs, valid = build_patch_queries(pts, 32, 32, 160, 160, cfg)
    o_ids, o_p, o_pairs, o_valid = oscene.build_patch_queries(pts, (0, (32, 32), (160, 160)), AttrDict(cfg))
    np.testing.assert_array_equal(ids, o_ids)
    np.testing.assert_array_equal(pairs, o_pairs)
    np.testing.assert_array_equal(valid, o_valid)
    assert (pairs[..., 0] == np.arange(len(ids))[:, None]).all()
    assert (valid[:, :-1] >= valid[:, 1:]).all()                      # valid is a prefix (App. D.5)
    assert (pairs[..., 1][~valid] == pairs[..., 0][~valid]).all()     # invalid target -> source
    # the library's all-tiles builder: same ids / valid / neighbour SETS per source (order inside a group of equidistant
    # neighbours is scipy-internal), scipy fallback for tiles whose k-th neighbour is tied
    from sam_road_amd.inferencer import build_all_patch_queries
    from sam_road_amd.tiling import get_patch_info_one_img
    infos = get_patch_info_one_img(0, 256, 0, 128, 4)
    allq = build_all_patch_queries(pts, infos, 0, len(infos), cfg)
    assert len(allq) == len(infos)
    for info, (a_ids, a_p, a_pairs, a_valid) in zip(infos, allq):
        r_ids, r_p, r_pairs, r_valid = oscene.build_patch_queries(pts, info, AttrDict(cfg))
        np.testing.assert_array_equal(a_ids, r_ids)
        np.testing.assert_array_equal(a_p, r_p)
        np.testing.assert_array_equal(a_valid, r_valid)
        assert a_pairs.shape == r_pairs.shape and (a_pairs[..., 0] == r_pairs[..., 0]).all()
        for i in range(len(a_ids)):
            assert set(a_pairs[i, a_valid[i], 1].tolist()) == set(r_pairs[i, r_valid[i], 1].tolist())
            assert (a_pairs[i, ~a_valid[i], 1] == i).all()
    # empty tile
    ids, p, pairs, valid = build_patch_queries(pts, 5000, 5000, 5100, 5100, cfg)
    assert p.shape == (0, 2) and pairs.shape == (0, 16, 2) and valid.shape == (0, 16)


def test_queries_with_tied_cutoff_match_reference_kdtree():
    """Integer pixel coordinates make equidistant neighbours common; when the tie straddles the K-th slot the chosen
    neighbour is decided by the kd-tree's shape, i.e. by the reference's `scipy.spatial.KDTree(pts)` defaults
    (inferencer.py:156; leafsize 10 — cKDTree's default of 16 picks differently).  Lattice-like points: every source
    has a tied cutoff."""
    from oracle import scene as oscene
    from oracle.samroad import AttrDict
    from sam_road_amd.inferencer import build_all_patch_queries
    from sam_road_amd.tiling import get_patch_info_one_img
    cfg = Config(NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
    rng = np.random.default_rng(3)
    g = np.arange(4, 380, 12)
    pts = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    pts = pts[rng.random(len(pts)) < 0.8].astype(np.int64)          # holes break the symmetry
    infos = get_patch_info_one_img(0, 384, 0, 256, 3)
    fq = build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
    n_src = n_tied = 0
    for t, info in enumerate(infos):
        a_ids, a_p, a_pairs, a_valid = fq.tile(t)
        tied = fq.tied[int(fq.offsets[t]):int(fq.offsets[t + 1])].astype(bool)
        r_ids, r_p, r_pairs, r_valid = oscene.build_patch_queries(pts, info, AttrDict(cfg))
        np.testing.assert_array_equal(a_ids, r_ids)
        np.testing.assert_array_equal(a_valid, r_valid)
        one = build_patch_queries(pts, *info[1], *info[2], cfg)
        np.testing.assert_array_equal(one[2], r_pairs)               # the single-tile path is the reference's call verbatim
        for i in range(len(a_ids)):
            assert set(a_pairs[i, a_valid[i], 1].tolist()) == set(r_pairs[i, r_valid[i], 1].tolist()), (info, i)
            n_src += 1
        # rows with a tied cut-off were decided by the library's restatement of scipy's kd-tree (csrc/kdtree_emul.hpp): they
        # equal the reference's call element for element, ORDER included
        np.testing.assert_array_equal(a_pairs[tied], r_pairs[tied])
        n_tied += int(tied.sum())
    assert n_src > 500 and n_tied > 300


def test_kdtree_restatement_matches_scipy():
    """csrc/kdtree_emul.hpp against scipy itself: the tree's index permutation (tree.indices — i.e. which points share a leaf and
    in which order, nth_element included) and the k-NN answers in scipy's output order, on data where ties are everywhere:
    dense lattices with duplicate points, 8-pixel lattices, uniform integers, NMS-like point sets; random k / radius; query
    points inside and outside the data."""
    import ctypes as C
    import scipy.spatial
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(0)
    for case in range(240):
        n = int(rng.integers(1, 700))
        mode = case % 4
        if mode == 0:
            pts = rng.integers(0, int(rng.integers(2, 40)), size=(n, 2))
        elif mode == 1:
            pts = rng.integers(0, 64, size=(n, 2)) * 8
        elif mode == 2:
            pts = rng.integers(0, 513, size=(n, 2))
        else:
            side = int(np.ceil(np.sqrt(n))) + 2
            allp = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2) * int(rng.integers(1, 20))
            pts = allp[rng.permutation(len(allp))[:n]]
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        k, r = (17, 64.0) if case % 3 else (int(rng.integers(1, 24)), float(rng.integers(1, 120)))
        q = np.ascontiguousarray(np.concatenate([pts, rng.integers(-30, 560, size=(8, 2)).astype(np.float64)]))
        tree = scipy.spatial.KDTree(pts)                     # the reference's class and defaults (leafsize 10)
        _, want = tree.query(q, k=k, distance_upper_bound=r)
        want = np.asarray(want).reshape(len(q), k)
        got = np.zeros((len(q), k), dtype=np.int32)
        perm = np.zeros(len(pts), dtype=np.int32)
        assert lib.srh_kdtree_knn_host(vp(pts), len(pts), 10, vp(q), len(q), k, r, vp(got), vp(perm)) == 0
        np.testing.assert_array_equal(perm, tree.indices)
        np.testing.assert_array_equal(got, want)


def test_mask_candidates_match_numpy_where():
    """srh_mask_candidates == np.where(mask > thr) + mask[sel] (graph_extraction.py:24-28), incl. fractional thresholds,
    nothing / everything above, and a non-square mask."""
    from sam_road_amd.graph_points import points_and_scores_from_mask
    rng = np.random.default_rng(5)
    for shape, thr in [((257, 300), 63.24), ((64, 64), 254.5), ((64, 64), 255.0), ((50, 70), -1.0), ((128, 128), 0.0), ((33, 31), 92.82)]:
        m = rng.integers(0, 256, size=shape).astype(np.uint8)
        xy, sc = points_and_scores_from_mask(m, thr)
        sel = m > thr
        rc = np.column_stack(np.where(sel))
        np.testing.assert_array_equal(xy, rc[:, ::-1])
        np.testing.assert_array_equal(sc, m[sel])
        assert xy.dtype == np.int64 and sc.dtype == np.uint8


def test_edge_vote_accumulate_matches_reference_dict_loop():
    """srh_edge_vote_accumulate == the reference's dict accumulation (inferencer.py:209-221), bit for bit: float64 sums in
    visiting order per (src, tgt) key; also the empty and single-key edge cases."""
    import ctypes as C
    from collections import defaultdict
    from sam_road_amd import _lib
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(k, s):
        uk, su, cn, fi = np.empty_like(k), np.empty_like(s), np.empty_like(s), np.empty_like(k)
        nu = C.c_int64(-1)
        assert lib.srh_edge_vote_accumulate(vp(k), vp(s), k.shape[0], vp(uk), vp(su), vp(cn), vp(fi), C.byref(nu)) == 0
        # the first-vote positions reproduce the dict's insertion order
        assert uk[:nu.value][np.argsort(fi[:nu.value])].tolist() == list(dict.fromkeys(k.tolist()))
        return uk[:nu.value], su[:nu.value], cn[:nu.value]

    rng = np.random.default_rng(11)
    for n_pts, n in [(50, 4000), (5000, 200000), (70000, 30000), (3, 1)]:
        src, tgt = rng.integers(0, n_pts, n), rng.integers(0, n_pts, n)
        k = (src.astype(np.int64) * n_pts + tgt).astype(np.int64)
        s = rng.random(n).astype(np.float32).astype(np.float64)
        sums, cnts = defaultdict(float), defaultdict(float)
        for kk, ss in zip(k.tolist(), s.tolist()):
            sums[kk] += ss
            cnts[kk] += 1.0
        uk, su, cn = run(k, s)
        want = sorted(sums)
        np.testing.assert_array_equal(uk, want)
        assert su.tolist() == [sums[q] for q in want]            # bit-identical float64 sums
        assert cn.tolist() == [cnts[q] for q in want]
    uk, su, cn = run(np.zeros(0, np.int64), np.zeros(0, np.float64))
    assert uk.shape == (0,) and su.shape == (0,)


def test_sat2graph_format_kats():
    """The reference's own known-answer tests (graph_utils.py:687-702)."""
    nodes = np.array([[0.0, 0.0], [1.1, 1.1], [1.6, 1.6]])
    edges = np.array([[0, 1], [1, 2]])
    got = convert_to_sat2graph_format(nodes, edges)
    want = {(0, 0): [(1, 1)], (1, 1): [(0, 0), (2, 2)], (2, 2): [(1, 1)]}
    assert set(got) == set(want) and all(set(got[k]) == set(want[k]) for k in want)
    n, e = convert_from_sat2graph_format({(0, 0): [(1, 1)], (1, 1): [(0, 0), (2, 2)], (2, 2): [(1, 1)]})
    np.testing.assert_array_equal(n, [[0, 0], [1, 1], [2, 2]])
    np.testing.assert_array_equal(np.array(e), [[0, 1], [1, 0], [1, 2], [2, 1]])


def test_shard_tiles_partition():
    for n, w in [(256, 8), (256, 3), (64, 4), (5, 8), (0, 2)]:
        spans = [shard_tiles(n, w, r) for r in range(w)]
        covered = [i for lo, hi in spans for i in range(lo, hi)]
        assert covered == list(range(n))


@pytest.mark.parametrize("version,patch,topo", [("vit_b", 512, "normal"), ("vit_b", 256, "no_transformer"), ("vit_l", 256, "normal")])
def test_state_dict_layout_matches_appendix_a(version, patch, topo):
    warnings.simplefilter("ignore")
    net = SAMRoad(Config(SAM_VERSION=version, PATCH_SIZE=patch, TOPONET_VERSION=topo, SAM_CKPT_PATH=""))
    sd = net.state_dict()
    S = patch // 16
    D = {"vit_b": 768, "vit_l": 1024}[version]
    assert tuple(sd["image_encoder.pos_embed"].shape) == (1, S, S, D)
    assert tuple(sd["image_encoder.patch_embed.proj.weight"].shape) == (D, 3, 16, 16)
    assert tuple(sd["image_encoder.blocks.0.attn.qkv.weight"].shape) == (3 * D, D)
    assert tuple(sd["image_encoder.blocks.0.attn.rel_pos_h"].shape) == (27, 64)
    g = {"vit_b": 2, "vit_l": 5}[version]
    assert tuple(sd[f"image_encoder.blocks.{g}.attn.rel_pos_w"].shape) == (2 * S - 1, 64)
    assert tuple(sd["image_encoder.neck.2.weight"].shape) == (256, 256, 3, 3)
    assert tuple(sd["map_decoder.0.weight"].shape) == (256, 128, 2, 2)
    assert tuple(sd["map_decoder.7.bias"].shape) == (2,)
    assert tuple(sd["topo_net.pair_proj.weight"].shape) == (128, 258)
    assert ("topo_net.transformer_encoder.layers.2.self_attn.in_proj_weight" in sd) == (topo != "no_transformer")
    assert "pixel_mean" not in sd                                        # non-persistent (model.py:229-230)
    # same key set / shapes as the oracle (one state dict drives both)
    from oracle.samroad import AttrDict, SAMRoadOracle
    o = SAMRoadOracle(AttrDict(SAM_VERSION=version, PATCH_SIZE=patch, TOPONET_VERSION=topo)).state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in o.items()}
    net.load_state_dict(o, strict=True)


def test_lora_keys_and_unsupported_configs():
    warnings.simplefilter("ignore")
    net = SAMRoad(Config(SAM_VERSION="vit_b", PATCH_SIZE=256, ENCODER_LORA=True, LORA_RANK=4, SAM_CKPT_PATH=""))
    sd = net.state_dict()
    assert tuple(sd["image_encoder.blocks.3.attn.qkv.linear_a_q.weight"].shape) == (4, 768)
    assert tuple(sd["image_encoder.blocks.3.attn.qkv.linear_b_v.weight"].shape) == (768, 4)
    assert "image_encoder.blocks.3.attn.qkv.weight" in sd
    with pytest.raises(NotImplementedError):
        SAMRoad(Config(SAM_VERSION="vit_b", PATCH_SIZE=256, NO_SAM=True))
    dec = SAMRoad(Config(SAM_VERSION="vit_b", PATCH_SIZE=256, USE_SAM_DECODER=True, SAM_CKPT_PATH="")).state_dict()
    assert "mask_decoder.mask_tokens.weight" in dec and not any(k.startswith("map_decoder") for k in dec)
    with pytest.raises(AssertionError):
        SAMRoad(Config(SAM_VERSION="vit_x", PATCH_SIZE=256))


def test_no_cpu_fallback():
    warnings.simplefilter("ignore")
    net = SAMRoad(Config(SAM_VERSION="vit_b", PATCH_SIZE=256, SAM_CKPT_PATH="", ENCODER_DEPTH=1))
    with pytest.raises(_lib.SrhError):
        net.infer_masks_and_img_features(torch.zeros(1, 256, 256, 3))


def test_c_abi_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "samroad_hip.h")).read()
    declared = set(re.findall(r"\b(srh_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.srh_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define SRH_ABI_VERSION (\d+)", header).group(1))
    # the loaded library is the build of the sources in this tree (content hash compiled into it)
    from sam_road_amd.build import library_id, source_id
    assert _lib.build_id() == library_id() == source_id()
    # product package must not reference the oracle
    pkg = os.path.join(ROOT, "sam_road_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_config_missing_keys_copy_pickle_and_toponet_default():
    """a YAML without TOPONET_VERSION (toponet_vith_256 / vitl_256 / vitb_256 / vitb_1024) must behave as 'normal';
    Config survives copy.deepcopy / pickle (addict.Dict does)."""
    import copy
    import pickle
    import warnings
    from sam_road_amd import Config, SAMRoad
    cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=256, SAM_CKPT_PATH="", ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[],
                 nested=dict(a=1))
    assert not cfg.TOPONET_VERSION and not cfg.NO_SAM and cfg.TOPONET_VERSION != "no_transformer"
    assert copy.deepcopy(cfg) == cfg and pickle.loads(pickle.dumps(cfg)) == cfg
    warnings.simplefilter("ignore")
    net = SAMRoad(cfg)
    assert net._topo_version == "normal" and hasattr(net.topo_net, "transformer_encoder")
    assert copy.deepcopy(net).state_dict().keys() == net.state_dict().keys()
    net2 = SAMRoad(Config(dict(cfg, TOPONET_VERSION="no_transformer")))
    assert net2._topo_version == "no_transformer" and not hasattr(net2.topo_net, "transformer_encoder")


def test_pass2_votes_match_reference_triple_loop():
    """srh_pass2_votes == the reference's triple loop over (tile, source point, neighbour slot) (inferencer.py:209-221): same
    keys and scores in the same visiting order; a score outside [0, 1] is refused (the reference asserts)."""
    import ctypes as C
    from sam_road_amd.inferencer import build_all_patch_queries
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(5)
    pts = np.unique(rng.integers(0, 300, size=(400, 2)), axis=0).astype(np.int64)
    cfg = Config(NEIGHBOR_RADIUS=40, MAX_NEIGHBOR_QUERIES=16)
    infos = get_patch_info_one_img(0, 320, 0, 128, 4)
    fq = build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
    K, n_pts = 16, pts.shape[0]
    nb, t0 = 5, 3                                     # a "batch" of tiles 3..7
    cnt = np.diff(fq.offsets[t0:t0 + nb + 1])
    n_max = int(cnt.max()) + 2                        # padded rows beyond a tile's points are never read
    scores = rng.random((nb, n_max, K)).astype(np.float32)
    cap = int((fq.knn >= 0).sum())
    keys, votes = np.empty(cap, np.int64), np.empty(cap, np.float64)
    c = C.c_int64(0)
    assert lib.srh_pass2_votes(vp(scores), nb, n_max, K, vp(fq.offsets[t0:]), vp(fq.ids), vp(fq.knn), n_pts, vp(keys), vp(votes),
                               cap, C.byref(c)) == 0
    want_k, want_s = [], []
    for b in range(nb):
        ids, _, pairs, valid = fq.tile(t0 + b)
        for si in range(len(ids)):
            for pi in range(K):
                if valid[si, pi]:
                    want_k.append(int(ids[pairs[si, pi, 0]]) * n_pts + int(ids[pairs[si, pi, 1]]))
                    want_s.append(float(scores[b, si, pi]))
    assert c.value == len(want_k) > 100
    np.testing.assert_array_equal(keys[:c.value], want_k)
    np.testing.assert_array_equal(votes[:c.value], want_s)
    scores[1, 0, 0] = 1.5 if fq.knn[fq.offsets[t0 + 1], 0] >= 0 else scores[1, 0, 0]
    scores[0, 0, 0] = np.nan
    c = C.c_int64(0)
    if fq.knn[fq.offsets[t0], 0] >= 0:
        assert lib.srh_pass2_votes(vp(scores), nb, n_max, K, vp(fq.offsets[t0:]), vp(fq.ids), vp(fq.knn), n_pts, vp(keys), vp(votes),
                                   cap, C.byref(c)) != 0


def test_hostcpu_and_host_quiet_restore_process_state():
    """usable_cpus() never exceeds the affinity mask; _host_quiet() switches the garbage collector and numpy's huge-page madvise off
    inside and restores exactly the previous state (also when the body raises, and when they were already off)."""
    import gc
    import os
    from sam_road_amd import inferencer as inf
    from sam_road_amd.hostcpu import usable_cpus, worker_threads
    n = usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and 1 <= worker_threads() <= max(1, n // 2) and worker_threads(cap=2) <= 2
    from sam_road_amd.hostcpu import fill_threads
    old = os.environ.get("LOCAL_WORLD_SIZE")
    try:                                                   # the ranks of one node share its CPUs (torchrun's LOCAL_WORLD_SIZE)
        os.environ["LOCAL_WORLD_SIZE"] = "1"
        one = (worker_threads(), fill_threads())
        os.environ["LOCAL_WORLD_SIZE"] = "4"
        assert 1 <= fill_threads() <= max(1, n // 4) and 1 <= worker_threads() <= max(1, n // 8) and fill_threads() <= one[1]
    finally:
        if old is None:
            os.environ.pop("LOCAL_WORLD_SIZE", None)
        else:
            os.environ["LOCAL_WORLD_SIZE"] = old
    was = inf._numpy_hugepages(True)                       # known starting point
    assert was in (True, False)
    try:
        assert gc.isenabled()
        with inf._host_quiet():
            assert not gc.isenabled() and inf._numpy_hugepages(False) is False
        assert gc.isenabled() and inf._numpy_hugepages(True) is True
        with pytest.raises(RuntimeError):
            with inf._host_quiet():
                raise RuntimeError("x")
        assert gc.isenabled() and inf._numpy_hugepages(True) is True
        gc.disable()
        inf._numpy_hugepages(False)
        with inf._host_quiet():
            pass
        assert not gc.isenabled() and inf._numpy_hugepages(False) is False      # stays off if it was off
    finally:
        gc.enable()
        inf._numpy_hugepages(was)


def test_edge_vote_accumulate_threads_identical():
    """srh_edge_vote_accumulate_mt with 1 / 3 / 8 worker threads: identical unique keys, float64 sums (bit for bit), counts and
    first-vote positions — on ~0.7 M votes keyed like a dense scene (src * n + tgt, every key voted ~20 times in scattered order)."""
    import ctypes as C
    from sam_road_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    n_pts = 5000
    src = rng.integers(0, n_pts, size=700_000)
    tgt = (src + rng.integers(1, 17, size=src.shape[0])) % n_pts
    keys = np.ascontiguousarray(src * n_pts + tgt, dtype=np.int64)
    scores = np.ascontiguousarray(rng.random(keys.shape[0]), dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    outs = []
    for nt in (1, 3, 8):
        uk, sums, cnts, first = np.empty_like(keys), np.empty_like(scores), np.empty_like(scores), np.empty_like(keys)
        nu = C.c_int64(0)
        assert lib.srh_edge_vote_accumulate_mt(vp(keys), vp(scores), keys.shape[0], vp(uk), vp(sums), vp(cnts), vp(first), C.byref(nu), nt) == 0
        outs.append(tuple(a[:nu.value].copy() for a in (uk, sums, cnts, first)))
    ref_k, inv = np.unique(keys, return_inverse=True)
    np.testing.assert_array_equal(outs[0][0], ref_k)
    np.testing.assert_array_equal(outs[0][2], np.bincount(inv).astype(np.float64))
    assert np.array_equal(keys[outs[0][3]], ref_k) and (np.diff(outs[0][0]) > 0).all()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert a.dtype == b.dtype and np.array_equal(a, b)        # array_equal on float64: exact
    # the sums are the sequential float64 sums in vote order
    want = np.zeros(ref_k.shape[0])
    for i in np.nonzero(inv < 50)[0]:
        want[inv[i]] += scores[i]
    np.testing.assert_array_equal(outs[2][1][:50], want[:50])


def test_pass2_pack_matches_per_tile_collate():
    """_pack_pass2_batches (srh_pass2_pack on the flat query arrays) == zero-padded collate of the per-tile (points, pairs, valid)
    tuples (inferencer.py:179-185), for ragged batches incl. an empty tile; the staging arrays may hold garbage beforehand."""
    from sam_road_amd import Config
    from sam_road_amd import inferencer as inf
    from sam_road_amd.tiling import get_patch_info_one_img
    rng = np.random.default_rng(3)
    cfg = Config(PATCH_SIZE=256, SAMPLE_MARGIN=16, INFER_PATCHES_PER_EDGE=4, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16, INFER_BATCH_SIZE=5)
    infos = get_patch_info_one_img(0, 640, 16, 256, 4)
    pts = rng.integers(300, 640, size=(400, 2)).astype(np.int64)            # the top-left tiles stay empty
    pts = pts[np.unique(pts[:, 0] * 1000 + pts[:, 1], return_index=True)[1]]
    fq = inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
    assert (np.diff(fq.offsets) == 0).any() and (np.diff(fq.offsets) > 20).any()
    K, bs = 16, 5
    junk = lambda name, shape, dtype: np.full(shape, 77, dtype)
    counts = np.diff(fq.offsets)
    for sort_tiles in (False, True):             # the reference's batches of consecutive tiles / tiles grouped by row count
        plan, p_h, q_h, v_h = inf._pack_pass2_batches(fq, 0, len(infos), bs, K, junk, sort_tiles=sort_tiles)
        assert len(plan) >= 2
        seen = np.concatenate([t for t, _, _ in plan])
        assert len(set(seen.tolist())) == len(seen) and set(np.flatnonzero(counts > 0).tolist()) <= set(seen.tolist())
        rows = 0
        for tile_idx, n_max, base in plan:
            assert base == rows and n_max == counts[tile_idx].max() and len(tile_idx) <= bs
            nb = len(tile_idx)
            rows += nb * n_max
            tiles = [fq.tile(int(t)) for t in tile_idx]
            sl = slice(base, base + nb * n_max)
            np.testing.assert_array_equal(p_h[sl].reshape(nb, n_max, 2), inf._collate([t[1].astype(np.float32) for t in tiles]))
            np.testing.assert_array_equal(q_h[sl].reshape(nb, n_max, K, 2), inf._collate([t[2].astype(np.int32) for t in tiles]))
            np.testing.assert_array_equal(v_h[sl].reshape(nb, n_max, K), inf._collate([t[3].astype(np.uint8) for t in tiles]))
        if sort_tiles:
            sorted_rows = rows
            assert all(counts[t].min() > 0 for t, _, _ in plan)                      # empty tiles join no batch
        else:
            scan_rows = rows
    assert sorted_rows <= scan_rows


def test_pass2_ragged_pack_is_the_unpadded_collate():
    """_pack_pass2_ragged (srh_pass2_pack_ragged: the rows of srh_toponet_ragged) == the per-tile (points, pairs, valid) tuples laid end
    to end, pairs shifted to rows of the flat list, every row naming its tile; a tile's scores are rows offsets[t] .. offsets[t + 1]
    (_ragged_batches feeds exactly those to the vote code), so the votes computed from a flat score array equal those computed from
    the padded batches holding the same scores."""
    from sam_road_amd import Config
    from sam_road_amd import inferencer as inf
    from sam_road_amd.tiling import get_patch_info_one_img
    rng = np.random.default_rng(3)
    cfg = Config(PATCH_SIZE=256, SAMPLE_MARGIN=16, INFER_PATCHES_PER_EDGE=4, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16, INFER_BATCH_SIZE=5)
    infos = get_patch_info_one_img(0, 640, 16, 256, 4)
    pts = rng.integers(300, 640, size=(400, 2)).astype(np.int64)            # the top-left tiles stay empty
    pts = pts[np.unique(pts[:, 0] * 1000 + pts[:, 1], return_index=True)[1]]
    fq = inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
    K = 16
    junk = lambda name, shape, dtype: np.full(shape, 77, dtype)
    R, p_h, t_h, q_h, v_h = inf._pack_pass2_ragged(fq, K, junk)
    off = np.asarray(fq.offsets)
    assert R == off[-1] and (np.diff(off) == 0).any()
    for t in range(fq.n_tiles):
        a, b = int(off[t]), int(off[t + 1])
        _, tp, tq, tv = fq.tile(t)
        np.testing.assert_array_equal(p_h[a:b], tp.astype(np.float32))
        np.testing.assert_array_equal(q_h[a:b], tq.astype(np.int32) + a)            # tile-local indices -> rows of the flat list
        np.testing.assert_array_equal(v_h[a:b], tv.astype(np.uint8))
        assert (t_h[a:b] == t).all()
    # votes from one flat score array == votes from padded batches holding the same scores
    flat = rng.random((R, K)).astype(np.float32)
    ragged = inf._ragged_batches(fq, flat)
    plan, _ = inf._pass2_plan(fq, 5, sort_tiles=True)
    padded = []
    for tiles, n_max, _ in plan:
        sc = np.zeros((len(tiles), n_max, K), np.float32)
        for j, t in enumerate(tiles):
            sc[j, :off[t + 1] - off[t]] = flat[off[t]:off[t + 1]]
        padded.append((tiles, sc))
    n_pts = pts.shape[0]
    for x, y in zip(inf._vote_sums(fq, 0, ragged, n_pts, K), inf._vote_sums(fq, 0, padded, n_pts, K)):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    for x, y in zip(inf._votes_from_scores(fq, 0, ragged, n_pts, K), inf._votes_from_scores(fq, 0, padded, n_pts, K)):
        assert np.array_equal(x, y)


def test_pass2_vote_sums_equal_reference_dicts():
    """srh_pass2_vote_sums (rows grouped by source point, a table of targets per point) == the reference's dicts filled by the
    triple loop (inferencer.py:206-228): keys, float64 sums in visiting order, counts and insertion positions, bit for bit; any
    thread count; several batches with different paddings; an out-of-range score is refused."""
    from collections import OrderedDict
    from sam_road_amd import inferencer as inf
    rng = np.random.default_rng(8)
    pts = np.unique(rng.integers(0, 420, size=(900, 2)), axis=0).astype(np.int64)
    cfg = Config(NEIGHBOR_RADIUS=40, MAX_NEIGHBOR_QUERIES=16)
    infos = get_patch_info_one_img(0, 448, 0, 128, 6)
    fq = inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
    K, n_pts, n_tiles = 16, pts.shape[0], len(infos)
    batches, bs = [], 7
    for off in range(0, n_tiles, bs):
        end = min(off + bs, n_tiles)
        n_max = int(np.diff(fq.offsets[off:end + 1]).max()) + (off % 3)
        if n_max:
            batches.append((off, end, rng.random((end - off, n_max, K)).astype(np.float32)))
    sums, cnts = OrderedDict(), OrderedDict()
    for off, end, sc in batches:                                  # the reference's loop
        for b in range(end - off):
            ids, _, pairs, valid = fq.tile(off + b)
            for si in range(len(ids)):
                for pi in range(K):
                    if valid[si, pi]:
                        key = int(ids[pairs[si, pi, 0]]) * n_pts + int(ids[pairs[si, pi, 1]])
                        sums[key] = sums.get(key, 0.0) + float(sc[b, si, pi])
                        cnts[key] = cnts.get(key, 0.0) + 1.0
    assert len(sums) > 1000
    old = inf._accumulate_votes(*inf._votes_from_scores(fq, 0, batches, n_pts, K))
    wt = inf.worker_threads
    try:
        for nt in (1, 2, 5):
            inf.worker_threads = lambda: nt
            uk, su, cn, fi = inf._vote_sums(fq, 0, batches, n_pts, K)
            assert uk.tolist() == sorted(sums)
            assert su.tolist() == [sums[k] for k in uk.tolist()] and cn.tolist() == [cnts[k] for k in uk.tolist()]
            assert uk[np.argsort(fi)].tolist() == list(sums)      # first-vote positions = the dict's insertion order
            for a, b in zip((uk, su, cn, fi), old):
                np.testing.assert_array_equal(a, b)
    finally:
        inf.worker_threads = wt
    # where a tile's scores lie does not matter: the same scores in batches of tiles grouped by row count give the same sums
    plan, _ = inf._pass2_plan(fq, bs, sort_tiles=True)
    where = {}
    for off, end, sc in batches:
        for b in range(end - off):
            where[off + b] = sc[b]
    permuted = []
    for tile_idx, n_max, _ in plan:
        sc = np.full((len(tile_idx), n_max + 1, K), 7.0, np.float32)          # padding is never read
        for jj, t in enumerate(tile_idx):
            n = int(fq.offsets[t + 1] - fq.offsets[t])
            sc[jj, :n] = where[int(t)][:n]
        permuted.append((tile_idx, sc))
    for a, b in zip(inf._vote_sums(fq, 0, permuted, n_pts, K), old):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(inf._votes_from_scores(fq, 0, permuted, n_pts, K), inf._votes_from_scores(fq, 0, batches, n_pts, K)):
        np.testing.assert_array_equal(a, b)
    off, end, sc = batches[1]
    a = int(fq.offsets[off])
    j = int(np.argmax(fq.knn[a] >= 0))
    bad = sc.copy(); bad[0, 0, j] = 1.25
    with pytest.raises(AssertionError):
        inf._vote_sums(fq, 0, [batches[0], (off, end, bad)] + batches[2:], n_pts, K)
    # a non-empty tile that no batch covers is an inconsistency, not a silent loss of votes
    with pytest.raises(AssertionError):
        inf._vote_sums(fq, 0, batches[1:], n_pts, K)


def test_mask_candidates_threads_and_capacity():
    """srh_mask_candidates with worker threads (bands of rows) == np.where order for every thread count, and the single-call
    protocol: a too small capacity reports the needed one."""
    import ctypes as C
    from sam_road_amd.graph_points import points_and_scores_from_mask
    lib = _lib.load()
    rng = np.random.default_rng(12)
    m = (rng.random((700, 333)) ** 6 * 255).astype(np.uint8)
    m[100:180] = 0                                                # bands without a candidate
    sel = m > 100.5
    want_xy, want_sc = np.column_stack(np.where(sel))[:, ::-1], m[sel]
    for nt in (1, 2, 3, 8, 64):
        xy, sc = points_and_scores_from_mask(m, 100.5, nt)
        np.testing.assert_array_equal(xy, want_xy)
        np.testing.assert_array_equal(sc, want_sc)
    n = C.c_int64(0)
    xy, sc = np.empty((10, 2), np.int64), np.empty(10, np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.srh_mask_candidates(vp(m), 700, 333, 100.5, vp(xy), vp(sc), 10, C.byref(n), 4) != 0 and n.value == len(want_sc)
    assert lib.srh_mask_candidates(vp(m), 700, 333, 100.5, None, None, 0, C.byref(n), 4) == 0 and n.value == len(want_sc)


def test_extract_graph_points_fast_path_equals_general_path():
    """The u8 fast path of extract_graph_points (threaded scans + one srh_nms_merge_points call) returns exactly what the three
    nms_points calls of graph_extraction.py:130-139 return (general path, forced here with non-contiguous masks)."""
    from oracle import scene as oscene
    from oracle.samroad import AttrDict
    rng = np.random.default_rng(21)
    for size, pk, pr in ((384, 8, 4), (640, 12, 6)):
        kp2 = (rng.random((size, 2 * size)) ** pk * 255).astype(np.uint8)
        road2 = (rng.random((size, 2 * size)) ** pr * 255).astype(np.uint8)
        kp_nc, road_nc = kp2[:, ::2], road2[:, ::2]               # same pixels, not C-contiguous -> general path
        kp, road = np.ascontiguousarray(kp_nc), np.ascontiguousarray(road_nc)
        cfg = Config(ITSC_THRESHOLD=0.4, ROAD_THRESHOLD=0.5, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16)
        fast = extract_graph_points(kp, road, cfg)
        np.testing.assert_array_equal(fast, extract_graph_points(kp_nc, road_nc, cfg))
        np.testing.assert_array_equal(fast, oscene.extract_graph_points(kp, road, AttrDict(cfg)))
        assert fast.dtype == np.int64 and fast.flags.c_contiguous and len(fast) > 50
    empty = np.zeros((128, 128), np.uint8)
    assert extract_graph_points(empty, empty, cfg).shape == (0, 2)
    np.testing.assert_array_equal(extract_graph_points(empty, road[:128, :128].copy(), cfg),
                                  extract_graph_points(empty[:, ::1][::1, :], np.asfortranarray(road[:128, :128]), cfg))


def test_pass2_fill_local_coordinates_and_thread_counts():
    """srh_pass2_fill's `local` output = point - tile origin (inferencer.py:151), and ids / knn / tied flags do not depend on the
    number of worker threads (shared pre-pass, tiles taken from a counter)."""
    from sam_road_amd import inferencer as inf
    rng = np.random.default_rng(2)
    pts = np.unique(rng.integers(0, 500, size=(1500, 2)), axis=0).astype(np.int64)
    cfg = Config(NEIGHBOR_RADIUS=48, MAX_NEIGHBOR_QUERIES=16)
    infos = get_patch_info_one_img(0, 512, 0, 160, 5)
    wt = inf.fill_threads
    try:
        inf.fill_threads = lambda: 1
        ref = inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
        tile_of = np.repeat(np.arange(len(infos)), np.diff(ref.offsets))
        origin = np.array([info[1] for info in infos], dtype=np.int64)
        np.testing.assert_array_equal(ref.local, pts[ref.ids] - origin[tile_of])
        for nt in (2, 7, 32):
            inf.fill_threads = lambda: nt
            fq = inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
            for name in ("offsets", "ids", "local", "knn", "tied"):
                np.testing.assert_array_equal(getattr(fq, name), getattr(ref, name))
    finally:
        inf.fill_threads = wt


def test_votes_to_edges_matches_numpy_form():
    """srh_votes_to_edges == the numpy statement of inferencer.py:224-228 (mean > threshold, insertion order) for dense positions
    (one process), positions offset per rank (multi-rank merge) and repeated positions (stable order)."""
    from sam_road_amd.inferencer import votes_to_edges
    rng = np.random.default_rng(4)
    def numpy_form(uk, sums, cnts, first, n_pts, thr):
        keep = (sums / np.maximum(cnts, 1.0)) > thr
        k = uk[keep][np.argsort(first[keep], kind="stable")]
        return np.stack([k // n_pts, k % n_pts], axis=1).reshape(-1, 2)
    n_pts, n = 700, 5000
    uk = np.sort(rng.choice(n_pts * n_pts, n, replace=False)).astype(np.int64)
    cnts = rng.integers(0, 6, n).astype(np.float64)
    sums = rng.random(n) * np.maximum(cnts, 1.0)
    sums[::7] = 0.499 * np.maximum(cnts[::7], 1.0)                # exactly at / next to the threshold
    dense = rng.permutation(40000)[:n].astype(np.int64)
    for first in (dense, dense + (rng.integers(0, 3, n).astype(np.int64) << 40), rng.integers(0, 50, n).astype(np.int64), np.zeros(n, np.int64)):
        got = votes_to_edges(uk, sums, cnts, first, n_pts, 0.499)
        np.testing.assert_array_equal(got, numpy_form(uk, sums, cnts, first, n_pts, 0.499))
        assert got.dtype == np.int64 and got.shape[1] == 2 and 0 < len(got) < n
    assert votes_to_edges(uk[:0], sums[:0], cnts[:0], dense[:0], n_pts, 0.499).shape == (0, 2)


def test_pass2_vote_sums_on_a_tile_sub_range():
    """srh_pass2_vote_sums called through the C ABI for tiles [t0, t0 + nb) only (offsets pointer advanced, ids / knn absolute): the
    sums of exactly those tiles' votes, first-vote positions counted from the sub-range's first vote."""
    import ctypes as C
    from sam_road_amd import inferencer as inf
    lib = _lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(17)
    pts = np.unique(rng.integers(0, 300, size=(500, 2)), axis=0).astype(np.int64)
    cfg = Config(NEIGHBOR_RADIUS=40, MAX_NEIGHBOR_QUERIES=16)
    infos = get_patch_info_one_img(0, 320, 0, 128, 4)
    fq = inf.build_all_patch_queries(pts, infos, 0, len(infos), cfg, flat=True)
    K, n_pts, t0, nb = 16, pts.shape[0], 5, 6
    n_max = int(np.diff(fq.offsets[t0:t0 + nb + 1]).max())
    sc = rng.random((nb, n_max, K)).astype(np.float32)
    want = inf._accumulate_votes(*inf._votes_from_scores(fq, 0, [(t0, t0 + nb, sc)], n_pts, K))
    cap = int(fq.knn.size)
    uk, su, cn, fi = np.empty(cap, np.int64), np.empty(cap, np.float64), np.empty(cap, np.float64), np.empty(cap, np.int64)
    nu = C.c_int64(0)
    ptrs = (C.c_void_p * 1)(sc.ctypes.data)
    tile0, cnt, nm = np.zeros(1, np.int32), np.full(1, nb, np.int32), np.full(1, n_max, np.int64)
    assert lib.srh_pass2_vote_sums(ptrs, vp(tile0), vp(cnt), vp(nm), 1, K, vp(fq.offsets[t0:]), nb, vp(fq.ids), vp(fq.knn), n_pts,
                                   vp(uk), vp(su), vp(cn), vp(fi), cap, C.byref(nu), 3) == 0
    assert nu.value == len(want[0]) > 50
    for got, w in zip((uk, su, cn, fi), want):
        np.testing.assert_array_equal(got[:nu.value], w)

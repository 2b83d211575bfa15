"""Scene-level parity (pass 1 of infer_one_img and the whole pipeline) of the HIP path vs the CPU oracle.
Run on an MI355X: pytest -m gpu."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scene as oscene
from oracle.samroad import AttrDict, SAMRoadOracle
from oracle.synth import synth_scene, synth_state_dict

CFG = dict(SAM_VERSION="vit_b", PATCH_SIZE=256, TOPONET_VERSION="normal", SAM_CKPT_PATH="",
           ENCODER_DEPTH=2, ENCODER_GLOBAL_ATTN_INDEXES=[1],
           INFER_BATCH_SIZE=5, SAMPLE_MARGIN=16, INFER_PATCHES_PER_EDGE=4,
           ITSC_THRESHOLD=0.5, ROAD_THRESHOLD=0.5, TOPO_THRESHOLD=0.5,
           ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
SCENE = 448


@pytest.fixture(scope="module")
def pair():
    from sam_road_amd import Config, SAMRoad
    warnings.simplefilter("ignore")
    oracle = SAMRoadOracle(AttrDict(CFG)).eval()
    sd = synth_state_dict(oracle, 77)
    sd["map_decoder.7.bias"] = torch.tensor([-0.3, 0.2])   # denser masks than the default -3
    oracle.load_state_dict(sd, strict=True)
    net = SAMRoad(Config(CFG))
    net.load_state_dict(sd, strict=True)
    net.eval().to("cuda")
    return oracle, net


def oracle_pass1(oracle, img, cfg):
    infos = oscene.get_patch_info_one_img(0, img.shape[0], cfg.SAMPLE_MARGIN, cfg.PATCH_SIZE, cfg.INFER_PATCHES_PER_EDGE)
    bs = cfg.INFER_BATCH_SIZE
    scores = [oracle.infer_masks_and_img_features(oscene.get_batch_img_patches(img, infos[i:i + bs]))[0]
              for i in range(0, len(infos), bs)]
    return infos, oscene.fuse_masks(img.shape[:2], infos, scores)


def test_scene_pass1_masks(pair):
    oracle, net = pair
    cfg = AttrDict(CFG)
    img = synth_scene(SCENE, seed=5)
    infos, (kp_ref, road_ref) = oracle_pass1(oracle, img, cfg)
    xy = torch.tensor([[p[1][0], p[1][1]] for p in infos], dtype=torch.int32)
    scene = torch.as_tensor(img).cuda()
    kp_c, road_c, emb = net.scene_pass1(scene, xy.cuda(), cfg.INFER_BATCH_SIZE)     # ragged last batch (16 = 3*5+1)
    kp, road = net.scene_normalise(kp_c, road_c, xy.cuda())
    kp, road = kp.cpu().numpy(), road.cpu().numpy()
    assert emb.shape == (16, 256, 16, 16)
    for got, ref in ((kp, kp_ref), (road, road_ref)):
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() <= 2 and (d <= 1).mean() >= 0.999, (d.max(), (d <= 1).mean())
    m = cfg.SAMPLE_MARGIN
    assert (kp[:m] == 0).all() and (kp[:, :m] == 0).all() and (road[-m:] == 0).all()   # uncovered border: NaN -> 0
    assert kp_ref.max() > 0 and road_ref.max() > 0


def test_infer_one_img_end_to_end(pair):
    """Whole pipeline.  Greedy NMS on the u8 masks is chaotic w.r.t. +-1-level differences (one different
    pick cascades), so the graph is compared stage-wise on IDENTICAL intermediate inputs: masks (pass 1),
    then edge votes of pass 2 on the same point set; the product's own infer_one_img must reproduce the
    product-side stages exactly."""
    from sam_road_amd import Config
    from sam_road_amd.graph_points import extract_graph_points
    from sam_road_amd.inferencer import infer_one_img
    oracle, net = pair
    img = synth_scene(SCENE, seed=6)
    cfg = dict(CFG)
    infos, feats, kp_r, road_r = oscene.infer_pass1(oracle, img, AttrDict(cfg))
    cfg["ITSC_THRESHOLD"] = float(np.percentile(kp_r[kp_r > 0], 99.5)) / 255.0
    cfg["ROAD_THRESHOLD"] = float(np.percentile(road_r[road_r > 0], 98.0)) / 255.0
    nodes, edges, kp, road = infer_one_img(net, img, Config(cfg))
    assert (np.abs(kp.astype(int) - kp_r.astype(int)) <= 2).all()
    assert (np.abs(road.astype(int) - road_r.astype(int)) <= 2).all()
    # points: product host stage on the product masks == oracle host stage on the same masks
    pts = extract_graph_points(kp, road, Config(cfg))
    np.testing.assert_array_equal(pts, oscene.extract_graph_points(kp, road, AttrDict(cfg)))
    np.testing.assert_array_equal(nodes, pts[:, ::-1])
    assert pts.shape[0] > 20, "synthetic scene produced too few points to be a meaningful test"
    # pass 2 on the same points: oracle (its own fp32 features) vs HIP (its own features)
    edges_r, sums_r, cnts_r = oscene.infer_pass2(oracle, feats, pts, infos, AttrDict(cfg))
    got = {(int(a), int(b)) for a, b in edges.tolist()}
    ref = {(int(a), int(b)) for a, b in edges_r.tolist()}
    # every oracle edge decision with a margin > 0.003 from the threshold must be reproduced (measured max |HIP - oracle| mean edge
    # score over 12k edges: 6e-4, tools/scene_edge_diag.py)
    firm = {e for e, s in sums_r.items() if abs(s / cnts_r[e] - cfg["TOPO_THRESHOLD"]) > 0.003}
    assert {e for e in ref if e in firm} == {e for e in got if e in firm}
    assert len(got ^ ref) <= max(2, 0.02 * len(ref))
    assert len(sums_r) > 50


def test_infer_imgs_pipeline_equals_serial(pair):
    """The software-pipelined scene loop (infer_imgs: side-stream uploads from page-locked staging, asynchronous mask / score
    downloads behind events, pass 1 of the next scene queued before this scene's host stages) returns exactly what infer_one_img
    returns for every scene — five different scenes of three sizes, so both staging pools are reused and regrown and results of neighbouring scenes would
    show up as differences if a buffer were recycled too early."""
    from sam_road_amd import Config
    from sam_road_amd.inferencer import infer_imgs, infer_one_img
    _, net = pair
    # different scene sizes in one run: the staging pools grow and are re-viewed per scene
    imgs = [synth_scene(size, seed=s) for size, s in ((SCENE, 6), (384, 7), (SCENE, 8), (512, 9), (384, 10))]
    _, _, kp0, road0 = infer_one_img(net, imgs[0], Config(dict(CFG)))
    cfg = Config(dict(CFG, ITSC_THRESHOLD=float(np.percentile(kp0[kp0 > 0], 99.5)) / 255.0,
                      ROAD_THRESHOLD=float(np.percentile(road0[road0 > 0], 98.0)) / 255.0))
    want = [infer_one_img(net, im, cfg) for im in imgs]
    print("points / edges per scene:", [(w[0].shape[0], w[1].shape[0]) for w in want])
    assert len({w[0].shape[0] for w in want}) > 1 and min(w[0].shape[0] for w in want) > 20 and max(w[1].shape[0] for w in want) > 20
    for _ in range(2):
        got = list(infer_imgs(net, iter(imgs), cfg))
        assert len(got) == len(want)
        for w, g in zip(want, got):
            for a, b in zip(w, g):
                np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_pass2_batching_sorted_vs_consecutive(pair):
    """TopoNet batches of tiles grouped by row count (the default: non-contiguous tiles, embeddings gathered in their storage
    layout) against batches of consecutive tiles (the reference's): every tile's rows are scored independently of its batch
    mates, so the pipeline must yield the same nodes, the same masks and the same edges in the same order."""
    from sam_road_amd import Config
    from sam_road_amd import inferencer as inf
    _, net = pair
    cfg = Config(dict(CFG, PASS2_RAGGED=False))         # the padded batches (the unpadded launch has no batches to group)
    imgs = [synth_scene(SCENE, seed=21), synth_scene(SCENE, seed=22)]
    assert inf.PASS2_SORT_TILES
    try:
        got = {}
        for flag in (True, False):
            inf.PASS2_SORT_TILES = flag
            got[flag] = list(inf.infer_imgs(net, iter(imgs), cfg, tile_sharded=False))
    finally:
        inf.PASS2_SORT_TILES = True
    assert got[True][0][0].shape[0] > 30 and got[True][0][1].shape[0] > 100
    for a, b in zip(got[True], got[False]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


def test_pass2_ragged_rows_equal_padded_batches(pair):
    """Pass 2 as ONE unpadded launch over all query rows of a scene (srh_toponet_ragged, the default) against the reference's padded
    batches (PASS2_RAGGED: False; inferencer.py:179-207): every row is scored on its own, so nodes, masks and the edge list — order
    included — are identical, through the serial call and through the pipelined loop; and at the op level the scores of the flat rows
    are bit for bit the scores the padded call gives the same rows."""
    from sam_road_amd import Config
    from sam_road_amd import inferencer as inf
    _, net = pair
    imgs = [synth_scene(SCENE, seed=31), synth_scene(384, seed=32), synth_scene(SCENE, seed=33)]
    got = {}
    for ragged in (True, False):
        cfg = Config(dict(CFG, PASS2_RAGGED=ragged))
        got[ragged] = (list(inf.infer_imgs(net, iter(imgs), cfg, tile_sharded=False)), [inf.infer_one_img(net, im, cfg) for im in imgs])
    assert got[True][0][0][0].shape[0] > 30 and got[True][0][0][1].shape[0] > 100
    for runs in zip(got[True][0], got[False][0], got[True][1], got[False][1]):
        for arrs in zip(*runs):
            for other in arrs[1:]:
                np.testing.assert_array_equal(np.asarray(arrs[0]), np.asarray(other))
    # op level: a batch of tiles with different point counts, padded vs flat
    cfg = Config(CFG)
    dev = next(net.parameters()).device
    img, infos, all_xy = inf._scene_plan(imgs[0], cfg)
    scene = torch.as_tensor(np.ascontiguousarray(img)).to(dev)
    _, _, emb = net.scene_pass1(scene, torch.as_tensor(all_xy).to(dev), int(cfg.INFER_BATCH_SIZE))
    nodes = got[True][1][0][0][:, ::-1].copy()              # (x, y) graph points of the scene
    fq = inf.build_all_patch_queries(np.ascontiguousarray(nodes), infos, 0, len(infos), cfg, flat=True)
    K = int(cfg.MAX_NEIGHBOR_QUERIES)
    R, p_h, t_h, q_h, v_h = inf._pack_pass2_ragged(fq, K)
    flat = net.infer_toponet_ragged(emb, *(torch.from_numpy(x[:R]).to(dev) for x in (p_h, t_h, q_h, v_h))).cpu().numpy()
    # ABI 8: with the tiles' row offsets the library scores the scene in chunks of whole tiles (<= 16 384 rows per launch, so its
    # workspace does not grow with the scene) — the same bits.  The scene is replicated until several chunks are needed.
    offs0 = inf._ragged_offsets(fq)
    np.testing.assert_array_equal(net.infer_toponet_ragged(emb, *(torch.from_numpy(x[:R]).to(dev) for x in (p_h, t_h, q_h, v_h)),
                                                           tile_offsets=offs0).cpu().numpy(), flat)
    rep, nt = int(np.ceil(40000 / R)), len(infos)
    assert rep * R <= 65536
    big = [np.concatenate([p_h[:R]] * rep), np.concatenate([t_h[:R] + i * nt for i in range(rep)]).astype(np.int32),
           np.concatenate([q_h[:R] + i * R for i in range(rep)]).astype(np.int32), np.concatenate([v_h[:R]] * rep)]
    big_off = np.concatenate([offs0[:-1] + i * R for i in range(rep)] + [np.array([rep * R], dtype=np.int64)])
    emb_rep = emb.repeat(rep, 1, 1, 1)
    one = net.infer_toponet_ragged(emb_rep, *(torch.from_numpy(x).to(dev) for x in big)).cpu().numpy()
    chunked = net.infer_toponet_ragged(emb_rep, *(torch.from_numpy(x).to(dev) for x in big), tile_offsets=big_off).cpu().numpy()
    vbig = big[3].astype(bool)
    np.testing.assert_array_equal(chunked[vbig], one[vbig])
    np.testing.assert_array_equal(chunked[:R][v_h[:R].astype(bool)], flat[v_h[:R].astype(bool)])
    from sam_road_amd import _lib
    with pytest.raises(_lib.SrhError):                      # no offsets: one launch, bounded
        net.infer_toponet_ragged(emb_rep.repeat(2, 1, 1, 1), *(torch.from_numpy(np.concatenate([x, x])).to(dev) for x in big))
    plan, pp, pq, pv = inf._pack_pass2_batches(fq, 0, len(infos), int(cfg.INFER_BATCH_SIZE), K)
    pts_d, pairs_d, valid_d = (torch.from_numpy(x).to(dev) for x in (pp, pq, pv))
    off = np.asarray(fq.offsets)
    checked = 0
    for tiles, sc in inf._launch_pass2_batches(net, emb, plan, pts_d, pairs_d, valid_d, K):
        sc = sc.cpu().numpy()
        for j, t in enumerate(tiles):
            n = int(off[t + 1] - off[t])
            vm = v_h[off[t]:off[t + 1]].astype(bool)
            np.testing.assert_array_equal(sc[j, :n][vm], flat[off[t]:off[t + 1]][vm])
            checked += int(vm.sum())
    assert checked > 500


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the box (RCCL over xGMI); gpurun boxes expose one")
def test_bench_two_gpus_rccl_tile_sharded_scene():
    """The first thing to run on a multi-GPU box: `python bench.py --gpus 2` (bare form: it re-execs under torch.distributed.run) —
    RCCL connects both ranks, the packed-weight broadcast, the serial AND the pipelined tile-sharded scene loops complete under RCCL
    and yield the same graph (sam_road_amd/distributed.py: batched point-to-point band reduce, point broadcast, vote gather)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-reference-gpu", "--no-workloads", "--no-sustained", "--scenes", "3", "--scene-timeout", "300"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    js = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1, r.stdout[-2000:]
    line = js[0]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["value"] > 0
    sc = line["scene"]
    assert "error" not in sc, sc
    assert sc["rccl_ranks"] == 2 and sc["pipelined_loop"].startswith("completed; same graph"), sc

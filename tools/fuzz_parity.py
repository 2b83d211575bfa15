#!/usr/bin/env python
"""Randomised HIP-vs-oracle parity sweep (run on an MI355X): random batch sizes (incl. 1 and odd ones that leave GEMM row tails),
random point counts per tile (incl. 1, ragged, all-invalid rows), every TOPONET_VERSION, ViT-B 256 / 512 / 1024-px tiles at depth 2 and a
small full infer_one_img with odd scene sizes / margins / batch sizes.  Prints one line per case and a summary; exit code 1 on any
violation of the stated tolerances (tests/tolerances.py).   python tools/fuzz_parity.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import tolerances as T          # the stated tolerances (tests/tolerances.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    from oracle import scene as oscene
    from oracle.samroad import AttrDict, SAMRoadOracle
    from oracle.synth import synth_queries, synth_scene, synth_state_dict, synth_tiles
    from sam_road_amd import Config, SAMRoad
    from sam_road_amd.hostcpu import usable_cpus
    from sam_road_amd.inferencer import infer_imgs, infer_one_img
    torch.set_num_threads(usable_cpus())
    rng = np.random.default_rng(args.seed)
    bad, t0 = 0, time.time()
    nets = {}

    def pair(P, ver, gidx, sam="vit_b"):
        key = (P, ver, tuple(gidx), sam)
        if key not in nets:
            cfg = dict(SAM_VERSION=sam, PATCH_SIZE=P, TOPONET_VERSION=ver, SAM_CKPT_PATH="", ENCODER_DEPTH=2,
                       ENCODER_GLOBAL_ATTN_INDEXES=list(gidx))
            o = SAMRoadOracle(AttrDict(cfg)).eval()
            sd = synth_state_dict(o, 1000 + len(nets))
            sd["map_decoder.7.bias"] = torch.tensor([-0.3, 0.2])
            o.load_state_dict(sd, strict=True)
            n = SAMRoad(Config(cfg)); n.load_state_dict(sd, strict=True); n.eval().to("cuda")
            nets[key] = (cfg, o, n)
        return nets[key]

    for c in range(args.cases):
        P = int(rng.choice([256, 256, 512]))
        ver = str(rng.choice(["normal", "no_offset", "no_transformer", "no_tgt_features"]))
        gidx = [int(rng.integers(0, 2))]
        sam = "vit_b"
        if c % 6 == 5:                                           # ViT-L / ViT-H widths (head dim 64 x 16 heads / 80 x 16 heads): 256 px tiles
            P, sam = 256, str(rng.choice(["vit_l", "vit_h"]))
        elif c % 8 == 6:                                         # round 6: 1024-px tiles (toponet_vitb_1024.yaml: 25 windows, the 64 x 64 global window)
            P = 1024
        cfg, oracle, net = pair(P, ver, gidx, sam)
        if c % 4 != 3 or sam != "vit_b" or P == 1024:
            # up to the shipped YAMLs' INFER_BATCH_SIZE = 64 (and one past it): B >= 8 (512 px) / 32 (256 px) takes the persistent q192 GEMMs
            B = (int(rng.choice([1, 2, 3, 5])) if P == 1024 else int(rng.choice([1, 2, 3, 5, 7, 16, 32, 33])) if P == 512
                 else int(rng.choice([1, 2, 3, 5, 9, 17, 32, 64, 65])))
            npts = int(rng.choice([1, 2, 17, 40, 96]))
            rgb = synth_tiles(B, P, seed=int(rng.integers(1 << 30)))
            points, pairs, valid = synth_queries(B, npts, P, seed=int(rng.integers(1 << 30)))
            if rng.random() < 0.3:
                valid[int(rng.integers(B))] = False                       # a tile whose pairs are all invalid
            ml_r, ms_r, tl_r, ts_r = oracle(rgb, points, pairs, valid)
            ml, ms, tl, ts = (t.cpu() for t in net(rgb.cuda(), points.cuda(), pairs.cuda(), valid.cuda()))
            v = valid.bool()
            d_s = (ms - ms_r).abs().max().item()
            d_t = (ts[v] - ts_r[v]).abs().max().item() if v.any() else 0.0
            ok = d_s < T.MASK_SCORE and d_t < T.TOPO_SCORE and torch.isfinite(ml).all() and torch.isfinite(tl[v]).all()
            print(f"case {c:3d} forward {sam} P={P} {ver:16s} global={gidx} B={B:2d} N={npts:3d}: mask {d_s:.1e} topo {d_t:.1e} {'ok' if ok else 'FAIL'}", flush=True)
        else:
            S = P + 2 * 16 + int(rng.integers(0, 5)) * 24
            scfg = dict(cfg, INFER_BATCH_SIZE=int(rng.choice([1, 3, 5, 8])), SAMPLE_MARGIN=16, INFER_PATCHES_PER_EDGE=int(rng.choice([2, 3])),
                        ITSC_THRESHOLD=0.5, ROAD_THRESHOLD=0.5, TOPO_THRESHOLD=0.5, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16,
                        NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
            img = synth_scene(S, seed=int(rng.integers(1 << 30)))
            infos, feats, kp_r, road_r = oscene.infer_pass1(oracle, img, AttrDict(scfg))
            scfg["ITSC_THRESHOLD"] = float(np.percentile(kp_r[kp_r > 0], 99.5)) / 255.0
            scfg["ROAD_THRESHOLD"] = float(np.percentile(road_r[road_r > 0], 98.0)) / 255.0
            nodes, edges, kp, road = infer_one_img(net, img, Config(scfg))
            piped = list(infer_imgs(net, iter([img, img]), Config(scfg)))
            same = all(all(np.array_equal(a, b) for a, b in zip(p, (nodes, edges, kp, road))) for p in piped)
            dk = np.abs(kp.astype(int) - kp_r.astype(int)); dr = np.abs(road.astype(int) - road_r.astype(int))
            pts = oscene.extract_graph_points(kp, road, AttrDict(scfg))
            ok = dk.max() <= 2 and dr.max() <= 2 and np.array_equal(nodes, pts[:, ::-1]) and same
            if ok and pts.shape[0] > 0:
                edges_r, sums_r, cnts_r = oscene.infer_pass2(oracle, feats, pts, infos, AttrDict(scfg))
                got = {(int(a), int(b)) for a, b in edges.tolist()}
                firm = {e for e, s in sums_r.items() if abs(s / cnts_r[e] - scfg["TOPO_THRESHOLD"]) > 0.003}
                ref = {(int(a), int(b)) for a, b in edges_r.tolist()}
                ok = {e for e in ref if e in firm} == {e for e in got if e in firm}
            print(f"case {c:3d} scene   P={P} {ver:16s} S={S} tiles={len(infos)} bs={scfg['INFER_BATCH_SIZE']}: masks +-{max(dk.max(), dr.max())} "
                  f"points {pts.shape[0]} edges {edges.shape[0]} pipelined==serial {same} {'ok' if ok else 'FAIL'}", flush=True)
        bad += 0 if ok else 1
    print(f"{args.cases} cases, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

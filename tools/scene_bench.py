#!/usr/bin/env python
"""Scene-level timing (BASELINE configs[3], one GPU): one synthetic 2048x2048 u8 scene (2 km x 2 km at 1 m/px),
toponet_vitb_512_cityscale.yaml tiling (SAMPLE_MARGIN 64, 16x16 = 256 tiles of 512^2), seeded random weights.
Reports ms per scene for pass 1 alone (crop -> encoder -> decoder -> fused u8 masks, all on the GPU) and for the
whole infer_one_img (pass 1 + host NMS + pass-2 queries + TopoNet + edge vote) = the latency of one scene, and for a run of
scenes through infer_imgs (the CLI's loop: scene i's host stages overlap scene i+1's pass 1) = the throughput figure.
The final map_decoder bias is lowered so that the random network yields sparse masks (a few thousand graph points, as a trained one does).

    python tools/scene_bench.py [--bias -2.2] [--wscale 16] [--batch 64] [--iters 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/scene_bench.py    # N GPUs:
        tiles sharded over the ranks (RCCL: packed-weight broadcast, banded canvas reduce, point broadcast, vote gather);
        rank 0 prints ms/scene (max over ranks) and the per-rank stage times
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bias", type=float, default=-2.2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--wscale", type=float, default=16.0)   # spread of the final layer: with 16 the random net yields ~4k graph points
    args = ap.parse_args()
    from sam_road_amd import Config, SAMRoad
    from sam_road_amd.inferencer import infer_one_img
    from sam_road_amd.tiling import get_patch_info_one_img, shard_tiles
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    from sam_road_amd.hostcpu import usable_cpus
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // world)))      # as the CLI does: respect the container's CPU quota
    cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", DATASET="cityscale",
                 INFER_BATCH_SIZE=args.batch, SAMPLE_MARGIN=64, INFER_PATCHES_PER_EDGE=16, ITSC_THRESHOLD=0.248,
                 ROAD_THRESHOLD=0.364, TOPO_THRESHOLD=0.499, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64,
                 MAX_NEIGHBOR_QUERIES=16)
    net = SAMRoad(cfg)
    g = torch.Generator().manual_seed(1234)
    sd = {}
    for k, v in net.state_dict().items():
        if v.dim() == 1 and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = 0.02 * torch.randn(v.shape, generator=g)
    sd["map_decoder.7.weight"] = args.wscale * torch.randn(sd["map_decoder.7.weight"].shape, generator=g)
    sd["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], args.bias)
    if rank == 0:
        net.load_state_dict(sd, strict=True)
    net.eval().to(dev)
    t_w = time.perf_counter()
    net.share_packed_weights(src=0)          # N > 1: rank 0 packs once, the packed fp16 arena is broadcast over RCCL / xGMI
    torch.cuda.synchronize()
    t_w = time.perf_counter() - t_w
    rng = np.random.default_rng(0)
    coarse = rng.integers(0, 256, size=(2048 // 8, 2048 // 8, 3)).astype(np.float32)
    img = np.kron(coarse, np.ones((8, 8, 1), np.float32)).astype(np.uint8)

    infos = get_patch_info_one_img(0, 2048, cfg.SAMPLE_MARGIN, cfg.PATCH_SIZE, cfg.INFER_PATCHES_PER_EDGE)
    xy = torch.as_tensor(np.array([[p[1][0], p[1][1]] for p in infos], dtype=np.int32)).to(dev)
    scene = torch.as_tensor(img).to(dev)
    lo, hi = shard_tiles(len(infos), world, rank)

    def pass1():              # this rank's share of the tiles (no collective: tile throughput)
        kp, road, emb = net.scene_pass1(scene, xy[lo:hi], args.batch)
        kpu, ru = net.scene_normalise(kp, road, xy)
        return kpu.cpu(), ru.cpu()

    pass1()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        pass1()
    torch.cuda.synchronize()
    p1 = (time.perf_counter() - t0) / args.iters
    # host-stage split: wrap the two host-side stages with timers
    import sam_road_amd.inferencer as inf
    acc = {"extract_graph_points": 0.0, "edge_votes": 0.0}
    def timed(name, fn):
        def w(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
            acc[name] += time.perf_counter() - t
            return r
        return w
    plain = inf.extract_graph_points, inf.edge_votes
    inf.extract_graph_points = timed("extract_graph_points", inf.extract_graph_points)
    inf.edge_votes = timed("edge_votes", inf.edge_votes)
    res = infer_one_img(net, img, cfg)
    for k in acc: acc[k] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        res = infer_one_img(net, img, cfg)
    torch.cuda.synchronize()
    full = (time.perf_counter() - t0) / args.iters
    # throughput of the CLI's scene loop: the same scenes through the software-pipelined generator (one GPU only; the timed
    # region of a run covers 12 scenes from the first upload to the last edge list, so the pipeline's fill and drain are included;
    # the median of the runs is reported next to every run)
    piped, piped_runs, same, piped48 = None, None, None, None
    inf.extract_graph_points, inf.edge_votes = plain            # the timing wrappers synchronise the device
    if world == 1:
        n, piped_runs, same = 12, [], True
        list(inf.infer_imgs(net, (img for _ in range(3)), cfg))
        for _ in range(max(args.iters, 3)):                     # several runs: the rate varies in phases of a second or two
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = list(inf.infer_imgs(net, (img for _ in range(n)), cfg))
            torch.cuda.synchronize()
            piped_runs.append(round(1e3 * (time.perf_counter() - t0) / n, 2))
            same = same and all(all(np.array_equal(a, b) for a, b in zip(o, res)) for o in outs)
        piped = float(np.median(piped_runs)) * 1e-3
        torch.cuda.synchronize()                                 # one long run: fill + drain (~one scene) amortised over 48 scenes
        t0 = time.perf_counter()
        for o in inf.infer_imgs(net, (img for _ in range(48)), cfg):
            pass
        torch.cuda.synchronize()
        piped48 = (time.perf_counter() - t0) / 48
    per_rank = None
    if world > 1:
        # per-rank stage times (ms): pass 1 of the rank's tile chunk, its pass-2 share, the whole call — gathered on rank 0
        mine = torch.tensor([1e3 * p1, 1e3 * acc["edge_votes"] / args.iters, 1e3 * acc["extract_graph_points"] / args.iters, 1e3 * full,
                             float(hi - lo)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(zip(("ms_pass1_own_tiles", "ms_edge_votes", "ms_extract_graph_points", "ms_full", "tiles"),
                             [round(v, 2) for v in a.tolist()])) for a in allr]
        t = torch.tensor([p1, full], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        p1, full = t.tolist()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    nodes, edges, kp, road = res
    print(json.dumps({"scene": "synthetic 2048x2048 u8, 256 tiles of 512^2 (16x16, margin 64)", "n_gpus": world,
                      "ms_weight_share": round(1e3 * t_w, 2) if world > 1 else None, "per_rank": per_rank,
                      "infer_batch_size": args.batch, "ms_per_scene_pass1": round(1e3 * p1, 2),
                      "tiles_per_s_pass1": round(256 / p1, 1), "ms_per_scene_full": round(1e3 * full, 2),
                      "ms_per_scene_pipelined": None if piped is None else round(1e3 * piped, 2), "pipelined_runs_of_12_scenes": piped_runs,
                      "ms_per_scene_pipelined_run_of_48": None if piped48 is None else round(1e3 * piped48, 2),
                      "pipelined_equals_serial": same,
                      "ms_extract_graph_points": round(1e3 * acc["extract_graph_points"] / args.iters, 2),
                      "ms_edge_votes": round(1e3 * acc["edge_votes"] / args.iters, 2), "graph_points": int(nodes.shape[0]), "edges": int(edges.shape[0]),
                      "kp_mask_frac": float((kp > cfg.ITSC_THRESHOLD * 255).mean()),
                      "road_mask_frac": float((road > cfg.ROAD_THRESHOLD * 255).mean())}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: 512x512 ViT-B tiles/sec through SAMRoad.infer_masks_and_img_features.

Workload = BASELINE.json configs[1]: toponet_vitb_512_cityscale.yaml, batch of 16 tiles of 512x512,
ViT-B encoder + mask decoder, synthetic tiles and seeded random weights of that architecture.
One "step" = one batch of 16 tiles per GPU, inputs already resident in HBM (f32 [16,512,512,3], what
the reference hands to the model, inferencer.py:94).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: tiles are independent (pass 1 of infer_one_img shards embarrassingly), so every
rank runs its own batches — weak scaling, no data-path collective; weights are broadcast from rank 0
over RCCL before the timed region.  Prints ONE JSON line on rank 0.

The line also carries
  roofline      — the dominant kernel (the f16 MFMA GEMM, gemm_z192.hip / gemm.hip): algorithmic FLOPs of its launches
                  / their summed duration, measured with HIP events on the launch stream in an
                  instrumented pass of the same steps (events around every launch would perturb the
                  headline timing, so they are a separate pass over the same work);
  cpu_baseline  — the CPU oracle (oracle/, plain PyTorch fp32 eager = the reference's op sequence)
                  timed on this host's cores on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_TILE = 196.18          # algorithmic, ViT-B 512^2 encoder 195.34 + map_decoder 0.84
MFMA_PEAK_TFLOPS = 2500.0        # MI355X dense f16/bf16 (MI355X_MICROARCH.md)


WORKLOADS = {
    "encdec": dict(version="vit_b", patch=512, batch=16, gflop=GFLOP_PER_TILE, yaml="toponet_vitb_512_cityscale.yaml",
                   what="ViT-B encoder + map_decoder (BASELINE configs[1])"),
    "full": dict(version="vit_b", patch=512, batch=16, gflop=GFLOP_PER_TILE + 2.8, yaml="toponet_vitb_512_cityscale.yaml",
                 what="full SAMRoad.forward: encoder + map_decoder + sampler + TopoNet on 256 points / tile (BASELINE configs[2])"),
    "vith256": dict(version="vit_h", patch=256, batch=8, gflop=332.23 + 0.21, yaml="toponet_vith_256.yaml",
                    what="ViT-H encoder + map_decoder (BASELINE configs[4])"),
    # not a BASELINE config: the reference's live config/toponet_vitb_1024.yaml at its own BATCH_SIZE (937.58 + 3.355 GF per tile)
    "vitb1024": dict(version="vit_b", patch=1024, batch=4, gflop=937.58 + 3.355, yaml="toponet_vitb_1024.yaml",
                     what="ViT-B encoder (64 x 64 tokens, global attention over 4096 keys) + map_decoder"),
}


def build_workload(name, batch, rank, dev, distributed, net=None, sd=None):
    """The model, seeded random weights and resident synthetic inputs of one BASELINE workload; returns (net, state_dict, cfg, step, B, P, WL)
    where step() is one pass of the hot path over one batch.  `net` / `sd`: reuse an already built model of the same architecture."""
    from sam_road_amd import Config, SAMRoad
    WL = WORKLOADS[name]
    P = WL["patch"]
    cfg = Config(SAM_VERSION=WL["version"], PATCH_SIZE=P, TOPONET_VERSION="normal", SAM_CKPT_PATH="",
                 NO_SAM=False, USE_SAM_DECODER=False, ENCODER_LORA=False, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
    if net is None:
        net = SAMRoad(cfg)
        # seeded random-init weights of the named architecture (no checkpoint exists offline)
        g = torch.Generator().manual_seed(1234)
        sd = {}
        for k, v in net.state_dict().items():
            if rank == 0:
                if v.dim() == 1 and k.endswith("weight"):
                    t = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
                else:
                    t = 0.02 * torch.randn(v.shape, generator=g)
            else:
                t = torch.empty(v.shape)
            sd[k] = t
        if rank == 0:
            net.load_state_dict(sd, strict=True)
        net.eval().to(dev)
        if distributed:
            # rank 0 packs the weights once; the PACKED fp16 arena (~175 MB for ViT-B) goes to the other ranks device-to-device in
            # one RCCL broadcast over xGMI (SAMRoad.share_packed_weights) — before the timed region
            net.share_packed_weights(src=0)

    B = batch or WL["batch"]
    gi = torch.Generator().manual_seed(100 + rank)
    rgb = (torch.rand((B, P, P, 3), generator=gi) * 255).round().to(dev)
    step = lambda: net.infer_masks_and_img_features(rgb)
    if name == "full":
        # 256 synthetic graph points per tile (integer pixels, >= 16 px apart like NMS output), pairs by the pass-2 query
        # builder (kNN 16 within 64 px, inferencer.py:148-176)
        import numpy as np
        from sam_road_amd.inferencer import build_patch_queries, _collate
        rng = np.random.default_rng(7 + rank)
        qs = []
        for _ in range(B):
            cand = rng.integers(0, P // 16, size=(4096, 2))
            _, first = np.unique(cand[:, 0] * 64 + cand[:, 1], return_index=True)
            pts = (cand[np.sort(first)][:256] * 16 + rng.integers(0, 4, size=(256, 2))).astype(np.int64)
            qs.append(build_patch_queries(pts, 0, 0, P, P, cfg))
        pts_t = torch.as_tensor(_collate([q[1] for q in qs])).to(dev)
        pairs_t = torch.as_tensor(_collate([q[2] for q in qs])).to(dev)
        valid_t = torch.as_tensor(_collate([q[3] for q in qs])).to(dev)
        step = lambda: net(rgb, pts_t, pairs_t, valid_t)[1::2]
    step.rgb = rgb
    return net, sd, cfg, step, B, P, WL


def instrumented_rows(ctx, step, dev, nrep):
    """Per-class GPU time of `nrep` steps: HIP events around every launch on the launch stream (srh_profile_*), first step discarded."""
    ctx.profile_enable(True)
    step()                                      # first instrumented step creates the event pool: discarded
    ctx.profile_read()
    torch.cuda.synchronize(dev)
    t_i = time.perf_counter()
    for _ in range(nrep):
        step()
    torch.cuda.synchronize(dev)
    instrumented_ms = 1e3 * (time.perf_counter() - t_i) / nrep
    rows = ctx.profile_read()
    ctx.profile_enable(False)
    return rows, instrumented_ms


SIDE_WARMUP = 20         # untimed steps in front of each side workload's timed region

PER_RANK_KEYS = ("rank", "dominant_kernel_frac", "gemm_frac", "dominant_avg_launch_ms", "sustained_tiles_per_s", "smi_sclk_mhz_under_load")


def gather_per_rank(entry, distributed):
    """Every rank's own roofline figures (north_star: "rocprof-reported ... MFMA utilisation at 1/2/4/8 GPUs"): one all_gather of a small
    dict per rank after the timed region, so that the N > 1 line carries each GPU's dominant-kernel fraction and sustained clock instead of
    rank 0's alone.  Same code path on RCCL and on gloo (--plumbing-cpu, where the values are None)."""
    import torch.distributed as dist
    entry = {k: entry.get(k) for k in PER_RANK_KEYS}
    if not distributed:
        return [entry]
    allr = [None] * dist.get_world_size()
    dist.all_gather_object(allr, entry)
    return allr


def side_workloads(args, dev, rank, local_rank, world, distributed, net_b, sd_b):
    """BASELINE configs[2] (`full`) and configs[4] (`vith256`) beside the headline, bounded (a few seconds each): the same warm-up /
    barrier / max-over-ranks timing as the headline's timed region, plus rank 0's all-GEMM roofline fraction from an instrumented pass."""
    import torch.distributed as dist
    from sam_road_amd import _lib
    res = {}
    for name, steps in (("full", 40), ("vith256", 40), ("vitb1024", 20)):
        reuse = name == "full"                  # configs[2] is the headline's model with the TopoNet branch switched on
        net, _, _, step, B, P, WL = build_workload(name, 0, rank, dev, distributed, net=net_b if reuse else None, sd=sd_b if reuse else None)

        def sync_all():
            torch.cuda.synchronize(dev)
            if distributed:
                dist.barrier()
            torch.cuda.synchronize(dev)
        # 20 warm-up steps (~0.1 s): these blocks start right after CPU-only legs (cpu_baseline) and a torch.cuda.empty_cache(); with 5
        # steps some runs caught a one-off ~25 ms stall (allocator refill / clock ramp from the idle GPU) inside the 40 timed steps
        # (full: 3 191 and 3 258 tiles/s in two runs of the same build that measured 3 710-3 740 otherwise)
        for _ in range(SIDE_WARMUP):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = step()
        sync_all()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        assert all(torch.isfinite(x).all() for x in o)
        tps = world * B * steps / el
        r = {"config": f"{WL['yaml']}, batch={B} {P}x{P} tiles per GPU, {WL['what']}", "tiles_per_s": round(tps, 2),
             "ms_per_step": round(1e3 * el / steps, 4), "steps": steps, "warmup": SIDE_WARMUP, "n_gpus": world, "dtype": "f16",
             "gflop_per_tile_algorithmic": WL["gflop"],
             "whole_path_mfma_frac": round(tps / world * WL["gflop"] / 1e3 / MFMA_PEAK_TFLOPS, 4)}
        if rank == 0:
            ctx = _lib.Context.get(local_rank)
            rows, _ = instrumented_rows(ctx, step, dev, 3)
            gemm = [x for x in rows if x["name"].startswith("gemm_")]
            fl, ms, n = sum(x["flops"] for x in gemm), sum(x["ms"] for x in gemm), sum(x["launches"] for x in gemm)
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            r["gemm"] = {"achieved_tflops": round(ach, 2), "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "launches_per_step": n // 3,
                         "avg_launch_ms": round(ms / max(n, 1), 5)}
            r["by_class_ms_per_step"] = {x["name"]: round(x["ms"] / 3, 4) for x in rows}
            ag = [x for x in rows if x["name"] == "attn_global" and x["ms"] > 0]
            if ag:      # the global attention on its own (algorithmic FLOPs incl. the softmax-free 4 S^4 hd convention of SURVEY App. C)
                r["attn_global"] = {"achieved_tflops": round(sum(x["flops"] for x in ag) / (sum(x["ms"] for x in ag) * 1e-3) / 1e12, 1),
                                    "avg_launch_ms": round(sum(x["ms"] for x in ag) / sum(x["launches"] for x in ag), 5)}
        res[name] = r
        if not reuse:
            del net, step
            torch.cuda.empty_cache()
    return res


def scene_block(args, net, sd, dev, rank, world, distributed, box):
    """BASELINE configs[3]: ms per synthetic 2048 x 2048 CityScale scene (toponet_vitb_512_cityscale.yaml tiling: 256 tiles of
    512^2, INFER_BATCH_SIZE 64) through the CLI's scene loop — end to end to the edge list.  N = 1: the one-GPU pipeline
    (infer_imgs).  N > 1: every scene's tiles are SHARDED over the ranks (banded canvas reduce, point broadcast, vote gather
    over RCCL): first scene by scene, then pipelined across the ranks (inferencer._infer_imgs_tile_sharded).  `rccl_ranks` is an
    on-device all-reduce of ones: the number of ranks the collective library actually connected.  Results go into box["scene"] as
    they complete (the caller runs this under a deadline)."""
    import numpy as np
    import torch.distributed as dist
    from sam_road_amd import Config
    from sam_road_amd import inferencer as inf
    cfg = Config(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", DATASET="cityscale",
                 INFER_BATCH_SIZE=64, SAMPLE_MARGIN=64, INFER_PATCHES_PER_EDGE=16, ITSC_THRESHOLD=0.248, ROAD_THRESHOLD=0.364,
                 TOPO_THRESHOLD=0.499, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
    # the random network gets a final decoder layer that yields sparse masks (a few thousand graph points per scene, as a trained
    # network does): same architecture, same kernels, realistic host stages
    if rank == 0:
        g = torch.Generator().manual_seed(4321)
        sd2 = dict(sd)
        sd2["map_decoder.7.weight"] = 16.0 * torch.randn(sd["map_decoder.7.weight"].shape, generator=g)
        sd2["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], -2.2)
        net.load_state_dict(sd2, strict=True)
    if distributed:
        net.share_packed_weights(src=0)
    rng = np.random.default_rng(0)
    imgs = []
    for _ in range(4):
        coarse = rng.integers(0, 256, size=(2048 // 8, 2048 // 8, 3)).astype(np.float32)
        imgs.append(np.kron(coarse, np.ones((8, 8, 1), np.float32)).astype(np.uint8))
    stream = lambda n: (imgs[i % 4] for i in range(n))
    rccl_ranks = 1
    if distributed:
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        rccl_ranks = int(t.item())
    def timed(run):
        run(2, {})                                         # warm-up: staging pools, workspaces, first-call costs
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        stats = {}
        t0 = time.perf_counter()
        res = run(args.scenes, stats)
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        el = time.perf_counter() - t0
        per_rank = None
        if distributed:
            keys = ("pass1_queue_ms", "points_host_ms", "pass2_ms", "merge_host_ms", "canvas_bytes", "points_bytes", "votes_bytes")
            mine = torch.tensor([el] + [stats.get(k, 0.0) / max(args.scenes, 1) for k in keys], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            el = max(a[0].item() for a in allr)
            per_rank = [dict(zip(keys, [round(v, 3) for v in a[1:].tolist()])) for a in allr]
        return res, el, per_rank

    what = ("synthetic 2048x2048 u8 CityScale-sized scenes, toponet_vitb_512_cityscale.yaml tiling (256 tiles of 512^2, batch 64), "
            "pass 1 + graph points + pass 2 + edge vote, end to end (BASELINE configs[3])")
    if not distributed:
        res, el, _ = timed(lambda n, st: list(inf.infer_imgs(net, stream(n), cfg, dev, tile_sharded=False)))
        last = res[-1]
        box["scene"] = {"what": what, "mode": "one GPU, scenes software-pipelined (infer_imgs)", "scenes": args.scenes,
                        "ms_per_scene": round(1e3 * el / args.scenes, 3), "scenes_per_s": round(args.scenes / el, 3), "n_gpus": world,
                        "rccl_ranks": rccl_ranks, "graph_points": int(last[0].shape[0]), "edges": int(last[1].shape[0])}
        return
    # N > 1, two loops over the same scenes.  First the SERIAL tile-sharded loop (infer_one_img scene by scene: the library's default
    # under torch.distributed, its three exchange steps strictly in sequence) — its figure is in the box before the second loop starts,
    # so a hang there cannot cost it.  Then the PIPELINED loop (opt-in in the library: it has only ever run on gloo), whose figure
    # replaces `ms_per_scene` when it completes; both are reported.
    res, el, _ = timed(lambda n, st: list(inf.infer_imgs(net, stream(n), cfg, dev, tile_sharded=True, pipelined=False)))
    out = None
    if rank == 0:
        last = res[-1]
        out = {"what": what, "mode": f"tiles of every scene sharded over {world} ranks, scene by scene (serial loop)", "scenes": args.scenes,
               "ms_per_scene": round(1e3 * el / args.scenes, 3), "scenes_per_s": round(args.scenes / el, 3), "n_gpus": world,
               "rccl_ranks": rccl_ranks, "serial_loop_ms_per_scene": round(1e3 * el / args.scenes, 3),
               "graph_points": int(last[0].shape[0]), "edges": int(last[1].shape[0]),
               "pipelined_loop": "did not complete (see error / deadline)"}
        box["scene"] = out
    res, el, per_rank = timed(lambda n, st: list(inf._infer_imgs_tile_sharded(net, stream(n), cfg, dev, stats=st)))
    if rank == 0:
        last = res[-1]
        same = int(last[0].shape[0]) == out["graph_points"] and int(last[1].shape[0]) == out["edges"]
        out = dict(out, mode=f"tiles of every scene sharded over {world} ranks, scenes pipelined across the ranks",
                   ms_per_scene=round(1e3 * el / args.scenes, 3), scenes_per_s=round(args.scenes / el, 3),
                   pipelined_loop="completed; same graph as the serial loop" if same else "completed; GRAPH DIFFERS from the serial loop",
                   pipelined_loop_ms_per_scene=round(1e3 * el / args.scenes, 3), per_rank_per_scene=per_rank,
                   bytes_per_scene_note="each rank's OWN traffic: canvas = its band of the two f32 canvases shipped to rank 0 (rank 0: all it "
                                        "receives); points = int64 [N,2] broadcast (rank 0: sent to every peer); votes = its (key, sum, count, first) rows of the gather")
        box["scene"] = out


def plumbing_cpu(args):
    """The launch contract without a GPU (gloo): RANK / WORLD_SIZE / MASTER_* from the environment, an all-reduce of ones, the
    pipelined tile-sharded scene loop on the tests' CPU stand-in, per-rank statistics gathered on rank 0, ONE JSON line.  The line
    carries no metric value — it is a plumbing check, not a benchmark."""
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_distributed_cpu as T
    from oracle.synth import synth_scene
    from sam_road_amd import Config
    from sam_road_amd import inferencer as inf
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.set_num_threads(1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    cfg = dict(T._E2E_CFG, SAMPLE_MARGIN=0, INFER_PATCHES_PER_EDGE=2)
    net = T._CpuStandIn(cfg)
    imgs = [synth_scene(512, seed=s) for s in (6, 9)]
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)
    stats = {}
    t0 = time.perf_counter()
    res = list(inf._infer_imgs_tile_sharded(net, iter(imgs), Config(cfg), "cpu", stats=stats)) if world > 1 else \
        [inf.infer_one_img(net, im, Config(cfg), device="cpu") for im in imgs]
    el = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        mine = torch.tensor([el, stats.get("canvas_bytes", 0.0), stats.get("points_bytes", 0.0), stats.get("votes_bytes", 0.0)], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(v, 3) for v in a.tolist()] for a in allr]
    per_rank_roofline = gather_per_rank({"rank": rank}, world > 1)      # the N > 1 line's per-rank block (values None: nothing is measured here)
    if rank == 0:
        print(json.dumps({"metric": "plumbing check (no measurement)", "value": None, "plumbing_only": True, "n_gpus": world,
                          "collective_ranks": int(ones.item()), "scenes": len(res), "graph_points": [int(r[0].shape[0]) for r in res],
                          "edges": [int(r[1].shape[0]) for r in res], "per_rank": per_rank, "per_rank_roofline": per_rank_roofline}))
    if world > 1:
        dist.destroy_process_group()


def _respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-exec the same command line under
    torch.distributed.run (one rank per GPU), so that a bare invocation can never print a one-rank number as the N-GPU point."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="tiles per step per GPU (default: the workload's BASELINE batch)")
    ap.add_argument("--workload", default="encdec", choices=["encdec", "full", "vith256", "vitb1024"],
                    help="encdec: BASELINE configs[1] (headline: ViT-B 512^2 B=16, encoder + map_decoder); "
                         "full: configs[2] (the same + sampler + TopoNet, 256 points per tile = SAMRoad.forward); "
                         "vith256: configs[4] (toponet_vith_256.yaml, ViT-H 256^2 tiles, B=8, encoder + map_decoder)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-gpu", action="store_true", help="skip the bounded reference-PyTorch-on-this-GPU leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="tuning aid: skip the finite-output check (kernel ablations)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 3 s sustained leg (sustained_tiles_per_s, SMI clock)")
    ap.add_argument("--no-scene", action="store_true", help="skip the scene block (ms per 2048^2 CityScale scene; tile-sharded over the ranks when N > 1)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the bounded `workloads` block (BASELINE configs[2] full SAMRoad.forward and configs[4] ViT-H 256^2 beside the headline)")
    ap.add_argument("--scenes", type=int, default=16, help="scenes of the scene block's timed stream")
    ap.add_argument("--scene-timeout", type=float, default=240.0, help="deadline of the scene block in seconds (the line is printed without it afterwards)")
    ap.add_argument("--plumbing-cpu", action="store_true",
                    help="NO measurement: run the N-rank launch contract and the tile-sharded scene block on CPU / gloo with the tests' "
                         "oracle stand-in model (tests/test_distributed_cpu.py) — checks env handling, collectives and the JSON line without a GPU")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_respawn_under_launcher(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} rank(s); launch with\n"
                 f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...")
    if args.plumbing_cpu:
        return plumbing_cpu(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    # host threads: this process's share of the CPUs the container may use (cgroup quota / affinity, sam_road_amd/hostcpu.py) —
    # torch's default of one OpenMP thread per machine core gets a container throttled during weight initialisation
    from sam_road_amd.hostcpu import usable_cpus
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // world)))
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from sam_road_amd import _lib

    rccl_ranks = 1
    if distributed:
        # the number of ranks the collective library actually connected (an on-device all-reduce of ones): the line is only printed
        # as an N-GPU line when this equals N
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        rccl_ranks = int(t.item())
        assert rccl_ranks == world == args.gpus, f"--gpus {args.gpus}, WORLD_SIZE {world}, but the collective spans {rccl_ranks} rank(s)"

    net, sd, cfg, step, B, P, WL = build_workload(args.workload, args.batch, rank, dev, distributed)
    rgb = step.rgb

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scores, emb = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert args.no_check or (torch.isfinite(scores).all() and torch.isfinite(emb).all())
    tiles_per_s = world * B * args.steps / elapsed

    # Sustained leg: the timed region above is a ~0.1 s burst from an idle (cool) chip; the matrix pipes of this part are power-limited
    # (1.55 GHz under random-data f16 MFMA), so the same step is run for >= 3 s and reported beside `value`, with
    # the shader clock rocm-smi shows while the queue is full.
    sustained = None
    if not args.no_sustained:
        import re
        import subprocess
        sync_all()
        t0 = time.perf_counter()
        n_s, smi_clock = 0, None
        for _ in range(160):
            step()
            n_s += 1
        if rank == 0 or distributed:
            try:        # the queue holds ~0.7 s of work: the reading is taken under load (every rank reads its own GPU's clock)
                txt = subprocess.run(["rocm-smi", "--showclocks", "-d", str(local_rank)], capture_output=True, text=True, timeout=20).stdout
                m = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", txt)
                smi_clock = int(m.group(1)) if m else None
            except Exception:
                smi_clock = None
        while True:
            for _ in range(40):
                step()
                n_s += 1
            torch.cuda.synchronize(dev)
            if time.perf_counter() - t0 >= 3.0:
                break
        sync_all()
        el_s = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el_s, float(n_s)], dtype=torch.float64, device=dev)
            tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            sustained = {"tiles_per_s": round(B * tsum[1].item() / tmax[0].item(), 3), "seconds": round(tmax[0].item(), 3)}
        else:
            sustained = {"tiles_per_s": round(B * n_s / el_s, 3), "seconds": round(el_s, 3)}
        sustained["steps_per_gpu"] = n_s
        sustained["smi_sclk_mhz_under_load"] = smi_clock
        sustained["this_rank_tiles_per_s"] = round(B * n_s / el_s, 3)

    out = {
        "metric": "tiles/sec (512x512 ViT-B, SAMRoad.infer_masks_and_img_features: encoder + mask decoder)" if args.workload == "encdec"
                  else f"tiles/sec ({P}x{P} {WL['version']}, {WL['what']})",
        "value": round(tiles_per_s, 3), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic", "rccl_ranks": rccl_ranks,
        "config": {"workload": f"{WL['yaml']}, batch={B} {P}x{P} tiles per GPU, {WL['what']}",
                   "tiles_per_step_per_gpu": B, "patch": P, "input": "f32 NHWC resident in HBM",
                   "weights": "seeded random init", "parallelism": f"tile-dp{world}",
                   "gflop_per_tile_algorithmic": WL["gflop"],
                   "whole_path_mfma_frac": round(tiles_per_s / world * WL["gflop"] / 1e3 / MFMA_PEAK_TFLOPS, 4)},
    }

    if sustained is not None:
        out["sustained_tiles_per_s"] = sustained["tiles_per_s"]
        out["sustained"] = sustained
    out["build_id"] = _lib.build_id()               # sha256 of the sources libsamroad_hip.so was compiled from (sam_road_amd/build.py)

    mine = {"rank": rank, "sustained_tiles_per_s": sustained.get("this_rank_tiles_per_s") if sustained else None,
            "smi_sclk_mhz_under_load": sustained.get("smi_sclk_mhz_under_load") if sustained else None}
    if (rank == 0 or distributed) and not args.no_roofline:
        ctx = _lib.Context.get(local_rank)
        stream = torch.cuda.current_stream(dev).cuda_stream
        torch.cuda.synchronize(dev)
        # Per-class GPU time: HIP events around every launch on the launch stream, in a separate pass over the same steps (the first
        # instrumented step creates the event pool and is discarded).  Two facts about these numbers, both checked against
        # `rocprofv3 --kernel-trace --stats` of this command (profiles/r03_event_overhead_check.txt):
        #  * the event-to-event time of a launch IS its rocprofv3 duration to within ~1 % — the figures below are reported as measured;
        #  * the instrumented pass (like a profiled run) is slower than the timed region above: the marker packets serialise the
        #    queue, so the classes sum to the instrumented pass's own step time (`instrumented_ms_per_step`), not to `ms_per_step`.
        # `event_overhead_us_per_launch` (srh_profile_overhead: event pair around a kernel of known duration) is what subtracting
        # the packet handling would remove; `achieved_if_overhead_subtracted` shows the effect on the GEMM figure.
        ovh = ctx.profile_overhead(stream)
        nrep = max(1, min(args.steps, 5))
        rows, instrumented_ms = instrumented_rows(ctx, step, dev, nrep)

        def agg(sel):
            fl, ms, n = sum(r["flops"] for r in sel), sum(r["ms"] for r in sel), sum(r["launches"] for r in sel)
            return fl, ms, n, (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        gemm = [r for r in rows if r["name"].startswith("gemm_")]
        fl, ms, n, ach = agg(gemm)
        total_ms = sum(r["ms"] for r in rows)
        # the dominant kernel by GPU time: the four big linear layers of every block (one kernel template)
        block_gemm = [r for r in gemm if r["name"] in ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2")]
        d_fl, d_ms, d_n, d_ach = agg(block_gemm)
        mine.update(dominant_kernel_frac=round(d_ach / MFMA_PEAK_TFLOPS, 4), gemm_frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                    dominant_avg_launch_ms=round(d_ms / max(d_n, 1), 5))
        uses_z192 = WL["version"] == "vit_b" and B * (P // 16) ** 2 >= 8192
        # HBM bytes per GEMM launch: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command and workload
        # (tools/profile_gpu.sh + tools/summarize_profile.py -> profiles/<tag>_hbm_traffic[_<workload>].json).  A summary is used
        # ONLY if it was measured on this very build (it carries the library's build id); otherwise traffic is null
        traffic, traffic_src, traffic_note = None, None, "no PMC summary of this workload under profiles/"
        import glob
        suffix = "_hbm_traffic.json" if args.workload == "encdec" else f"_hbm_traffic_{args.workload}.json"
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), reverse=True):
            try:
                js = json.load(open(path))
            except Exception:
                continue
            if js.get("_build_id") != out["build_id"] or B != WL["batch"]:
                traffic_note = f"newest PMC summary ({os.path.relpath(path, ROOT)}) was measured on build {js.get('_build_id')}, not this one"
                continue
            traffic, traffic_src, traffic_note = js["_gemm_all"]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), None
            break
        big = max(gemm, key=lambda r: r["ms"])["name"] if gemm else "none"
        kname = ("srh::gemm_z192_kernel (hand-scheduled persistent 256x192 f16 MFMA GEMM, one wave per SIMD, deferred epilogue: qkv / proj / fc1 / fc2) + small-layer GEMMs"
                 if uses_z192 else
                 "srh::gemm_pp_kernel / gemm_r320_kernel / gemm_ring_kernel (LDS-DMA f16 MFMA GEMMs: 128x256 ping-pong incl. split-K slices folded by the next LayerNorm, 128x320, 128x128 ring; small M, where the z192 tile does not fill the chip)")
        out["roofline"] = {"bound": "mfma", "kernel": kname, "largest_class": big,
                           "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes/launch",
                           "traffic_source": traffic_src, "algorithmic_flops_per_launch": round(fl / max(n, 1), 1),
                           "launches": n, "avg_launch_ms": round(ms / max(n, 1), 5),
                           "dominant_kernel": {"what": "the block GEMMs alone (qkv, proj, fc1, fc2: one kernel template, " +
                                                       ("gemm_z192_kernel" if uses_z192 else "gemm_glds*_kernel") + ")",
                                               "achieved": round(d_ach, 2), "frac": round(d_ach / MFMA_PEAK_TFLOPS, 4),
                                               "launches": d_n, "avg_launch_ms": round(d_ms / max(d_n, 1), 5),
                                               "algorithmic_flops_per_launch": round(d_fl / max(d_n, 1), 1)},
                           "share_of_gpu_time": round(ms / total_ms, 4) if total_ms else None,
                           "timing": "HIP events around every launch on the launch stream (separate instrumented pass; agrees with rocprofv3 "
                                     "--kernel-trace durations to ~1 %)",
                           "event_overhead_us_per_launch": round(ovh * 1e3, 3),
                           "achieved_if_overhead_subtracted": round(fl / (max(ms - n * ovh, 1e-9) * 1e-3) / 1e12, 2) if ms > 0 else None,
                           "instrumented_ms_per_step": round(instrumented_ms, 4),
                           "sum_of_classes_ms_per_step": round(total_ms / nrep, 4),
                           "by_class_ms_per_step": {r["name"]: round(r["ms"] / nrep, 4) for r in rows}}
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
    if distributed:
        out["per_rank_roofline"] = gather_per_rank(mine, True)       # every rank's figures, not rank 0's alone

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded CPU sample: the oracle (reference op sequence, eager fp32) on 2 tiles, 1 warm-up + 3 timed
        from oracle.samroad import AttrDict, SAMRoadOracle
        oracle = SAMRoadOracle(AttrDict(cfg)).eval()
        oracle.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
        assert args.workload == "encdec", "the CPU baseline leg is defined for the headline workload (use --no-cpu-baseline)"
        # one thread per PHYSICAL core (torch's default = logical CPUs oversubscribes the FP units: measured slower)
        try:
            import psutil
            phys = psutil.cpu_count(logical=False) or os.cpu_count()
        except Exception:
            phys = os.cpu_count()
        from sam_road_amd.hostcpu import usable_cpus
        usable = usable_cpus()                  # affinity mask and cgroup CPU quota: more threads than that only get throttled
        threads = max(1, min(int(phys), 64, usable))
        prev_threads = torch.get_num_threads()
        torch.set_num_threads(threads)
        nt = 4
        x = rgb[:nt].cpu()
        oracle.infer_masks_and_img_features(x)
        t0 = time.perf_counter()
        iters = 0
        while iters < 3 or (time.perf_counter() - t0 < 10.0 and iters < 12):
            oracle.infer_masks_and_img_features(x)
            iters += 1
        dt = time.perf_counter() - t0
        torch.set_num_threads(prev_threads)
        out["cpu_baseline"] = {"value": round(nt * iters / dt, 3), "unit": "tiles/s", "cores": threads,
                               "physical_cores": int(phys), "logical_cpus": os.cpu_count(), "usable_cpus": usable,
                               "kind": "port", "sample": f"oracle (plain PyTorch fp32 eager = the reference's op sequence, inferencer.py --device cpu) "
                                                         f"infer_masks_and_img_features on {nt} of the {B} tiles, 1 warm-up + {iters} timed "
                                                         f"iterations, {threads} threads"}

    if rank == 0 and world == 1 and not args.no_reference_gpu and args.workload == "encdec":
        # bounded leg: "the reference PyTorch path" of BASELINE.md §3 C2(ii)/(iii) on THIS GPU — the oracle (same eager op
        # sequence as model.py on stock PyTorch-ROCm: rocBLAS / MIOpen kernels), fp32 as the reference runs it and under fp16
        # autocast (what its training used) as the stronger baseline.  1 warm-up + 3 timed steps of the same B tiles each.
        from oracle.samroad import AttrDict, SAMRoadOracle
        ref = SAMRoadOracle(AttrDict(cfg)).eval()
        ref.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
        ref.to(dev)
        legs = {}
        for name, ctxmgr in (("fp32_eager", None), ("fp16_autocast", torch.autocast("cuda", dtype=torch.float16))):
            def run():
                if ctxmgr is None:
                    return ref.infer_masks_and_img_features(rgb)
                with ctxmgr:
                    return ref.infer_masks_and_img_features(rgb)
            run()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                run()
            torch.cuda.synchronize(dev)
            legs[name] = round(3 * B / (time.perf_counter() - t0), 2)
        del ref
        torch.cuda.empty_cache()
        out["reference_gpu"] = {"unit": "tiles/s", **legs, "what": "oracle (reference op sequence) on stock PyTorch-ROCm eager, same GPU, "
                                f"same {B} resident tiles, 1 warm-up + 3 timed steps per leg"}
        # BASELINE.md §5 records this leg as the baseline number of the >= 4x target (no number is published by the reference)
        out["vs_baseline"] = round(tiles_per_s / legs["fp32_eager"], 3)
        out["vs_baseline_def"] = "value / reference_gpu.fp32_eager (reference PyTorch path on the same MI355X, BASELINE.md §3 C2(ii), §5)"

    if not args.no_workloads and args.workload == "encdec" and not args.batch:
        # BASELINE configs[2] and configs[4] beside the headline, so that the driver's own run of the default command times them
        wl = side_workloads(args, dev, rank, local_rank, world, distributed, net, sd)
        if rank == 0:
            out["workloads"] = wl

    if not args.no_scene and args.workload == "encdec":
        # The scene block must never cost the tiles/s line.  On N > 1 its exchange steps (banded point-to-point canvas reduce, point
        # broadcast, vote gather) have only ever run on gloo (tests/test_distributed_cpu.py) — gpurun exposes one GPU — so it runs on a
        # worker thread under a deadline: on an exception or a hang (a rank that failed leaves the others inside a collective) rank 0
        # still prints the line, with the reason in place of the scene figures, and every rank leaves without the collective teardown.
        import threading
        box = {}

        def _scene():
            try:
                torch.cuda.set_device(dev)          # the current device is per thread
                scene_block(args, net, sd, dev, rank, world, distributed, box)
            except Exception as e:      # noqa: BLE001 — reported in the line
                box["error"] = f"{type(e).__name__}: {e}"[:400]
        th = threading.Thread(target=_scene, daemon=True)
        th.start()
        th.join(timeout=args.scene_timeout)
        hung = th.is_alive()
        if rank == 0:
            why = box.get("error") or (f"no result after {args.scene_timeout:.0f} s (hang in an exchange step?)" if hung else None)
            out["scene"] = dict(box.get("scene") or {"n_gpus": world}, **({"error": why} if why else {}))
        if hung or "error" in box:
            if rank == 0:
                print(json.dumps(out), flush=True)
            else:
                time.sleep(2.0)         # let rank 0's line out before the launcher sees a rank leave
            os._exit(0)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
  roofline      — the dominant kernel (the f16 MFMA GEMM, gemm_q192.hip / gemm.hip): algorithmic FLOPs of its launches
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="tiles per step per GPU (default: the workload's BASELINE batch)")
    ap.add_argument("--workload", default="encdec", choices=["encdec", "full", "vith256"],
                    help="encdec: BASELINE configs[1] (headline: ViT-B 512^2 B=16, encoder + map_decoder); "
                         "full: configs[2] (the same + sampler + TopoNet, 256 points per tile = SAMRoad.forward); "
                         "vith256: configs[4] (toponet_vith_256.yaml, ViT-H 256^2 tiles, B=8, encoder + map_decoder)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-gpu", action="store_true", help="skip the bounded reference-PyTorch-on-this-GPU leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="tuning aid: skip the finite-output check (kernel ablations)")
    args = ap.parse_args()
    warnings.simplefilter("ignore")

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

    from sam_road_amd import Config, SAMRoad
    from sam_road_amd import _lib

    WL = {"encdec": dict(version="vit_b", patch=512, batch=16, gflop=GFLOP_PER_TILE, yaml="toponet_vitb_512_cityscale.yaml",
                         what="ViT-B encoder + map_decoder (BASELINE configs[1])"),
          "full": dict(version="vit_b", patch=512, batch=16, gflop=GFLOP_PER_TILE + 2.8, yaml="toponet_vitb_512_cityscale.yaml",
                       what="full SAMRoad.forward: encoder + map_decoder + sampler + TopoNet on 256 points / tile (BASELINE configs[2])"),
          "vith256": dict(version="vit_h", patch=256, batch=8, gflop=332.23 + 0.21, yaml="toponet_vith_256.yaml",
                          what="ViT-H encoder + map_decoder (BASELINE configs[4])")}[args.workload]
    P = WL["patch"]
    cfg = Config(SAM_VERSION=WL["version"], PATCH_SIZE=P, TOPONET_VERSION="normal", SAM_CKPT_PATH="",
                 NO_SAM=False, USE_SAM_DECODER=False, ENCODER_LORA=False, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
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

    B = args.batch or WL["batch"]
    gi = torch.Generator().manual_seed(100 + rank)
    rgb = (torch.rand((B, P, P, 3), generator=gi) * 255).round().to(dev)
    step = lambda: net.infer_masks_and_img_features(rgb)
    if args.workload == "full":
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

    out = {
        "metric": "tiles/sec (512x512 ViT-B, SAMRoad.infer_masks_and_img_features: encoder + mask decoder)" if args.workload == "encdec"
                  else f"tiles/sec ({P}x{P} {WL['version']}, {WL['what']})",
        "value": round(tiles_per_s, 3), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{WL['yaml']}, batch={B} {P}x{P} tiles per GPU, {WL['what']}",
                   "tiles_per_step_per_gpu": B, "patch": P, "input": "f32 NHWC resident in HBM",
                   "weights": "seeded random init", "parallelism": f"tile-dp{world}",
                   "gflop_per_tile_algorithmic": WL["gflop"],
                   "whole_path_mfma_frac": round(tiles_per_s / world * WL["gflop"] / 1e3 / MFMA_PEAK_TFLOPS, 4)},
    }

    if rank == 0:
        out["build_id"] = _lib.build_id()           # sha256 of the sources libsamroad_hip.so was compiled from (sam_road_amd/build.py)

    if rank == 0 and not args.no_roofline:
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

        def agg(sel):
            fl, ms, n = sum(r["flops"] for r in sel), sum(r["ms"] for r in sel), sum(r["launches"] for r in sel)
            return fl, ms, n, (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        gemm = [r for r in rows if r["name"].startswith("gemm_")]
        fl, ms, n, ach = agg(gemm)
        total_ms = sum(r["ms"] for r in rows)
        # the dominant kernel by GPU time: the four big linear layers of every block (one kernel template)
        block_gemm = [r for r in gemm if r["name"] in ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2")]
        d_fl, d_ms, d_n, d_ach = agg(block_gemm)
        uses_q192 = WL["version"] == "vit_b" and B * (P // 16) ** 2 >= 8192
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
        kname = ("srh::gemm_q192_kernel (persistent 256x192 f16 MFMA GEMM with deferred epilogue: qkv / proj / fc1 / fc2) + small-layer GEMMs"
                 if uses_q192 else
                 "srh::gemm_glds_kernel / gemm_glds256_kernel (LDS-DMA 128x128 split-K and 256x256 f16 MFMA GEMMs: N, K not multiples of the q192 tile)")
        out["roofline"] = {"bound": "mfma", "kernel": kname, "largest_class": big,
                           "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes/launch",
                           "traffic_source": traffic_src, "algorithmic_flops_per_launch": round(fl / max(n, 1), 1),
                           "launches": n, "avg_launch_ms": round(ms / max(n, 1), 5),
                           "dominant_kernel": {"what": "the block GEMMs alone (qkv, proj, fc1, fc2: one kernel template, " +
                                                       ("gemm_q192_kernel" if uses_q192 else "gemm_glds*_kernel") + ")",
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

    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

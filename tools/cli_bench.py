#!/usr/bin/env python
"""End-to-end timing of the drop-in CLI (python -m sam_road_amd.inferencer) on a fake CityScale checkout: 27 test scenes of
2048x2048 RGB PNGs (synthetic texture), toponet_vitb_512_cityscale-style config, seeded random weights (final map_decoder bias
lowered as in tools/scene_bench.py so that a scene yields a few thousand graph points).  Reports wall seconds of main() per scene —
PNG decoding, the software-pipelined scene loop, PNG / pickle writing — next to the loop's own inference_time.txt.

    python tools/cli_bench.py [--scenes 27] [--keep DIR]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=27)
    ap.add_argument("--keep", default=None)
    args = ap.parse_args()
    from PIL import Image
    from sam_road_amd import Config, SAMRoad
    from sam_road_amd import inferencer as inf
    work = args.keep or tempfile.mkdtemp(prefix="srh_cli_")
    os.makedirs(os.path.join(work, "cityscale", "20cities"), exist_ok=True)
    cfg = dict(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", DATASET="cityscale",
               INFER_BATCH_SIZE=64, SAMPLE_MARGIN=64, INFER_PATCHES_PER_EDGE=16, ITSC_THRESHOLD=0.248, ROAD_THRESHOLD=0.364,
               TOPO_THRESHOLD=0.499, ITSC_NMS_RADIUS=8, ROAD_NMS_RADIUS=16, NEIGHBOR_RADIUS=64, MAX_NEIGHBOR_QUERIES=16)
    with open(os.path.join(work, "cfg.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    ids = inf.cityscale_data_partition()[2][:args.scenes]
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    for i in ids:
        coarse = rng.integers(0, 256, size=(256, 256, 3)).astype(np.int16)
        img = np.clip(np.kron(coarse, np.ones((8, 8, 1), np.int16)) + rng.integers(-12, 12, size=(2048, 2048, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(work, "cityscale", "20cities", f"region_{i}_sat.png"), compress_level=1)
    net = SAMRoad(Config(cfg))
    g = torch.Generator().manual_seed(1234)
    sd = {}
    for k, v in net.state_dict().items():
        sd[k] = (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if (v.dim() == 1 and k.endswith("weight")) else 0.02 * torch.randn(v.shape, generator=g)
    sd["map_decoder.7.weight"] = 16.0 * torch.randn(sd["map_decoder.7.weight"].shape, generator=g)
    sd["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], -2.2)
    torch.save({"state_dict": sd}, os.path.join(work, "ckpt.ckpt"))
    print(f"fake dataset ({len(ids)} scenes) + checkpoint written in {time.perf_counter() - t0:.1f} s", flush=True)
    # only the partition is overridden (27 regions would need 180 files otherwise): the CLI enumerates, loads and writes as usual
    inf.cityscale_data_partition = lambda: ([], [], ids)
    os.chdir(work)
    import warnings
    warnings.simplefilter("ignore")
    build = inf._build_net
    t_build = [0.0]
    def timed_build(*a, **k):
        t = time.perf_counter(); n = build(*a, **k)
        n.infer_masks_and_img_features(torch.zeros((1, 512, 512, 3), device="cuda"))     # pack the weights outside the timed loop
        torch.cuda.synchronize(); t_build[0] = time.perf_counter() - t
        return n
    inf._build_net = timed_build
    t0 = time.perf_counter()
    inf.main(["--config", "cfg.yaml", "--checkpoint", "ckpt.ckpt", "--output_dir", "bench"])
    wall = time.perf_counter() - t0 - t_build[0]
    txt = open(os.path.join(work, "save", "bench", "inference_time.txt")).read()
    import json
    print(json.dumps({"scenes": len(ids), "cli_wall_s_without_model_load": round(wall, 3), "ms_per_scene_end_to_end": round(1e3 * wall / len(ids), 1),
                      "model_load_and_pack_s": round(t_build[0], 2), "inference_time_txt": txt,
                      "outputs": len(os.listdir(os.path.join(work, "save", "bench", "mask"))) + len(os.listdir(os.path.join(work, "save", "bench", "graph")))}))


if __name__ == "__main__":
    main()

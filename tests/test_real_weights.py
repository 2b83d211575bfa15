"""Real-weights hook.  No trained checkpoint or imagery exists in the build environment, so every
tolerance in tests/tolerances.py was established on synthetic weights.  This test turns the FIRST deployment with real weights
into a parity run: point it at a `congrui/sam_road` Lightning checkpoint and a CityScale / SpaceNet scene and it compares the HIP
path with the oracle on real tiles and prints the measured errors next to the stated tolerances.

    SRH_REAL_CKPT=/data/cityscale_vitb_512_e10.ckpt \\
    SRH_REAL_CONFIG=/path/to/sam_road/config/toponet_vitb_512_cityscale.yaml \\
    SRH_REAL_SCENE=/data/cityscale/20cities/region_8_sat.png \\
    python -m pytest tests/test_real_weights.py -m gpu -s

Skips (not fails) when the variables are not set.  Reference: README.md:28-51 (checkpoints), inferencer.py:246-254 (how a
checkpoint is loaded: ckpt["state_dict"], strict), inferencer.py:52-58 (tiles are f32 copies of the u8 scene crop)."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tolerances as T

CKPT, SCENE, CONFIG = (os.environ.get(k) for k in ("SRH_REAL_CKPT", "SRH_REAL_SCENE", "SRH_REAL_CONFIG"))


@pytest.mark.skipif(not (CKPT and SCENE and CONFIG), reason="set SRH_REAL_CKPT, SRH_REAL_CONFIG and SRH_REAL_SCENE to run on real weights")
def test_real_checkpoint_tiles_hip_vs_oracle():
    from oracle import scene as oscene
    from oracle.samroad import AttrDict, SAMRoadOracle
    from sam_road_amd import SAMRoad
    from sam_road_amd.config import load_config
    from sam_road_amd.inferencer import read_rgb_img
    warnings.simplefilter("ignore")
    cfg = load_config(CONFIG)
    cfg["SAM_CKPT_PATH"] = ""                                    # the trained checkpoint carries every weight (model.py:365 is training-time only)
    sd = torch.load(CKPT, map_location="cpu")["state_dict"]
    oracle = SAMRoadOracle(AttrDict(cfg)).eval()
    oracle.load_state_dict(sd, strict=True)
    net = SAMRoad(cfg)
    net.load_state_dict(sd, strict=True)
    net.eval().to("cuda")
    img = np.load(SCENE) if SCENE.endswith(".npy") else read_rgb_img(SCENE)
    P = int(cfg.PATCH_SIZE)
    infos = oscene.get_patch_info_one_img(0, img.shape[0], cfg.SAMPLE_MARGIN, P, cfg.INFER_PATCHES_PER_EDGE)
    n = int(os.environ.get("SRH_REAL_TILES", "4"))
    pick = [infos[i] for i in np.linspace(0, len(infos) - 1, n).round().astype(int)]
    rgb = oscene.get_batch_img_patches(img, pick)
    ms_r, e_r = oracle.infer_masks_and_img_features(rgb)
    ms, e = net.infer_masks_and_img_features(rgb.cuda())
    ms, e = ms.cpu(), e.float().cpu()
    rel = ((e - e_r).norm() / e_r.norm()).item()
    lv = ((ms * 255).to(torch.uint8).int() - (ms_r * 255).to(torch.uint8).int()).abs()
    print(f"[real weights] {os.path.basename(CKPT)} on {n} tiles of {os.path.basename(SCENE)}: |x|max of the embeddings {e_r.abs().max().item():.2f}")
    T.check("real_ckpt_emb_rel_l2", rel, T.EMB_REL_L2)
    T.check("real_ckpt_emb_max_abs", (e - e_r).abs().max().item(), T.EMB_MAX_ABS * max(1.0, e_r.abs().max().item() / 4.0))
    T.check("real_ckpt_mask_score", (ms - ms_r).abs().max().item(), T.MASK_SCORE)
    assert lv.max().item() <= 2
    T.check("real_ckpt_u8_within1", (lv <= 1).float().mean().item(), T.U8_WITHIN1, at_least=True)
    # pass 2 on the points the oracle masks give for these tiles (the reference's own thresholds from the YAML)
    from oracle.synth import synth_queries
    points, pairs, valid = synth_queries(n, 128, P, seed=5)
    ts_r = oracle.infer_toponet(e_r, points, pairs, valid)
    ts = net.infer_toponet(e.cuda(), points.cuda(), pairs.cuda(), valid.cuda()).cpu()
    v = valid.bool()
    T.check("real_ckpt_topo_score", (ts[..., 0][v] - ts_r[..., 0][v]).abs().max().item(), T.TOPO_SCORE)
    agree = ((ts[..., 0][v] > cfg.TOPO_THRESHOLD) == (ts_r[..., 0][v] > cfg.TOPO_THRESHOLD)).float().mean().item()
    T.check("real_ckpt_topo_decisions", agree, T.TOPO_DECISIONS, at_least=True)

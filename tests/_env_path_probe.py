"""Helper of tests/test_gpu_env_paths.py: runs one parity case in THIS process (the library reads its SRH_* switches once, at
first use, so every switch needs a fresh interpreter) and prints a JSON line with the errors against the oracle."""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle.samroad import AttrDict, SAMRoadOracle  # noqa: E402
from oracle.synth import synth_state_dict, synth_tiles  # noqa: E402
from sam_road_amd import Config, SAMRoad  # noqa: E402

CASES = {   # name -> (config, batch)
    "vitb512_b8": (dict(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="", ENCODER_DEPTH=2,
                        ENCODER_GLOBAL_ATTN_INDEXES=[1]), 8),          # 8192 tokens: the q192 GEMMs when enabled
    "vith256_b2": (dict(SAM_VERSION="vit_h", PATCH_SIZE=256, TOPONET_VERSION="normal", SAM_CKPT_PATH="", ENCODER_DEPTH=2,
                        ENCODER_GLOBAL_ATTN_INDEXES=[1]), 2),          # head dim 80 attention, split-K GEMMs
}


def main():
    warnings.simplefilter("ignore")
    cfg, B = CASES[sys.argv[1]]
    oracle = SAMRoadOracle(AttrDict(cfg)).eval()
    sd = synth_state_dict(oracle, 1234)
    oracle.load_state_dict(sd, strict=True)
    net = SAMRoad(Config(cfg))
    net.load_state_dict(sd, strict=True)
    net.eval().to("cuda")
    rgb = synth_tiles(B, cfg["PATCH_SIZE"], seed=3)
    s_ref, e_ref = oracle.infer_masks_and_img_features(rgb)
    s, e = net.infer_masks_and_img_features(rgb.cuda())
    e, s = e.cpu(), s.cpu()
    print(json.dumps({"emb_rel_l2": ((e - e_ref).norm() / e_ref.norm()).item(), "score_max_abs": (s - s_ref).abs().max().item(),
                      "finite": bool(torch.isfinite(e).all() and torch.isfinite(s).all())}))


if __name__ == "__main__":
    main()

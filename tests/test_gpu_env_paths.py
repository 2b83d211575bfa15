"""The functional alternates the library can be switched to by environment variable are product code: each one is run against
the oracle here, in a fresh interpreter (the switches are read once per process).
    SRH_GEMM_Q192=0    every big layer through the 128x128 / 256x256 LDS-DMA GEMMs with the residual add in the GEMM epilogue
    SRH_GEMM_SPLITK=0  small-M layers (ViT-H / ViT-L at 256 px) without the deterministic split-K
    SRH_ATTN_HDX=0     head dim 80 (ViT-H) through the generic f32 attention kernel instead of attention_hdx.hip
(Ablation / tuning switches that change results exist only in probe builds: -DSRH_TUNING, tools/probes/build_probes.sh.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env,case", [({}, "vitb512_b8"), ({"SRH_GEMM_Q192": "0"}, "vitb512_b8"),
                                      ({"SRH_GEMM_SPLITK": "0"}, "vith256_b2"), ({"SRH_ATTN_HDX": "0"}, "vith256_b2")])
def test_alternate_path_matches_oracle(env, case):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_env_path_probe.py"), case], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(env, case, out)
    assert out["finite"] and out["emb_rel_l2"] < 5e-3 and out["score_max_abs"] < 1e-2

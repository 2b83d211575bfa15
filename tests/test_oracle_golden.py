"""Pins the CPU oracle (oracle/) against the committed golden fixtures: outputs of the
reference's own source (TopoNet, BilinearSampler, get_patch_info_one_img, nms_points —
extracted by AST in tests/golden/make_golden.py) and of the independent HF SAM encoder."""
import numpy as np
import pytest
import torch

from conftest import load_golden_module
from oracle.samroad import AttrDict, BilinearSampler, TopoNet
from oracle.sam_encoder import ImageEncoderViT
from oracle import scene


def test_toponet_and_sampler_match_reference_source(golden_dir):
    g = np.load(f"{golden_dir}/toponet_sampler.npz")
    mg = load_golden_module()
    cfg = AttrDict(PATCH_SIZE=512, TOPONET_VERSION="normal")
    topo = TopoNet(cfg, 256).eval()
    topo.load_state_dict(mg.topo_weights(topo.state_dict()), strict=True)
    feats = mg.topo_feats()
    points, pairs, valid = (torch.tensor(g[k]) for k in ("points", "pairs", "valid"))
    with torch.no_grad():
        sampled = BilinearSampler(cfg)(feats, points)
        logits, scores = topo(points, sampled, pairs, valid)
    np.testing.assert_allclose(sampled.numpy(), g["sampled"], atol=1e-6)
    v = valid.numpy().astype(bool)
    np.testing.assert_allclose(logits.numpy()[..., 0][v], g["logits"][..., 0][v], atol=2e-5)
    np.testing.assert_allclose(scores.numpy()[..., 0][v], g["scores"][..., 0][v], atol=1e-5)


@pytest.mark.parametrize("version", ["no_offset", "no_transformer", "no_tgt_features"])
def test_toponet_variants_match_reference_source(golden_dir, version):
    """The oracle's TopoNet for the other TOPONET_VERSIONs against the reference SOURCE executed on the same inputs
    ('no_tgt_features' is overwritten by the reference's if/else chain and equals 'normal')."""
    g = np.load(f"{golden_dir}/toponet_sampler.npz")
    gv = np.load(f"{golden_dir}/toponet_variants.npz")
    mg = load_golden_module()
    cfg = AttrDict(PATCH_SIZE=512, TOPONET_VERSION=version)
    topo = TopoNet(cfg, 256).eval()
    topo.load_state_dict(mg.topo_weights(topo.state_dict()), strict=True)
    points, pairs, valid = (torch.tensor(g[k]) for k in ("points", "pairs", "valid"))
    with torch.no_grad():
        logits, scores = topo(points, torch.tensor(g["sampled"]), pairs, valid)
    v = valid.numpy().astype(bool)
    np.testing.assert_allclose(logits.numpy()[..., 0][v], gv[version + "_logits"][..., 0][v], atol=2e-5)
    np.testing.assert_allclose(scores.numpy()[..., 0][v], gv[version + "_scores"][..., 0][v], atol=1e-5)
    if version == "no_tgt_features":
        np.testing.assert_allclose(gv[version + "_logits"][..., 0][v], g["logits"][..., 0][v], atol=1e-6)


def test_extract_graph_points_matches_reference_source(golden_dir):
    """mask -> graph points: the product's host stage (library mask scan + grid NMS, sam_road_amd/graph_points.py) and the
    oracle's restatement against the reference SOURCE (graph_extraction.py:130-139) — same points in the same order."""
    from sam_road_amd import Config
    from sam_road_amd.graph_points import extract_graph_points
    g = np.load(f"{golden_dir}/graph_points.npz")
    for i in range(3):
        itsc, road, r0, r1 = g[f"cfg{i}"].tolist()
        kw = dict(ITSC_THRESHOLD=itsc, ROAD_THRESHOLD=road, ITSC_NMS_RADIUS=int(r0), ROAD_NMS_RADIUS=int(r1))
        want = g[f"pts{i}"]
        assert want.shape[0] > 5
        np.testing.assert_array_equal(extract_graph_points(g[f"kp{i}"], g[f"road{i}"], Config(**kw)), want)
        np.testing.assert_array_equal(scene.extract_graph_points(g[f"kp{i}"], g[f"road{i}"], AttrDict(kw)), want)


def test_sat2graph_format_matches_reference_source(golden_dir):
    """sam_road_amd.formats against the reference SOURCE (graph_utils.py:383-434) on seeded random graphs: same dict
    (key order, neighbour order) and the same round trip."""
    from sam_road_amd.formats import convert_from_sat2graph_format, convert_to_sat2graph_format
    g = np.load(f"{golden_dir}/sat2graph_format.npz")
    for i in range(3):
        got = convert_to_sat2graph_format(g[f"nodes{i}"], g[f"edges{i}"])
        keys = [tuple(k) for k in g[f"keys{i}"].tolist()]
        assert list(got.keys()) == keys
        flat = [tuple(p) for v in got.values() for p in v]
        assert [len(v) for v in got.values()] == g[f"lens{i}"].tolist()
        assert flat == [tuple(p) for p in g[f"nbrs{i}"].tolist()]
        n2, e2 = convert_from_sat2graph_format(got)
        np.testing.assert_array_equal(np.asarray(n2, dtype=np.int64).reshape(-1, 2), g[f"rt_nodes{i}"])
        np.testing.assert_array_equal(np.asarray(e2, dtype=np.int64).reshape(-1, 2), g[f"rt_edges{i}"])


def test_patch_info_matches_reference_source(golden_dir):
    g = np.load(f"{golden_dir}/patch_info.npz")
    i = 0
    while f"case{i}" in g:
        info = scene.get_patch_info_one_img(0, *[int(v) for v in g[f"case{i}"]])
        got = np.array([[p[1][0], p[1][1], p[2][0], p[2][1]] for p in info])
        np.testing.assert_array_equal(got, g[f"xy{i}"])
        i += 1
    assert i == 7


def test_nms_points_matches_reference_source(golden_dir):
    g = np.load(f"{golden_dir}/nms_points.npz")
    for i in range(4):
        kept = scene.nms_points(g[f"pts{i}"], g[f"sc{i}"], int(g[f"r{i}"]))
        np.testing.assert_array_equal(kept, g[f"kept{i}"])
    np.testing.assert_array_equal(scene.nms_points(g["pts_p"], g["sc_p"], 16), g["kept_p"])


def test_encoder_matches_hf_golden(golden_dir):
    """Oracle encoder restatement vs the stored output of transformers' SamVisionEncoder."""
    mg = load_golden_module()
    y_ref = np.load(f"{golden_dir}/encoder_hf.npz")["y"]
    enc = ImageEncoderViT(img_size=256, embed_dim=128, depth=3, num_heads=2, global_attn_indexes=[1]).eval()

    def ren(k):
        for a, b in (("layers.", "blocks."), ("layer_norm1", "norm1"), ("layer_norm2", "norm2"),
                     ("patch_embed.projection", "patch_embed.proj"), ("neck.conv1", "neck.0"),
                     ("neck.norm1", "neck.1"), ("neck.conv2", "neck.2"), ("neck.norm2", "neck.3")):
            k = k.replace(a, b)
        return k
    # regenerate HF-ordered weights from the frozen stream, rename to the fork's keys
    from transformers.models.sam.modeling_sam import SamVisionEncoder
    hf_sd = mg.hf_small_weights(SamVisionEncoder(mg.hf_small_config()).state_dict())
    enc.load_state_dict({ren(k): v for k, v in hf_sd.items()}, strict=True)
    x = torch.tensor(np.random.RandomState(22).randn(1, 3, 256, 256).astype(np.float32))
    with torch.no_grad():
        y = enc(x)
    np.testing.assert_allclose(y.numpy(), y_ref, atol=2e-4, rtol=1e-4)


def test_torch_kats(golden_dir):
    g = np.load(f"{golden_dir}/torch_kats.npz")
    a = torch.arange(1024, dtype=torch.float32).view(1, 1, 32, 32)
    cfg = AttrDict(PATCH_SIZE=512)
    out = BilinearSampler(cfg)(a, torch.tensor(g["grid_px"]))
    np.testing.assert_allclose(out.flatten().numpy(), g["grid_out"], atol=1e-6)
    np.testing.assert_allclose(g["grid_out"], [0, 0, 0.46875, 0.5, 17.4375], atol=1e-6)
    assert int(g["nan_u8"][0]) == 0
    # dtype promotions the reference relies on
    assert (torch.tensor([3], dtype=torch.int64) / 512).dtype == torch.float32
    assert torch.concat([torch.zeros(1), torch.zeros(1), torch.zeros(1, dtype=torch.int64)]).dtype == torch.float32


def test_encoder_matches_hf_at_true_vitb_512_dims():
    """The oracle encoder vs transformers' independent SamVisionEncoder at the REAL ViT-B / 512^2 geometry of BASELINE
    configs[1] (D 768, 12 heads, all 12 blocks, 9 windows of which 5 are padded, 32x32 global attention with 63-row rel-pos
    tables) — computed live (HF is importable wherever the tests run; ~2 s), one tile."""
    from transformers import SamVisionConfig
    from transformers.models.sam.modeling_sam import SamVisionEncoder
    cfg = SamVisionConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, image_size=512, patch_size=16,
                          window_size=14, global_attn_indexes=[2, 5, 8, 11], mlp_dim=3072, output_channels=256,
                          layer_norm_eps=1e-6, use_rel_pos=True, use_abs_pos=True, qkv_bias=True, hidden_act="gelu")
    cfg._attn_implementation = "eager"
    hf = SamVisionEncoder(cfg).eval()
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, v in hf.state_dict().items():
        if v.dim() == 1 and "layer_norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = 0.02 * torch.randn(v.shape, generator=g)
    hf.load_state_dict(sd)

    def ren(k):
        for a, b in (("layers.", "blocks."), ("layer_norm1", "norm1"), ("layer_norm2", "norm2"),
                     ("patch_embed.projection", "patch_embed.proj"), ("neck.conv1", "neck.0"),
                     ("neck.norm1", "neck.1"), ("neck.conv2", "neck.2"), ("neck.norm2", "neck.3")):
            k = k.replace(a, b)
        return k
    enc = ImageEncoderViT(img_size=512, embed_dim=768, depth=12, num_heads=12, global_attn_indexes=[2, 5, 8, 11]).eval()
    enc.load_state_dict({ren(k): v for k, v in sd.items()}, strict=True)
    assert enc.blocks[2].attn.rel_pos_h.shape == (63, 64) and enc.blocks[0].attn.rel_pos_h.shape == (27, 64)
    x = torch.randn(1, 3, 512, 512, generator=g)
    with torch.no_grad():
        y_hf = hf(x).last_hidden_state
        y = enc(x)
    assert y.shape == (1, 256, 32, 32)
    err = (y - y_hf).abs().max().item()
    print("oracle vs HF SamVisionEncoder, ViT-B 512^2, 12 blocks: max abs", err)
    assert err < 2e-4

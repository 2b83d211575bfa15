"""Module-level parity of the HIP path (SAMRoad through the C ABI) against the CPU oracle on the same
seeded synthetic state_dict and inputs.  Tolerances (fp16 MFMA operands, f32 accumulate, f32 residual
stream vs the reference's eager fp32):
    image embeddings   rel-L2 <= 3e-3, max-abs <= 1.7e-2  (post-LayerNorm2d values, O(1))
    mask scores        max-abs <= 5e-4, u8 masks within +-2 levels, +-1 on >= 99.9 % of pixels
    topo scores        max-abs <= 3e-3 on valid pairs, edge decisions (>0.5) equal on >= 99.8 %
— every bound is <= 3x the value measured on MI355X and lives in tests/tolerances.py.  Run on an MI355X: pytest -m gpu."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.samroad import AttrDict, SAMRoadOracle
from oracle import scene as oscene
from oracle.synth import synth_queries, synth_state_dict, synth_tiles, synth_scene

import tolerances as T


def build_pair(cfg_kwargs, seed=1234):
    from sam_road_amd import Config, SAMRoad
    warnings.simplefilter("ignore")
    oracle = SAMRoadOracle(AttrDict(cfg_kwargs)).eval()
    sd = synth_state_dict(oracle, seed)
    oracle.load_state_dict(sd, strict=True)
    net = SAMRoad(Config(cfg_kwargs))
    net.load_state_dict(sd, strict=True)
    net.eval()
    net.to("cuda")
    return oracle, net


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


CFG512 = dict(SAM_VERSION="vit_b", PATCH_SIZE=512, TOPONET_VERSION="normal", SAM_CKPT_PATH="")
CFG256 = dict(SAM_VERSION="vit_b", PATCH_SIZE=256, TOPONET_VERSION="normal", SAM_CKPT_PATH="")
CFG_L256 = dict(SAM_VERSION="vit_l", PATCH_SIZE=256, TOPONET_VERSION="normal", SAM_CKPT_PATH="")
CFG_H256 = dict(SAM_VERSION="vit_h", PATCH_SIZE=256, TOPONET_VERSION="normal", SAM_CKPT_PATH="")   # toponet_vith_256.yaml (BASELINE configs[4])


@pytest.mark.parametrize("cfg,B,depth,gidx", [
    (CFG512, 2, 1, []),        # one windowed block
    (CFG512, 1, 1, [0]),       # one global block
    (CFG256, 2, 2, [1]),       # 256 tile: windowed + global(16)
    (CFG_L256, 1, 2, [1]),     # ViT-L: D = 1024, 16 heads x 64
    (CFG_H256, 2, 2, [1]),     # ViT-H: D = 1280, 16 heads x 80 (attention_hdx.hip MFMA kernels; split-K GEMMs)
    (dict(CFG512, PATCH_SIZE=1024), 1, 2, [1]),   # toponet_vitb_1024.yaml: 5x5 windows + 64x64 global window
    (CFG512, 8, 2, [1]),       # 8192 tokens: the persistent q192 GEMM + residual-add-in-LayerNorm path is taken
])
def test_shallow_encoder_parity(cfg, B, depth, gidx):
    cfg = dict(cfg, ENCODER_DEPTH=depth, ENCODER_GLOBAL_ATTN_INDEXES=gidx)
    oracle, net = build_pair(cfg)
    rgb = synth_tiles(B, cfg["PATCH_SIZE"], seed=3)
    s_ref, e_ref = oracle.infer_masks_and_img_features(rgb)
    s, e = net.infer_masks_and_img_features(rgb.cuda())
    assert tuple(e.shape) == tuple(e_ref.shape) and tuple(s.shape) == tuple(s_ref.shape)
    e, s = e.cpu(), s.cpu()
    print("emb rel-l2", rel_l2(e, e_ref), "max", (e - e_ref).abs().max().item(),
          "score max", (s - s_ref).abs().max().item())
    tag = f"shallow_{cfg['SAM_VERSION']}_{cfg['PATCH_SIZE']}_B{B}_d{depth}_g{len(gidx)}"
    T.check(tag + "_emb_rel_l2", rel_l2(e, e_ref), T.EMB_REL_L2_SHALLOW)
    T.check(tag + "_mask_score", (s - s_ref).abs().max().item(), T.MASK_SCORE)


def test_full_forward_parity_vitb_512():
    """BASELINE config 3 shape at a CPU-tractable batch: full SAMRoad.forward."""
    oracle, net = build_pair(CFG512)
    B = 2
    rgb = synth_tiles(B, 512, seed=0)
    points, pairs, valid = synth_queries(B, 96, 512, seed=7)
    ml_r, ms_r, tl_r, ts_r = oracle(rgb, points, pairs, valid)
    ml, ms, tl, ts = net(rgb.cuda(), points.cuda(), pairs.cuda(), valid.cuda())
    ml, ms, tl, ts = ml.cpu(), ms.cpu(), tl.cpu(), ts.cpu()
    _, e_r = oracle.infer_masks_and_img_features(rgb)
    _, e = net.infer_masks_and_img_features(rgb.cuda())
    e = e.cpu()
    v = valid.bool()
    print("emb rel-l2", rel_l2(e, e_r), "emb max", (e - e_r).abs().max().item())
    print("mask score max", (ms - ms_r).abs().max().item(), "logit max", (ml - ml_r).abs().max().item())
    print("topo score max", (ts[..., 0][v] - ts_r[..., 0][v]).abs().max().item())
    T.check("forward_vitb512_b2_emb_rel_l2", rel_l2(e, e_r), T.EMB_REL_L2)
    T.check("forward_vitb512_b2_emb_max_abs", (e - e_r).abs().max().item(), T.EMB_MAX_ABS)
    T.check("forward_vitb512_b2_mask_score", (ms - ms_r).abs().max().item(), T.MASK_SCORE)
    T.check("forward_vitb512_b2_mask_logit", (ml - ml_r).abs().max().item(), T.MASK_LOGIT)
    u8 = lambda t: (t * 255).to(torch.uint8).int()
    lv = (u8(ms) - u8(ms_r)).abs()
    assert lv.max().item() <= 2
    T.check("forward_vitb512_b2_u8_within1", (lv <= 1).float().mean().item(), T.U8_WITHIN1, at_least=True)
    T.check("forward_vitb512_b2_topo_score", (ts[..., 0][v] - ts_r[..., 0][v]).abs().max().item(), T.TOPO_SCORE)
    agree = ((ts[..., 0][v] > 0.5) == (ts_r[..., 0][v] > 0.5)).float().mean().item()
    T.check("forward_vitb512_b2_topo_decisions", agree, T.TOPO_DECISIONS, at_least=True)
    assert torch.isfinite(ts[..., 0][v]).all()


def test_config1_256_tile():
    """BASELINE config 1 (toponet_vitb_256_spacenet, single tile) against the oracle."""
    oracle, net = build_pair(CFG256)
    rgb = synth_tiles(1, 256, seed=1)
    points, pairs, valid = synth_queries(1, 64, 256, seed=9)
    ml_r, ms_r, tl_r, ts_r = oracle(rgb, points, pairs, valid)
    ml, ms, tl, ts = [t.cpu() for t in net(rgb.cuda(), points.cuda(), pairs.cuda(), valid.cuda())]
    v = valid.bool()
    T.check("config0_vitb256_b1_mask_score", (ms - ms_r).abs().max().item(), T.MASK_SCORE)
    T.check("config0_vitb256_b1_topo_score", (ts[..., 0][v] - ts_r[..., 0][v]).abs().max().item(), T.TOPO_SCORE)


def test_toponet_golden_through_hip(golden_dir):
    """TopoNet + BilinearSampler outputs of the REFERENCE's own source (golden fixture) vs the HIP path."""
    from conftest import load_golden_module
    mg = load_golden_module()
    g = np.load(f"{golden_dir}/toponet_sampler.npz")
    oracle, net = build_pair(CFG512 | dict(ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[]))
    sd = net.state_dict()
    for k, v in mg.topo_weights(oracle.topo_net.state_dict()).items():
        sd["topo_net." + k] = v
    net.load_state_dict(sd, strict=True)
    net.to("cuda")
    feats = mg.topo_feats()
    points, pairs, valid = (torch.tensor(g[k]) for k in ("points", "pairs", "valid"))
    ts = net.infer_toponet(feats.cuda(), points.cuda(), pairs.cuda(), valid.cuda()).cpu()
    v = valid.numpy().astype(bool)
    err = np.abs(ts.numpy()[..., 0][v] - g["scores"][..., 0][v]).max()
    print("toponet vs reference-source golden: max abs", err)
    T.check("toponet_normal_vs_reference_source_golden", err, T.TOPO_GOLDEN)
    # all-invalid row (flipped to all-valid) and out-of-tile points stay finite
    assert np.isfinite(ts.numpy()).all()


@pytest.mark.parametrize("version", ["no_offset", "no_transformer", "no_tgt_features"])
def test_toponet_variants_golden_through_hip(golden_dir, version):
    """The other TOPONET_VERSIONs: outputs of the REFERENCE's own source (golden fixture) vs the HIP path."""
    from conftest import load_golden_module
    mg = load_golden_module()
    g = np.load(f"{golden_dir}/toponet_sampler.npz")
    gv = np.load(f"{golden_dir}/toponet_variants.npz")
    oracle, net = build_pair(CFG512 | dict(ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[], TOPONET_VERSION=version))
    sd = net.state_dict()
    for k, v in mg.topo_weights(oracle.topo_net.state_dict()).items():
        sd["topo_net." + k] = v
    net.load_state_dict(sd, strict=True)
    net.to("cuda")
    points, pairs, valid = (torch.tensor(g[k]) for k in ("points", "pairs", "valid"))
    ts = net.infer_toponet(mg.topo_feats().cuda(), points.cuda(), pairs.cuda(), valid.cuda()).cpu()
    v = valid.numpy().astype(bool)
    err = np.abs(ts.numpy()[..., 0][v] - gv[version + "_scores"][..., 0][v]).max()
    print(version, "vs reference-source golden: max abs", err)
    T.check(f"toponet_{version}_vs_reference_source_golden", err, T.TOPO_GOLDEN)
    assert np.isfinite(ts.numpy()).all()


def test_u8_and_f32_inputs_agree():
    _, net = build_pair(CFG512 | dict(ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[]))
    rgb = synth_tiles(1, 512, seed=4)
    s0, e0 = net.infer_masks_and_img_features(rgb.cuda())
    s1, e1 = net.infer_masks_and_img_features(rgb.to(torch.uint8).cuda())
    assert torch.equal(e0, e1) and torch.equal(s0, s1)


def test_full_size_properties_baseline_config():
    """BASELINE configs[1] at its full size (B = 16 tiles of 512^2, ViT-B, all 12 blocks) — too big for the CPU oracle, so
    the size-independent properties of the path are checked instead: (1) bit-exact determinism, (2) tiles of a batch are
    independent: permuting the tiles permutes the outputs bit-exactly, (3) a tile's result does not depend on the batch it
    travels in beyond kernel-selection rounding (B = 16 takes the persistent q192 GEMMs, B = 2 the 128x128 kernels)."""
    _, net = build_pair(CFG512)
    rgb = synth_tiles(16, 512, seed=9).cuda()
    s0, e0 = net.infer_masks_and_img_features(rgb)
    s1, e1 = net.infer_masks_and_img_features(rgb)
    assert torch.isfinite(s0).all() and torch.isfinite(e0).all()
    assert torch.equal(s0, s1) and torch.equal(e0, e1)
    perm = torch.tensor([5, 0, 11, 3, 15, 8, 1, 13, 2, 7, 10, 4, 14, 6, 9, 12], device="cuda")
    sp, ep = net.infer_masks_and_img_features(rgb[perm].contiguous())
    assert torch.equal(sp, s0[perm]) and torch.equal(ep, e0[perm])
    s2, e2 = net.infer_masks_and_img_features(rgb[:2].contiguous())
    # B = 32 and B = 64 (INFER_BATCH_SIZE of the shipped CityScale YAML: 65 536-row GEMMs): the first 16 tiles are the same tiles —
    # every kernel on this path works per token row / per (tile, head, window), so the results must be BIT-identical
    for Bbig in (32, 64):
        big = torch.cat([rgb, synth_tiles(Bbig - 16, 512, seed=40 + Bbig).cuda()], 0)
        sb, eb = net.infer_masks_and_img_features(big)
        assert torch.isfinite(sb).all() and torch.isfinite(eb).all()
        assert torch.equal(sb[:16], s0) and torch.equal(eb[:16], e0), f"B={Bbig} changes the result of a tile"
        del big, sb, eb
    T.check("batch16_vs_batch2_emb_rel_l2", rel_l2(e2.float().cpu(), e0[:2].float().cpu()), T.BATCH_INDEP_EMB_REL)
    T.check("batch16_vs_batch2_mask_score", (s2 - s0[:2]).abs().max().item(), T.BATCH_INDEP_SCORE)


@pytest.mark.parametrize("version", ["normal", "no_offset", "no_transformer"])
def test_toponet_ragged_and_variants(version):
    """TopoNet (fused trunk) vs the oracle on ragged shapes: 3 tiles x 37 source points (111 sequences: the last workgroup
    of the fused kernel is partly empty), sequences with no valid pair at all (model.py:129-130 flip), points outside the
    tile, for every TOPONET_VERSION."""
    cfg = CFG256 | dict(ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[], TOPONET_VERSION=version)
    oracle, net = build_pair(cfg, seed=21)
    g = torch.Generator().manual_seed(5)
    B, N, K = 3, 37, 16
    emb = torch.randn(B, 256, 16, 16, generator=g)
    points = torch.randint(-8, 264, (B, N, 2), generator=g)
    pairs = torch.stack([torch.arange(N)[None, :, None].expand(B, N, K), torch.randint(0, N, (B, N, K), generator=g)], -1)
    valid = torch.rand(B, N, K, generator=g) < 0.6
    valid[0, 3] = False
    valid[2, 36] = False
    valid[1, 0] = True
    with torch.no_grad():
        feats = oracle.bilinear_sampler(emb, points)
        _, ts_r = oracle.topo_net(points, feats, pairs, valid)
    ts = net.infer_toponet(emb.cuda(), points.cuda(), pairs.cuda(), valid.cuda()).cpu()
    assert tuple(ts.shape) == (B, N, K, 1) and torch.isfinite(ts).all()
    v = valid.clone()
    v[0, 3] = True          # flipped rows: every key takes part, every score is defined
    v[2, 36] = True
    err = (ts[..., 0][v] - ts_r[..., 0][v]).abs().max().item()
    T.check(f"toponet_ragged_{version}_score", err, T.TOPO_SCORE)


def test_packed_weights_export_import_roundtrip():
    """The multi-GPU weight path on one GPU (SAMRoad.share_packed_weights = export on rank 0, RCCL broadcast, import elsewhere):
    a second model of the same configuration with DIFFERENT parameters adopts the exported packed arena and must reproduce the
    first model's outputs bit for bit, TopoNet included; a mismatching configuration is refused; editing the parameters of a
    model that runs on imported weights raises instead of silently re-packing the local ones."""
    from sam_road_amd import Config, SAMRoad, _lib
    cfg = CFG256 | dict(ENCODER_DEPTH=2, ENCODER_GLOBAL_ATTN_INDEXES=[1])
    _, net = build_pair(cfg)
    rgb = synth_tiles(2, 256, seed=8).cuda()
    points, pairs, valid = (t.cuda() for t in synth_queries(2, 40, 256, seed=4))
    want = net(rgb, points, pairs, valid)
    buf = net.export_packed(torch.device("cuda", 0))
    assert buf.dtype == torch.uint8 and buf.is_cuda and buf.numel() > 1 << 20
    other = SAMRoad(Config(cfg))                      # its own (default-initialised) parameters
    other.eval().to("cuda")
    assert not torch.equal(other(rgb, points, pairs, valid)[1], want[1])
    other.import_packed(buf.clone())
    got = other(rgb, points, pairs, valid)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    # an _apply that moves nothing must not make the model fall back to its own (never loaded) parameters
    other.to("cuda")
    other.cuda()
    other.float()
    for a, b in zip(want, other(rgb, points, pairs, valid)):
        assert torch.equal(a, b)
    wrong = SAMRoad(Config(cfg | dict(ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[])))
    wrong.eval().to("cuda")
    with pytest.raises(_lib.SrhError):
        wrong.import_packed(buf)
    with torch.no_grad():
        next(other.parameters()).add_(1.0)
    with pytest.raises(_lib.SrhError):
        other.infer_masks_and_img_features(rgb)
    # ... and once the arena is gone for any reason the imported model refuses instead of re-packing
    third = SAMRoad(Config(cfg))
    third.eval().to("cuda")
    third.import_packed(buf.clone())
    third._invalidate()
    with pytest.raises(_lib.SrhError):
        third.infer_masks_and_img_features(rgb)


def test_packed_weights_through_rccl_broadcast():
    """The collective itself on the GPU box (one rank: gpurun exposes one GPU): a process group on backend nccl (= RCCL), the
    packed arena of a model through distributed.broadcast_bytes exactly as SAMRoad.share_packed_weights sends it, imported into
    a second model — same outputs bit for bit.  (World sizes > 1: the same code under gloo in tests/test_distributed_cpu.py.)"""
    import os
    import socket
    import torch.distributed as dist
    from sam_road_amd import Config, SAMRoad
    from sam_road_amd import distributed as D
    cfg = CFG256 | dict(ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[])
    _, net = build_pair(cfg)
    rgb = synth_tiles(1, 256, seed=2).cuda()
    want = net.infer_masks_and_img_features(rgb)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        buf = D.broadcast_bytes(net.export_packed(torch.device("cuda", 0)), src=0, device=torch.device("cuda", 0))
        net.share_packed_weights(src=0)                   # world size 1: a no-op by definition
    finally:
        dist.destroy_process_group()
    other = SAMRoad(Config(cfg))
    other.eval().to("cuda")
    other.import_packed(buf)
    got = other.infer_masks_and_img_features(rgb)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


@pytest.mark.parametrize("which,B", [("image_encoder.blocks.0.mlp.lin2.weight", 8), ("image_encoder.blocks.1.attn.qkv.weight", 2),
                                     ("image_encoder.blocks.1.attn.qkv.weight", 8)])
def test_fp16_overflow_is_reported_not_returned_as_masks(which, B):
    """ABI 8 non-finite sentinel (include/samroad_hip.h srh_ctx_check, SRH_ERR_NONFINITE; the reference's own guards,
    inferencer.py:206,219, only see TopoNet's output): one weight scaled until an fp16 inter-kernel tensor overflows.  The call
    itself is asynchronous and returns; the condition is raised by check_finite() and — lazily — by the next call on the context,
    with the first stage that saw it in the message; reporting clears it and a healthy model runs again.  B = 8 takes the z192
    GEMM path (fp16 branch outputs folded by the LayerNorm pass: lin2's output overflows there), B = 2 the generic epilogues (f32
    residual add in the GEMM — only qkv16 / attn16 / hid16 are fp16)."""
    from sam_road_amd import Config, SAMRoad, _lib
    cfg = dict(CFG512, ENCODER_DEPTH=2, ENCODER_GLOBAL_ATTN_INDEXES=[1])
    oracle, good = build_pair(cfg)
    sd = {k: v.clone() for k, v in oracle.state_dict().items()}
    sd[which] = sd[which] * 3.0e6
    bad = SAMRoad(Config(cfg))
    bad.load_state_dict(sd, strict=True)
    bad.eval().to("cuda")
    rgb = synth_tiles(B, 512, seed=3).cuda()
    good.check_finite()                                      # nothing pending
    s, e = bad.infer_masks_and_img_features(rgb)             # queued; the overflow happens on the device
    with pytest.raises(_lib.SrhError, match="non-finite.*encoder block 1"):
        bad.check_finite()
    good.check_finite()                                      # reported once, then clear
    s, e = bad.infer_masks_and_img_features(rgb)
    torch.cuda.synchronize()
    with pytest.raises(_lib.SrhError, match="non-finite"):   # lazily, by the next call on the context (no synchronisation inside)
        good.infer_masks_and_img_features(rgb)
    s, e = good.infer_masks_and_img_features(rgb)
    good.check_finite()
    assert torch.isfinite(e).all() and torch.isfinite(s).all()

"""Pins the oracle's SAM MaskDecoder / TwoWayTransformer / PromptEncoder restatement (oracle/sam_decoder.py — the fork's
source is absent) to the independent implementation in transformers (SamMaskDecoder, SamPositionalEmbedding),
computed live on seeded weights with the fork's -> HF key renames; and checks the reference's key layout for the
USE_SAM_DECODER branch (model.py:260-282).  HF's decoder LayerNorms use eps 1e-6 by default where upstream SAM's nn.LayerNorm
uses 1e-5: the HF config is set to 1e-5 for the comparison."""
import numpy as np
import torch

from oracle.sam_decoder import MaskDecoder, PromptEncoder, TwoWayTransformer
from oracle.samroad import AttrDict, SAMRoadOracle


def _ren(k):
    """fork (upstream segment-anything) key -> HF key"""
    for a, b in ((".norm1.", ".layer_norm1."), (".norm2.", ".layer_norm2."), (".norm3.", ".layer_norm3."), (".norm4.", ".layer_norm4."),
                 ("transformer.norm_final_attn", "transformer.layer_norm_final_attn"),
                 ("output_upscaling.0.", "upscale_conv1."), ("output_upscaling.1.", "upscale_layer_norm."),
                 ("output_upscaling.3.", "upscale_conv2.")):
        k = k.replace(a, b)
    if "output_hypernetworks_mlps" in k or "iou_prediction_head" in k:
        k = k.replace("layers.0.", "proj_in.").replace("layers.2.", "proj_out.").replace("layers.1.", "layers.0.")
    return k


def test_mask_decoder_matches_hf():
    from transformers import SamMaskDecoderConfig
    from transformers.models.sam.modeling_sam import SamMaskDecoder, SamPositionalEmbedding
    from transformers import SamVisionConfig
    g = torch.Generator().manual_seed(3)
    B, S = 2, 16
    dec = MaskDecoder(num_multimask_outputs=2, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                      transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256).eval()
    pe = PromptEncoder(embed_dim=256, image_embedding_size=(S, S), input_image_size=(256, 256), mask_in_chans=16).eval()
    sd = {}
    for k, v in dec.state_dict().items():
        sd[k] = (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if (v.dim() == 1 and "norm" in k and k.endswith("weight")) or \
            k in ("output_upscaling.1.weight",) else 0.05 * torch.randn(v.shape, generator=g)
    dec.load_state_dict(sd)
    psd = {k: 0.5 * torch.randn(v.shape, generator=g) for k, v in pe.state_dict().items()}
    pe.load_state_dict(psd)

    cfg = SamMaskDecoderConfig(hidden_size=256, hidden_act="relu", mlp_dim=2048, num_hidden_layers=2, num_attention_heads=8,
                               attention_downsample_rate=2, num_multimask_outputs=2, iou_head_depth=3, iou_head_hidden_dim=256,
                               layer_norm_eps=1e-5)
    cfg._attn_implementation = "eager"
    hf = SamMaskDecoder(cfg).eval()
    hf_sd = {_ren(k): v for k, v in sd.items()}
    missing, unexpected = hf.load_state_dict(hf_sd, strict=True)
    assert not missing and not unexpected

    emb = torch.randn(B, 256, S, S, generator=g)
    sparse, dense = pe(points=None, boxes=None, masks=None)
    image_pe = pe.get_dense_pe()
    # HF's random-Fourier grid encoding with the same gaussian matrix
    hf_pe = SamPositionalEmbedding(SamVisionConfig(num_pos_feats=128, scale=1.0))
    hf_pe.positional_embedding.data.copy_(psd["pe_layer.positional_encoding_gaussian_matrix"])
    grid = torch.ones((S, S))
    y, x = (grid.cumsum(0) - 0.5) / S, (grid.cumsum(1) - 0.5) / S
    with torch.no_grad():
        hf_image_pe = hf_pe(torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)
    np.testing.assert_allclose(image_pe.numpy(), hf_image_pe.numpy(), atol=1e-5)

    with torch.no_grad():
        masks, iou = dec(image_embeddings=emb, image_pe=image_pe, sparse_prompt_embeddings=sparse,
                         dense_prompt_embeddings=dense, multimask_output=True)
        hf_masks, hf_iou = hf(image_embeddings=emb, image_positional_embeddings=hf_image_pe.expand(B, -1, -1, -1),
                              sparse_prompt_embeddings=None, dense_prompt_embeddings=dense, multimask_output=True)
    assert masks.shape == (B, 2, 4 * S, 4 * S) and iou.shape == (B, 2)
    err = (masks - hf_masks[:, 0]).abs().max().item()
    print("oracle MaskDecoder vs HF SamMaskDecoder: max abs", err, "ref max", hf_masks.abs().max().item())
    assert err < 1e-4 * max(1.0, hf_masks.abs().max().item())
    np.testing.assert_allclose(iou.numpy(), hf_iou[:, 0].numpy(), atol=1e-4)


def test_samroad_oracle_with_sam_decoder_runs_and_has_the_forks_keys():
    cfg = AttrDict(SAM_VERSION="vit_b", PATCH_SIZE=256, USE_SAM_DECODER=True, TOPONET_VERSION="normal",
                   ENCODER_DEPTH=1, ENCODER_GLOBAL_ATTN_INDEXES=[])
    net = SAMRoadOracle(cfg).eval()
    keys = set(net.state_dict().keys())
    assert not any(k.startswith("map_decoder") for k in keys)
    for k in ("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", "prompt_encoder.no_mask_embed.weight",
              "prompt_encoder.point_embeddings.3.weight", "prompt_encoder.mask_downscaling.6.bias",
              "mask_decoder.transformer.layers.1.cross_attn_image_to_token.out_proj.weight",
              "mask_decoder.transformer.norm_final_attn.bias", "mask_decoder.iou_token.weight", "mask_decoder.mask_tokens.weight",
              "mask_decoder.output_upscaling.3.weight", "mask_decoder.output_hypernetworks_mlps.2.layers.2.bias",
              "mask_decoder.iou_prediction_head.layers.0.weight"):
        assert k in keys, k
    assert net.state_dict()["mask_decoder.mask_tokens.weight"].shape == (3, 256)
    scores, emb = net.infer_masks_and_img_features(torch.rand(2, 256, 256, 3) * 255)
    assert scores.shape == (2, 256, 256, 2) and emb.shape == (2, 256, 16, 16) and torch.isfinite(scores).all()

"""Oracle: ``SAMRoad`` inference surface, ``BilinearSampler``, ``TopoNet``
(test infrastructure only — see ``oracle/__init__.py``).

Follows reference ``model.py:29-58`` (sampler), ``:61-148`` (TopoNet),
``:190-258,283-300`` (module tree) and ``:414-508`` (the three entry points),
as a plain ``nn.Module`` with identical ``state_dict`` key names.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .sam_decoder import MaskDecoder, PromptEncoder, TwoWayTransformer
from .sam_encoder import ARCH, ImageEncoderViT, LayerNorm2d


class AttrDict(dict):
    """``addict.Dict`` semantics the reference relies on (``utils.py:6-9``):
    attribute access, and a missing key evaluates to an empty falsy dict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            return AttrDict()

    def __setattr__(self, k, v):
        self[k] = v


def load_config(path):
    import yaml
    with open(path) as f:
        return AttrDict(yaml.safe_load(f))


class BilinearSampler(nn.Module):  # model.py:29-58
    def __init__(self, config):
        super().__init__()
        self.config = config

    def forward(self, feature_maps, sample_points):
        grid = (sample_points / self.config.PATCH_SIZE) * 2.0 - 1.0  # :47
        out = F.grid_sample(feature_maps, grid.unsqueeze(2), mode="bilinear",
                            align_corners=False)  # :54 (zero padding)
        return out.squeeze(dim=-1).permute(0, 2, 1)


class TopoNet(nn.Module):  # model.py:61-148
    def __init__(self, config, feature_dim):
        super().__init__()
        self.config = config
        self.hidden_dim, self.heads, self.num_attn_layers = 128, 4, 3
        self.feature_proj = nn.Linear(feature_dim, self.hidden_dim)
        self.pair_proj = nn.Linear(2 * self.hidden_dim + 2, self.hidden_dim)
        layer = nn.TransformerEncoderLayer(
            d_model=self.hidden_dim, nhead=self.heads, dim_feedforward=self.hidden_dim,
            dropout=0.1, activation="relu", batch_first=True)
        if config.TOPONET_VERSION != "no_transformer":
            self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=self.num_attn_layers)
        self.output_proj = nn.Linear(self.hidden_dim, 1)

    def forward(self, points, point_features, pairs, pairs_valid):
        point_features = F.relu(self.feature_proj(point_features))
        B, n_samples, n_pairs, _ = pairs.shape
        pairs = pairs.view(B, -1, 2).long()
        bidx = torch.arange(B).view(-1, 1).expand(-1, n_samples * n_pairs)
        src_f = point_features[bidx, pairs[:, :, 0]]
        tgt_f = point_features[bidx, pairs[:, :, 1]]
        offset = points[bidx, pairs[:, :, 1]] - points[bidx, pairs[:, :, 0]]
        # model.py:111-116 — 'no_tgt_features' is overwritten by the following
        # non-elif if/else and therefore behaves as 'normal' 
        if self.config.TOPONET_VERSION == "no_offset":
            feats = torch.concat([src_f, tgt_f, torch.zeros_like(offset)], dim=2)
        else:
            feats = torch.concat([src_f, tgt_f, offset], dim=2)
        feats = F.relu(self.pair_proj(feats)).view(B * n_samples, n_pairs, -1)
        valid = pairs_valid.view(B * n_samples, n_pairs).bool()
        # model.py:129-130: all-invalid rows are flipped to all-valid
        valid = torch.logical_or(valid, torch.eq(valid.sum(-1), 0).unsqueeze(-1))
        if self.config.TOPONET_VERSION != "no_transformer":
            feats = self.transformer_encoder(feats, src_key_padding_mask=~valid)
        feats = feats.view(B, n_samples, feats.shape[1], -1)
        logits = self.output_proj(feats)
        return logits, torch.sigmoid(logits)


class SAMRoadOracle(nn.Module):
    """model.py:190-300 (inference-relevant part): naive map_decoder, or SAM's MaskDecoder when USE_SAM_DECODER."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        assert config.SAM_VERSION in {"vit_b", "vit_l", "vit_h"}
        if config.NO_SAM:
            raise NotImplementedError("NO_SAM ablation (model.py:232-242)")
        arch = dict(ARCH[config.SAM_VERSION])
        # test hook (not a reference key; absent => falsy => ignored)
        if config.ENCODER_DEPTH:
            arch["depth"] = int(config.ENCODER_DEPTH)
            arch["global_attn_indexes"] = [int(i) for i in config.ENCODER_GLOBAL_ATTN_INDEXES or []]
        self.image_size = config.PATCH_SIZE
        self.register_buffer("pixel_mean", torch.Tensor([123.675, 116.28, 103.53]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor([58.395, 57.12, 57.375]).view(-1, 1, 1), False)
        self.image_encoder = ImageEncoderViT(img_size=config.PATCH_SIZE, patch_size=16,
                                             window_size=14, out_chans=256, **arch)
        act = nn.GELU
        if config.USE_SAM_DECODER:  # model.py:260-282
            s = config.PATCH_SIZE // 16
            self.prompt_encoder = PromptEncoder(embed_dim=256, image_embedding_size=(s, s),
                                                input_image_size=(config.PATCH_SIZE, config.PATCH_SIZE), mask_in_chans=16)
            self.mask_decoder = MaskDecoder(
                num_multimask_outputs=2,
                transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
        else:
            self.map_decoder = nn.Sequential(  # model.py:286-295
                nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2), LayerNorm2d(128), act(),
                nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2), act(),
                nn.ConvTranspose2d(64, 32, kernel_size=2, stride=2), act(),
                nn.ConvTranspose2d(32, 2, kernel_size=2, stride=2))
        self.bilinear_sampler = BilinearSampler(config)
        self.topo_net = TopoNet(config, 256)

    def _encode(self, rgb):
        x = rgb.permute(0, 3, 1, 2)
        x = (x - self.pixel_mean) / self.pixel_std  # model.py:465-467
        return self.image_encoder(x)

    def _decode(self, emb):
        """model.py:425-446 / :470-491 -> mask logits [B,2,P,P]."""
        if not self.config.USE_SAM_DECODER:
            return self.map_decoder(emb)
        sparse, dense = self.prompt_encoder(points=None, boxes=None, masks=None)
        low_res, _ = self.mask_decoder(image_embeddings=emb, image_pe=self.prompt_encoder.get_dense_pe(),
                                       sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                       multimask_output=True)
        return F.interpolate(low_res, (self.image_encoder.img_size, self.image_encoder.img_size), mode="bilinear",
                             align_corners=False)

    @torch.no_grad()
    def forward(self, rgb, graph_points, pairs, valid):  # model.py:414-457
        emb = self._encode(rgb)
        mask_logits = self._decode(emb)
        mask_scores = torch.sigmoid(mask_logits)
        feats = self.bilinear_sampler(emb, graph_points)
        topo_logits, topo_scores = self.topo_net(graph_points, feats, pairs, valid)
        return (mask_logits.permute(0, 2, 3, 1), mask_scores.permute(0, 2, 3, 1),
                topo_logits, topo_scores)

    @torch.no_grad()
    def infer_masks_and_img_features(self, rgb):  # model.py:459-495
        emb = self._encode(rgb)
        mask_scores = torch.sigmoid(self._decode(emb))
        return mask_scores.permute(0, 2, 3, 1), emb

    @torch.no_grad()
    def infer_toponet(self, image_embeddings, graph_points, pairs, valid):  # model.py:498-508
        feats = self.bilinear_sampler(image_embeddings, graph_points)
        return self.topo_net(graph_points, feats, pairs, valid)[1]

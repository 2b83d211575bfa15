"""Oracle: SAM ``PromptEncoder`` (no-prompt path), ``TwoWayTransformer`` and ``MaskDecoder`` (test infrastructure only).

The reference builds these at ``model.py:260-282`` (``USE_SAM_DECODER: True``, archived configs only) and runs them at
``model.py:426-443`` / ``:471-488`` from the un-vendored fork ``sam/segment_anything/modeling/{prompt_encoder,mask_decoder,
transformer}.py`` (absent from the snapshot, see ``oracle/__init__.py``).  The algorithm below restates upstream
facebookresearch/segment-anything with the fork's parameter names (so ``prompt_encoder.*`` / ``mask_decoder.*`` checkpoint keys
load unchanged) and is cross-checked against ``transformers.models.sam.modeling_sam.{SamMaskDecoder, SamPositionalEmbedding}``
in ``tests/test_oracle_sam_decoder.py``.  PARITY UNPINNED by the reference itself, exactly like the encoder.

What the reference's call computes (per batch of B tiles, S = PATCH_SIZE / 16):
    sparse, dense = prompt_encoder(points=None, boxes=None, masks=None)         # sparse [1,0,256]; dense = no_mask_embed -> [1,256,S,S]
    low_res, iou = mask_decoder(image_embeddings[B,256,S,S], image_pe = prompt_encoder.get_dense_pe()[1,256,S,S],
                                sparse, dense, multimask_output=True)           # low_res [B,2,4S,4S] = mask tokens 1, 2
    mask_logits = F.interpolate(low_res, (P, P), mode="bilinear", align_corners=False)
The token batch is 1 and broadcasts against the B image batches inside the attention matmuls; here the 4 output tokens
(iou token + 3 mask tokens) are expanded to B up front, which is the same arithmetic.
"""
import math

import numpy as np
import torch
from torch import nn

from .sam_encoder import LayerNorm2d


class PositionEmbeddingRandom(nn.Module):
    """fork ``prompt_encoder.py``: random-Fourier positional encoding of the S x S grid."""

    def __init__(self, num_pos_feats=64, scale=None):
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))

    def _pe_encoding(self, coords):
        coords = 2 * coords - 1
        coords = coords @ self.positional_encoding_gaussian_matrix
        coords = 2 * np.pi * coords
        return torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)

    def forward(self, size):
        h, w = size
        grid = torch.ones((h, w), dtype=torch.float32, device=self.positional_encoding_gaussian_matrix.device)
        y_embed = (grid.cumsum(dim=0) - 0.5) / h
        x_embed = (grid.cumsum(dim=1) - 0.5) / w
        return self._pe_encoding(torch.stack([x_embed, y_embed], dim=-1)).permute(2, 0, 1)  # C x H x W


class PromptEncoder(nn.Module):
    """Constructor as called at reference ``model.py:263-268``; only the no-prompt path is ever run (``:427-429``), but every
    parameter of the fork's module exists so that ``load_state_dict(strict=True)`` accepts the checkpoint."""

    def __init__(self, embed_dim, image_embedding_size, input_image_size, mask_in_chans, activation=nn.GELU):
        super().__init__()
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(4)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans // 4), activation(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans), activation(),
            nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1))
        self.no_mask_embed = nn.Embedding(1, embed_dim)

    def get_dense_pe(self):
        return self.pe_layer(self.image_embedding_size).unsqueeze(0)

    def forward(self, points, boxes, masks):
        assert points is None and boxes is None and masks is None, "the reference only runs the no-prompt path"
        bs = 1
        sparse = torch.empty((bs, 0, self.embed_dim), device=self.no_mask_embed.weight.device)
        dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(
            bs, -1, self.image_embedding_size[0], self.image_embedding_size[1])
        return sparse, dense


class Attention(nn.Module):
    """fork ``transformer.py``: multi-head attention whose q/k/v projections shrink the width by ``downsample_rate``."""

    def __init__(self, embedding_dim, num_heads, downsample_rate=1):
        super().__init__()
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)

    def _heads(self, x):
        b, n, c = x.shape
        return x.reshape(b, n, self.num_heads, c // self.num_heads).transpose(1, 2)

    def forward(self, q, k, v):
        q, k, v = self._heads(self.q_proj(q)), self._heads(self.k_proj(k)), self._heads(self.v_proj(v))
        attn = (q @ k.permute(0, 1, 3, 2)) / math.sqrt(q.shape[-1])
        out = torch.softmax(attn, dim=-1) @ v
        b, h, n, c = out.shape
        return self.out_proj(out.transpose(1, 2).reshape(b, n, h * c))


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim, mlp_dim, act=nn.GELU):
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act()

    def forward(self, x):
        return self.lin2(self.act(self.lin1(x)))


class TwoWayAttentionBlock(nn.Module):
    def __init__(self, embedding_dim, num_heads, mlp_dim=2048, activation=nn.ReLU, attention_downsample_rate=2,
                 skip_first_layer_pe=False):
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe

    def forward(self, queries, keys, query_pe, key_pe):
        if self.skip_first_layer_pe:
            queries = self.self_attn(q=queries, k=queries, v=queries)
        else:
            q = queries + query_pe
            queries = queries + self.self_attn(q=q, k=q, v=queries)
        queries = self.norm1(queries)
        q, k = queries + query_pe, keys + key_pe
        queries = self.norm2(queries + self.cross_attn_token_to_image(q=q, k=k, v=keys))
        queries = self.norm3(queries + self.mlp(queries))
        q, k = queries + query_pe, keys + key_pe
        keys = self.norm4(keys + self.cross_attn_image_to_token(q=k, k=q, v=queries))
        return queries, keys


class TwoWayTransformer(nn.Module):
    """Constructor as called at reference ``model.py:273-278`` (depth 2, 256 wide, 8 heads, mlp 2048; ReLU MLP, attention
    downsample 2 are the fork's defaults)."""

    def __init__(self, depth, embedding_dim, num_heads, mlp_dim, activation=nn.ReLU, attention_downsample_rate=2):
        super().__init__()
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate,
                                 skip_first_layer_pe=(i == 0)) for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)

    def forward(self, image_embedding, image_pe, point_embedding):
        image_embedding = image_embedding.flatten(2).permute(0, 2, 1)   # [B, HW, C]
        image_pe = image_pe.flatten(2).permute(0, 2, 1)
        queries, keys = point_embedding, image_embedding
        for layer in self.layers:
            queries, keys = layer(queries, keys, query_pe=point_embedding, key_pe=image_pe)
        q, k = queries + point_embedding, keys + image_pe
        queries = self.norm_final_attn(queries + self.final_attn_token_to_image(q=q, k=k, v=keys))
        return queries, keys


class MLP(nn.Module):
    """fork ``mask_decoder.py``: ReLU MLP of ``num_layers`` Linear layers (hyper-networks and IoU head)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = torch.relu(layer(x)) if i < len(self.layers) - 1 else layer(x)
        return x


class MaskDecoder(nn.Module):
    """Constructor as called at reference ``model.py:271-282``."""

    def __init__(self, transformer_dim, transformer, num_multimask_outputs=3, activation=nn.GELU, iou_head_depth=3,
                 iou_head_hidden_dim=256):
        super().__init__()
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2),
            LayerNorm2d(transformer_dim // 4), activation(),
            nn.ConvTranspose2d(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2), activation())
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth)

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output):
        B = image_embeddings.shape[0]
        output_tokens = torch.cat([self.iou_token.weight, self.mask_tokens.weight], dim=0)
        tokens = torch.cat((output_tokens.unsqueeze(0).expand(sparse_prompt_embeddings.size(0), -1, -1),
                            sparse_prompt_embeddings), dim=1)
        tokens = tokens.expand(B, -1, -1)                                       # broadcast made explicit (see module docstring)
        src = image_embeddings + dense_prompt_embeddings
        b, c, h, w = src.shape
        hs, src = self.transformer(src, image_pe, tokens)
        iou_token_out = hs[:, 0, :]
        mask_tokens_out = hs[:, 1:(1 + self.num_mask_tokens), :]
        up = self.output_upscaling(src.transpose(1, 2).reshape(b, c, h, w))
        hyper_in = torch.stack([self.output_hypernetworks_mlps[i](mask_tokens_out[:, i, :])
                                for i in range(self.num_mask_tokens)], dim=1)
        b, c, h, w = up.shape
        masks = (hyper_in @ up.view(b, c, h * w)).view(b, -1, h, w)
        iou_pred = self.iou_prediction_head(iou_token_out)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, sl, :, :], iou_pred[:, sl]

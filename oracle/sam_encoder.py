"""Oracle: SAM ``ImageEncoderViT`` arithmetic (test infrastructure only).

The reference instantiates the encoder at ``model.py:245-258`` from the
un-vendored fork ``sam/segment_anything/modeling/image_encoder.py`` (absent, see
``oracle/__init__.py``).  The algorithm below restates SURVEY.md Appendix B
(patch-embed, abs pos, windowed(14)/global attention with decomposed rel-pos,
MLP with exact-erf GELU, neck) with the fork's parameter names so the reference's
checkpoint keys (``image_encoder.*``) load unchanged.  It is cross-checked
against ``transformers.models.sam.modeling_sam.SamVisionEncoder``.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


class LayerNorm2d(nn.Module):
    """Per-pixel LayerNorm over the channel axis of NCHW (fork ``common.py``;
    used at reference ``model.py:288`` and in the neck).  Biased variance."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):  # [B,3,P,P] -> [B,S,S,D]
        return self.proj(x).permute(0, 2, 3, 1)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1 = nn.Linear(dim, hidden)
        self.lin2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.lin2(F.gelu(self.lin1(x)))  # exact erf GELU


def _rel_table(rel_pos, q_size, k_size):
    """R[i, j] = rel_pos[i - j + (k-1)] (Appendix B.3).  The stored table has
    length 2*max(q,k)-1 by construction; a different length is linearly
    interpolated first (never triggers on this path)."""
    max_rel = 2 * max(q_size, k_size) - 1
    if rel_pos.shape[0] != max_rel:
        rel_pos = F.interpolate(
            rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1),
            size=max_rel, mode="linear").reshape(-1, max_rel).permute(1, 0)
    q = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    idx = (q - k) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[idx.long()]


class _Attention(nn.Module):
    def __init__(self, dim, heads, grid):
        super().__init__()
        self.heads = heads
        hd = dim // heads
        self.scale = hd ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * grid - 1, hd))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * grid - 1, hd))

    def forward(self, x):  # [B', H, W, D]
        Bp, H, W, _ = x.shape
        qkv = self.qkv(x).reshape(Bp, H * W, 3, self.heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.reshape(3, Bp * self.heads, H * W, -1).unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        # decomposed rel-pos uses the UNSCALED q (Appendix B.3)
        Rh = _rel_table(self.rel_pos_h, H, H)
        Rw = _rel_table(self.rel_pos_w, W, W)
        r_q = q.reshape(-1, H, W, q.shape[-1])
        rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
        attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None]
                + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
        attn = attn.softmax(dim=-1)
        out = (attn @ v).view(Bp, self.heads, H, W, -1).permute(0, 2, 3, 1, 4)
        return self.proj(out.reshape(Bp, H, W, -1))


class _Block(nn.Module):
    def __init__(self, dim, heads, window, grid, eps):
        super().__init__()
        self.window = window
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attention(dim, heads, grid if window == 0 else window)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, 4 * dim)

    def forward(self, x):  # [B,S,S,D]
        sc = x
        x = self.norm1(x)
        w = self.window
        if w > 0:
            B, H, W, D = x.shape
            ph, pw = (w - H % w) % w, (w - W % w) % w
            # zero pad AFTER LN1: pad tokens become q=b_q,k=b_k,v=b_v (Appendix B.4)
            x = F.pad(x, (0, 0, 0, pw, 0, ph))
            Hp, Wp = H + ph, W + pw
            x = x.view(B, Hp // w, w, Wp // w, w, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, w, w, D)
        x = self.attn(x)
        if w > 0:
            x = x.view(B, Hp // w, Wp // w, w, w, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)
            x = x[:, :H, :W, :]
        x = sc + x
        return x + self.mlp(self.norm2(x))


class ImageEncoderViT(nn.Module):
    """Same constructor meaning as the call at reference ``model.py:245-258``."""

    def __init__(self, img_size, embed_dim, depth, num_heads, global_attn_indexes,
                 patch_size=16, window_size=14, out_chans=256, eps=1e-6,
                 mlp_ratio=4, norm_layer=None, qkv_bias=True, use_rel_pos=True):
        super().__init__()
        # the remaining keyword arguments of the call at model.py:245-258; only the values the reference passes exist here
        assert mlp_ratio == 4 and qkv_bias and use_rel_pos
        if norm_layer is not None:
            eps = norm_layer(1).eps
        self.img_size = img_size
        grid = img_size // patch_size
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        self.pos_embed = nn.Parameter(torch.zeros(1, grid, grid, embed_dim))
        self.blocks = nn.ModuleList([
            _Block(embed_dim, num_heads, 0 if i in global_attn_indexes else window_size, grid, eps)
            for i in range(depth)])
        self.neck = nn.Sequential(
            nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False),
            LayerNorm2d(out_chans),
            nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
            LayerNorm2d(out_chans),
        )

    def forward(self, x, return_tokens=False):  # [B,3,P,P] -> [B,256,S,S]
        x = self.patch_embed(x) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        if return_tokens:
            return x
        return self.neck(x.permute(0, 3, 1, 2))


ARCH = {  # reference model.py:197-218
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=[2, 5, 8, 11]),
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=[5, 11, 17, 23]),
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=[7, 15, 23, 31]),
}

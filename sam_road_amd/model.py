"""`SAMRoad` — drop-in inference surface of the reference's LightningModule (reference
model.py:190-508) backed by the gfx950 HIP library (C ABI in include/samroad_hip.h).

Same constructor (`SAMRoad(config)`), same `state_dict` key names and shapes, same
three entry points with the same argument meaning:

    infer_masks_and_img_features(rgb)                       model.py:459-495
    infer_toponet(image_embeddings, graph_points, pairs, valid)   model.py:498-508
    forward(rgb, graph_points, pairs, valid)                model.py:414-457

The module tree below only HOLDS parameters (so `load_state_dict(strict=True)`, `.eval()`,
`.to(device)` behave as in the reference); no arithmetic runs in PyTorch.  There is no CPU path: a
call without the built library or without an MI355X raises.  Training (`training_step`, losses,
optimizers: model.py:349-363,511-685) is out of scope.
"""
import ctypes as C
import os
import warnings

import torch
from torch import nn

from . import _lib

ARCH = {  # model.py:197-218
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=[2, 5, 8, 11]),
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=[5, 11, 17, 23]),
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=[7, 15, 23, 31]),
}


class _LayerNorm2dParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _LoRAQkvParams(nn.Module):
    """Key layout of the reference's `_LoRA_qkv` (model.py:152-186): weight/bias keep their names and
    four rank-r adapters are added.  Folded into qkv.weight at pack time."""

    def __init__(self, dim, r):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(3 * dim, dim))
        self.bias = nn.Parameter(torch.zeros(3 * dim))
        self.linear_a_q = nn.Linear(dim, r, bias=False)
        self.linear_b_q = nn.Linear(r, dim, bias=False)
        self.linear_a_v = nn.Linear(dim, r, bias=False)
        self.linear_b_v = nn.Linear(r, dim, bias=False)
        nn.init.zeros_(self.linear_b_q.weight)
        nn.init.zeros_(self.linear_b_v.weight)


class _AttnParams(nn.Module):
    def __init__(self, dim, heads, grid, lora_rank):
        super().__init__()
        self.qkv = _LoRAQkvParams(dim, lora_rank) if lora_rank else nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * grid - 1, dim // heads))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * grid - 1, dim // heads))


class _MlpParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lin1 = nn.Linear(dim, 4 * dim)
        self.lin2 = nn.Linear(4 * dim, dim)


class _BlockParams(nn.Module):
    def __init__(self, dim, heads, grid, lora_rank):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _AttnParams(dim, heads, grid, lora_rank)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _MlpParams(dim)


class _PatchEmbedParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16)


class _ImageEncoderParams(nn.Module):
    """Parameter layout of the fork's ImageEncoderViT as constructed at model.py:245-258."""

    def __init__(self, img_size, embed_dim, depth, num_heads, global_attn_indexes, lora_rank=0):
        super().__init__()
        self.img_size = img_size  # read at model.py:439
        grid = img_size // 16
        self.patch_embed = _PatchEmbedParams(embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, grid, grid, embed_dim))
        self.blocks = nn.ModuleList([
            _BlockParams(embed_dim, num_heads, grid if i in global_attn_indexes else 14, lora_rank)
            for i in range(depth)])
        self.neck = nn.Sequential(nn.Conv2d(embed_dim, 256, 1, bias=False), _LayerNorm2dParams(256),
                                  nn.Conv2d(256, 256, 3, padding=1, bias=False), _LayerNorm2dParams(256))


class _SamAttnParams(nn.Module):
    def __init__(self, dim, downsample):
        super().__init__()
        inner = dim // downsample
        self.q_proj, self.k_proj, self.v_proj = nn.Linear(dim, inner), nn.Linear(dim, inner), nn.Linear(dim, inner)
        self.out_proj = nn.Linear(inner, dim)


class _SamTwoWayBlockParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = _SamAttnParams(256, 1)
        self.norm1 = nn.LayerNorm(256)
        self.cross_attn_token_to_image = _SamAttnParams(256, 2)
        self.norm2 = nn.LayerNorm(256)
        self.mlp = nn.Module()
        self.mlp.lin1, self.mlp.lin2 = nn.Linear(256, 2048), nn.Linear(2048, 256)
        self.norm3 = nn.LayerNorm(256)
        self.norm4 = nn.LayerNorm(256)
        self.cross_attn_image_to_token = _SamAttnParams(256, 2)


class _SamMlpParams(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))


class _SamPromptEncoderParams(nn.Module):
    """Parameter layout of the fork's PromptEncoder as constructed at model.py:263-268 (embed 256, mask_in_chans 16).  Only
    no_mask_embed and the positional-encoding matrix are read by the no-prompt path the reference runs (model.py:427-429)."""

    def __init__(self):
        super().__init__()
        self.pe_layer = nn.Module()
        self.pe_layer.register_buffer("positional_encoding_gaussian_matrix", torch.randn((2, 128)))
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, 256) for _ in range(4)])
        self.not_a_point_embed = nn.Embedding(1, 256)
        self.mask_downscaling = nn.Sequential(nn.Conv2d(1, 4, 2, stride=2), _LayerNorm2dParams(4), nn.GELU(),
                                              nn.Conv2d(4, 16, 2, stride=2), _LayerNorm2dParams(16), nn.GELU(),
                                              nn.Conv2d(16, 256, 1))
        self.no_mask_embed = nn.Embedding(1, 256)
        for p in self.parameters():
            p.requires_grad = False          # model.py:269-270


class _SamMaskDecoderParams(nn.Module):
    """Parameter layout of the fork's MaskDecoder + TwoWayTransformer as constructed at model.py:271-282."""

    def __init__(self):
        super().__init__()
        self.transformer = nn.Module()
        self.transformer.layers = nn.ModuleList([_SamTwoWayBlockParams() for _ in range(2)])
        self.transformer.final_attn_token_to_image = _SamAttnParams(256, 2)
        self.transformer.norm_final_attn = nn.LayerNorm(256)
        self.iou_token = nn.Embedding(1, 256)
        self.mask_tokens = nn.Embedding(3, 256)
        self.output_upscaling = nn.Sequential(nn.ConvTranspose2d(256, 64, 2, stride=2), _LayerNorm2dParams(64), nn.GELU(),
                                              nn.ConvTranspose2d(64, 32, 2, stride=2), nn.GELU())
        self.output_hypernetworks_mlps = nn.ModuleList([_SamMlpParams([256, 256, 256, 32]) for _ in range(3)])
        self.iou_prediction_head = _SamMlpParams([256, 256, 256, 3])


class _TopoNetParams(nn.Module):
    def __init__(self, version):
        super().__init__()
        self.feature_proj = nn.Linear(256, 128)
        self.pair_proj = nn.Linear(258, 128)
        if version != "no_transformer":
            layer = nn.TransformerEncoderLayer(d_model=128, nhead=4, dim_feedforward=128, dropout=0.1,
                                               activation="relu", batch_first=True)
            self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=3)
        self.output_proj = nn.Linear(128, 1)


class SAMRoad(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        assert config.SAM_VERSION in {"vit_b", "vit_l", "vit_h"}  # model.py:197
        if config.NO_SAM:
            raise NotImplementedError("NO_SAM ablation is not part of the release (reference model.py:232-242)")
        arch = dict(ARCH[config.SAM_VERSION])
        if config.ENCODER_DEPTH:  # test hook, not a reference key (absent => falsy => ignored)
            arch["depth"] = int(config.ENCODER_DEPTH)
            arch["global_attn_indexes"] = [int(i) for i in (config.ENCODER_GLOBAL_ATTN_INDEXES or [])]
        self.arch = arch
        self.image_size = config.PATCH_SIZE
        self.register_buffer("pixel_mean", torch.Tensor([123.675, 116.28, 103.53]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor([58.395, 57.12, 57.375]).view(-1, 1, 1), False)
        lora_rank = int(config.LORA_RANK) if config.ENCODER_LORA else 0
        self.image_encoder = _ImageEncoderParams(config.PATCH_SIZE, lora_rank=lora_rank, **arch)
        if config.USE_SAM_DECODER:  # model.py:260-282 (archived configs): SAM PromptEncoder (no prompts) + MaskDecoder
            self.prompt_encoder = _SamPromptEncoderParams()
            self.mask_decoder = _SamMaskDecoderParams()
        else:
            self.map_decoder = nn.Sequential(  # model.py:286-295 (indices 0,1,3,5,7 carry parameters)
                nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2), _LayerNorm2dParams(128), nn.GELU(),
                nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2), nn.GELU(),
                nn.ConvTranspose2d(64, 32, kernel_size=2, stride=2), nn.GELU(),
                nn.ConvTranspose2d(32, 2, kernel_size=2, stride=2))
        # a YAML without TOPONET_VERSION (toponet_vith_256 / vitl_256 / vitb_256 / vitb_1024) yields an empty Config here; the
        # reference then takes every `!=` / `==` string comparison's "normal" branch (model.py:84,111-116)
        v = config.TOPONET_VERSION
        self._topo_version = v if isinstance(v, str) else "normal"
        self.topo_net = _TopoNetParams(self._topo_version)
        self._packed = {}  # device index -> (Context, weights handle)
        self._packed_stamp = -1
        self._stamp_tensors = None
        self._imported = False                    # True: the packed weights came from another rank (import_packed)
        self._init_from_sam_checkpoint()

    # ---- init-time SAM checkpoint (model.py:365-411) ------------------------------------------------------
    def _init_from_sam_checkpoint(self):
        path = self.config.SAM_CKPT_PATH
        if not path or not os.path.exists(str(path)):
            warnings.warn(f"SAM checkpoint {path!r} not found: skipping the SAM initialisation that the "
                          "reference performs unconditionally (model.py:367); load a fine-tuned state_dict instead")
            return
        ckpt = torch.load(path, map_location="cpu")
        grid = self.image_size // 16
        if self.image_size != 1024 and ckpt["image_encoder.pos_embed"].shape[1] != grid:
            ckpt = dict(ckpt)
            pe = ckpt["image_encoder.pos_embed"].permute(0, 3, 1, 2)
            pe = nn.functional.interpolate(pe, (grid, grid), mode="bilinear", align_corners=False)
            ckpt["image_encoder.pos_embed"] = pe.permute(0, 2, 3, 1)
            # the reference selects "global" rel-pos keys by substring match on the block index
            # (model.py:403), which also catches e.g. block 17 for index 7 — reproduced as is.
            for k in [k for k in ckpt if "rel_pos" in k and any(str(i) in k for i in self.arch["global_attn_indexes"])]:
                t = ckpt[k][None, None]
                ckpt[k] = nn.functional.interpolate(t, (grid * 2 - 1, t.shape[-1]), mode="bilinear",
                                                    align_corners=False)[0, 0]
        own = dict(self.named_parameters())
        matched = {k: v for k, v in ckpt.items() if k in own and own[k].shape == v.shape}
        self.matched_param_names = set(matched)
        self.load_state_dict(matched, strict=False)

    # ---- packed weights life cycle ------------------------------------------------------------------------
    def _invalidate(self):
        for ctx, handle in self._packed.values():
            ctx.lib.srh_weights_free(handle)
        self._packed = {}
        self._stamp_tensors = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._invalidate()
        self._imported = False
        return out

    def _apply(self, fn, *args, **kwargs):
        # a model that runs on an IMPORTED arena (share_packed_weights) keeps it across an _apply that moved nothing
        # (net.to(same_device), .cuda(), .float()): its Python parameters are not what it computes with, so dropping the
        # arena here would make the next call re-pack this rank's never-loaded parameters
        before = self._stamp() if self._imported else None
        out = super()._apply(fn, *args, **kwargs)
        if self._imported:
            self._stamp_tensors = None
            if self._stamp() == before:
                return out
        self._invalidate()
        return out

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass

    def _model_cfg(self):
        """The architecture record the library packs / lays out weights for (include/samroad_hip.h srh_model_cfg)."""
        cfg = _lib.ModelCfg()
        cfg.embed_dim, cfg.depth, cfg.num_heads = self.arch["embed_dim"], self.arch["depth"], self.arch["num_heads"]
        cfg.patch_size = int(self.config.PATCH_SIZE)
        gi = list(self.arch["global_attn_indexes"])
        cfg.n_global = len(gi)
        for i, g in enumerate(gi):
            cfg.global_attn_indexes[i] = g
        cfg.window_size = 14
        cfg.toponet_version = {"no_offset": 1, "no_transformer": 2}.get(self._topo_version, 0)
        cfg.use_sam_decoder = 1 if self.config.USE_SAM_DECODER else 0
        return cfg

    def _stamp(self):
        if self._stamp_tensors is None:           # the module-tree walk costs more than the stamp: cached until _apply / load_state_dict
            self._stamp_tensors = list(self.parameters()) + list(self.buffers())
        return hash(tuple((t._version, t.data_ptr()) for t in self._stamp_tensors))

    # ---- multi-GPU: the PACKED weights travel, not the state_dict (north_star: "RCCL broadcast of weights over xGMI") -------
    def export_packed(self, device):
        """The packed weight arena of this model on `device` (fp16 MFMA operands, f32 biases / LayerNorm / pos-embed, packed
        TopoNet fragments — what srh_weights_pack built) as a uint8 tensor on that device."""
        ctx, wh = self._weights(torch.device(device))
        n = C.c_size_t(0)
        ctx.check(ctx.lib.srh_weights_export(ctx.handle, wh, None, 0, C.byref(n)), "srh_weights_export")
        buf = torch.empty(n.value, dtype=torch.uint8, device=device)
        ctx.check(ctx.lib.srh_weights_export(ctx.handle, wh, buf.data_ptr(), n.value, C.byref(n)), "srh_weights_export")
        return buf

    def import_packed(self, buf):
        """Adopt packed weights produced by export_packed of a model with the SAME configuration (on another rank).  From here on
        this model's Python parameters are NOT what it computes with; editing them raises instead of silently re-packing."""
        dev = buf.device
        if dev.type != "cuda" or buf.dtype != torch.uint8:
            raise _lib.SrhError("import_packed expects a uint8 tensor on an MI355X")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        ctx = _lib.Context.get(idx)
        self._invalidate()
        cfg = self._model_cfg()
        handle = C.c_void_p()
        with torch.cuda.device(dev):
            ctx.check(ctx.lib.srh_weights_import(ctx.handle, C.byref(cfg), buf.data_ptr(), buf.numel(), C.byref(handle)),
                      "srh_weights_import")
        self._packed[idx] = (ctx, handle)
        self._packed_stamp = self._stamp()
        self._imported = True

    def share_packed_weights(self, src=0):
        """torch.distributed (backend nccl = RCCL over xGMI): rank `src` packs its checkpoint once and broadcasts the packed
        arena device-to-device (one large collective: ~175 MB for ViT-B instead of 360 MB of f32 state_dict plus a host re-pack
        on every rank); the other ranks never read the checkpoint.  No-op without an initialised process group."""
        from . import distributed as D
        if not D.is_distributed():
            return
        dev = next(self.parameters()).device
        rank = torch.distributed.get_rank()
        buf = self.export_packed(dev) if rank == src else None
        buf = D.broadcast_bytes(buf, src=src, device=dev)
        if rank != src:
            self.import_packed(buf)

    def _weights(self, device):
        if device.type != "cuda":
            raise _lib.SrhError("SAMRoad runs on an MI355X only (tensor is on %s); there is no CPU fallback" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        # in-place parameter edits (p.data.copy_, an optimizer step, a manual LoRA merge, load_state_dict on a submodule) bump
        # the tensors' version counters: the packed fp16 copy is rebuilt instead of silently serving stale weights
        # (buffers too — the prompt encoder's Gaussian matrix is baked into the packed SAM-decoder weights — and each tensor's
        # storage address, so that `p.data = new_tensor` rebinding is seen as well)
        stamp = self._stamp()
        if stamp != self._packed_stamp:
            if self._imported:
                raise _lib.SrhError("this model computes with packed weights imported from another rank (share_packed_weights); "
                                    "its parameters were edited or moved afterwards — re-share the weights instead")
            self._invalidate()
            self._packed_stamp = stamp
        hit = self._packed.get(idx)
        if hit is not None:
            return hit
        if self._imported:      # whatever dropped the arena (or another device index asked for it): never re-pack local parameters
            raise _lib.SrhError("this model computes with packed weights imported from another rank (share_packed_weights) and has "
                                "none for cuda:%d — re-share the weights instead of re-packing this rank's parameters" % idx)
        ctx = _lib.Context.get(idx)
        sd = {k: v.detach().to(torch.float32).cpu().contiguous() for k, v in self.state_dict().items()}
        # fold LoRA adapters into the fused qkv weight (q rows [0:D], v rows [2D:3D]; model.py:179-185)
        for k in [k for k in sd if k.endswith("attn.qkv.linear_a_q.weight")]:
            base = k[: -len("linear_a_q.weight")]
            w = sd[base + "weight"].clone()
            d = w.shape[1]
            w[:d] += sd[base + "linear_b_q.weight"] @ sd[base + "linear_a_q.weight"]
            w[2 * d:] += sd[base + "linear_b_v.weight"] @ sd[base + "linear_a_v.weight"]
            sd[base + "weight"] = w
        cfg = self._model_cfg()
        names = [k for k in sd if ".linear_" not in k]
        arr = (_lib.NamedTensor * len(names))()
        keep = []
        for i, k in enumerate(names):
            t = sd[k]
            keep.append(k.encode())
            arr[i].name = keep[-1]
            arr[i].data = t.data_ptr()
            arr[i].on_device = 0
            arr[i].ndim = t.dim()
            for j, s in enumerate(t.shape):
                arr[i].shape[j] = s
        handle = C.c_void_p()
        ctx.check(ctx.lib.srh_weights_pack(ctx.handle, C.byref(cfg), arr, len(names), C.byref(handle)),
                  "srh_weights_pack")
        self._packed[idx] = (ctx, handle)
        return self._packed[idx]

    # ---- entry points -------------------------------------------------------------------------------------------
    @staticmethod
    def _stream(device):
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def _encode(self, rgb, want_logits, want_scores):
        if rgb.dim() != 4 or rgb.shape[-1] != 3 or rgb.shape[1] != self.image_size or rgb.shape[2] != self.image_size:
            raise ValueError(f"rgb must be [B,{self.image_size},{self.image_size},3], got {tuple(rgb.shape)}")
        ctx, wh = self._weights(rgb.device)
        if rgb.dtype == torch.uint8:
            dt = _lib.SRH_U8
        else:
            rgb = rgb.to(torch.float32)
            dt = _lib.SRH_F32
        rgb = rgb.contiguous()
        B, P = rgb.shape[0], self.image_size
        h = P // 16
        emb = torch.empty((B, h, h, 256), dtype=torch.float32, device=rgb.device)
        logits = torch.empty((B, P, P, 2), dtype=torch.float32, device=rgb.device) if want_logits else None
        scores = torch.empty((B, P, P, 2), dtype=torch.float32, device=rgb.device) if want_scores else None
        with torch.cuda.device(rgb.device):
            ctx.check(ctx.lib.srh_encode_decode(
                ctx.handle, wh, rgb.data_ptr(), dt, B,
                logits.data_ptr() if want_logits else None, scores.data_ptr() if want_scores else None,
                emb.data_ptr(), self._stream(rgb.device)), "srh_encode_decode")
        # reference shape [B,256,h,w]; memory stays channels-last (a permuted view, no copy)
        return logits, scores, emb.permute(0, 3, 1, 2)

    def _topo(self, image_embeddings, graph_points, pairs, valid, want_logits):
        dev = image_embeddings.device
        ctx, wh = self._weights(dev)
        emb = image_embeddings.permute(0, 2, 3, 1)
        if emb.dtype != torch.float32 or not emb.is_contiguous():
            emb = emb.to(torch.float32).contiguous()
        B, Ns, K = pairs.shape[0], pairs.shape[1], pairs.shape[2]
        N = graph_points.shape[1]
        if graph_points.dtype == torch.int64:
            pts, pdt = graph_points.contiguous(), _lib.SRH_I64
        else:
            pts, pdt = graph_points.to(torch.float32).contiguous(), _lib.SRH_F32
        if pairs.dtype == torch.int64:
            prs, qdt = pairs.contiguous(), _lib.SRH_I64
        else:
            prs, qdt = pairs.to(torch.int32).contiguous(), _lib.SRH_I32
        vld = valid.to(torch.uint8).contiguous()
        scores = torch.empty((B, Ns, K, 1), dtype=torch.float32, device=dev)
        logits = torch.empty((B, Ns, K, 1), dtype=torch.float32, device=dev) if want_logits else None
        if B * Ns * K > 0:
            with torch.cuda.device(dev):
                ctx.check(ctx.lib.srh_toponet(
                    ctx.handle, wh, emb.data_ptr(), pts.data_ptr(), pdt, prs.data_ptr(), qdt, vld.data_ptr(),
                    B, N, Ns, K, logits.data_ptr() if want_logits else None, scores.data_ptr(),
                    self._stream(dev)), "srh_toponet")
        return logits, scores

    @torch.no_grad()
    def forward(self, rgb, graph_points, pairs, valid):
        mask_logits, mask_scores, emb = self._encode(rgb, True, True)
        topo_logits, topo_scores = self._topo(emb, graph_points, pairs, valid, True)
        return mask_logits, mask_scores, topo_logits, topo_scores

    @torch.no_grad()
    def infer_masks_and_img_features(self, rgb):
        _, mask_scores, emb = self._encode(rgb, False, True)
        return mask_scores, emb

    @torch.no_grad()
    def infer_toponet(self, image_embeddings, graph_points, pairs, valid):
        return self._topo(image_embeddings, graph_points, pairs, valid, False)[1]

    @torch.no_grad()
    def infer_toponet_ragged(self, image_embeddings, points, point_tile, pairs, valid, tile_offsets=None):
        """infer_toponet over the UNPADDED query rows of many tiles at once (pass 2 of infer_one_img, reference inferencer.py:179-207,
        which pads every batch to its longest tile): image_embeddings [n,256,h,w] (the NCHW view of the library's channels-last
        buffer), points f32 [R,2] tile-local (x, y), point_tile i32 [R] (index into image_embeddings), pairs i32 [R,K,2] (rows of
        the flat list), valid u8 [R,K]  ->  scores f32 [R,K] (srh_toponet_ragged; rows as built by srh_pass2_pack_ragged).
        tile_offsets (host int64 [n + 1], rows of tile t = offsets[t] .. offsets[t+1], from 0 to R): the library then scores the scene in
        chunks of whole tiles (<= 16 k rows each, same bits) so that its workspace does not grow with the scene; without them at most
        65 536 rows per call."""
        dev = image_embeddings.device
        ctx, wh = self._weights(dev)
        emb = image_embeddings.permute(0, 2, 3, 1)
        if emb.dtype != torch.float32 or not emb.is_contiguous():
            emb = emb.to(torch.float32).contiguous()
        R, K = int(pairs.shape[0]), int(pairs.shape[1])
        pts = points.to(device=dev, dtype=torch.float32).contiguous()
        pt = point_tile.to(device=dev, dtype=torch.int32).contiguous()
        prs = pairs.to(device=dev, dtype=torch.int32).contiguous()
        vld = valid.to(device=dev, dtype=torch.uint8).contiguous()
        scores = torch.empty((R, K), dtype=torch.float32, device=dev)
        off = None
        if tile_offsets is not None:
            import numpy as np
            off = np.ascontiguousarray(tile_offsets, dtype=np.int64)
            if off.shape != (emb.shape[0] + 1,):
                raise ValueError(f"tile_offsets must have {emb.shape[0] + 1} entries (one per tile of image_embeddings + 1), got {off.shape}")
        if R * K > 0:
            with torch.cuda.device(dev):
                ctx.check(ctx.lib.srh_toponet_ragged(ctx.handle, wh, emb.data_ptr(), int(emb.shape[0]), pts.data_ptr(), pt.data_ptr(),
                                                     prs.data_ptr(), vld.data_ptr(), R, K, off.ctypes.data if off is not None else None,
                                                     scores.data_ptr(), self._stream(dev)), "srh_toponet_ragged")
        return scores

    def check_finite(self, device=None, synchronize=True):
        """Raise SrhError if a LayerNorm pass of any earlier call on this device's context saw an Inf / NaN (an fp16 overflow in the
        encoder: srh_ctx_check, SRH_ERR_NONFINITE) — the masks / embeddings of that call are invalid.  synchronize=True waits for the
        current stream first; False only looks at what has already completed (the scene loop polls right after a scene's masks reached
        the host).  The same condition is raised lazily by the next infer_* / scene_pass1 call.  The reference's own guards
        (inferencer.py:206,219) only see TopoNet's scores."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        ctx = _lib.Context.get(idx)
        with torch.cuda.device(dev):
            ctx.check(ctx.lib.srh_ctx_check(ctx.handle, self._stream(dev), 1 if synchronize else 0), "srh_ctx_check")

    # ---- scene level (pass 1 of infer_one_img: tile batcher + model + mask fusion) -------------------------
    @torch.no_grad()
    def scene_pass1(self, scene_u8, tile_xy, batch_size, canvas_kp=None, canvas_road=None):
        """scene_u8 [S,S,3] uint8 on the GPU, tile_xy int32 [n,2] (x0,y0) on the GPU.  Runs the tiles in
        batches through the encoder + decoder and accumulates the two mask canvases in the reference's
        sequential order (inferencer.py:87-104).  Returns (canvas_kp, canvas_road, embeddings[n,256,h,w])."""
        dev = scene_u8.device
        ctx, wh = self._weights(dev)
        assert scene_u8.dtype == torch.uint8 and scene_u8.dim() == 3 and scene_u8.shape[2] == 3
        scene_u8 = scene_u8.contiguous()
        tile_xy = tile_xy.to(device=dev, dtype=torch.int32).contiguous()
        S, n, h = scene_u8.shape[0], tile_xy.shape[0], self.image_size // 16
        if scene_u8.shape[1] != S:
            raise ValueError(f"scene must be square, got {tuple(scene_u8.shape)}")
        if canvas_kp is None:
            canvas_kp = torch.zeros((S, S), dtype=torch.float32, device=dev)
            canvas_road = torch.zeros((S, S), dtype=torch.float32, device=dev)
        emb = torch.empty((n, h, h, 256), dtype=torch.float32, device=dev)
        if n == 0:                                   # a rank without tiles (world_size > tile count): nothing to add
            return canvas_kp, canvas_road, emb.permute(0, 3, 1, 2)
        with torch.cuda.device(dev):
            ctx.check(ctx.lib.srh_scene_pass1(ctx.handle, wh, scene_u8.data_ptr(), S, tile_xy.data_ptr(), n,
                                              int(batch_size), canvas_kp.data_ptr(), canvas_road.data_ptr(),
                                              emb.data_ptr(), self._stream(dev)), "srh_scene_pass1")
        return canvas_kp, canvas_road, emb.permute(0, 3, 1, 2)

    @torch.no_grad()
    def scene_normalise(self, canvas_kp, canvas_road, tile_xy):
        """(canvas / coverage count) * 255 -> uint8 masks (inferencer.py:106-110); tile_xy = ALL tiles."""
        dev = canvas_kp.device
        ctx, _ = self._weights(dev)
        S = canvas_kp.shape[0]
        tile_xy = tile_xy.to(device=dev, dtype=torch.int32).contiguous()
        kp = torch.empty((S, S), dtype=torch.uint8, device=dev)
        road = torch.empty((S, S), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            ctx.check(ctx.lib.srh_scene_normalise(ctx.handle, canvas_kp.data_ptr(), canvas_road.data_ptr(), S,
                                                  tile_xy.data_ptr(), tile_xy.shape[0], self.image_size,
                                                  kp.data_ptr(), road.data_ptr(), self._stream(dev)),
                      "srh_scene_normalise")
        return kp, road

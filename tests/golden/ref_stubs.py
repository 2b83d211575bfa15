"""Stub modules that let the reference's ``model.py`` / ``inferencer.py`` be imported VERBATIM in the build container
(used only by tests/golden/make_golden_refrun.py, never by the product or at test time).

The reference needs lightning, torchmetrics, wandb, torchvision, cv2, imageio, addict, rtree, shapely, igraph, tcod, skimage
and its un-vendored SAM fork — none installed here, no network.  None of them carries arithmetic of the hot path except the
fork, which is replaced by the oracle's restatement (oracle/sam_encoder.py, oracle/sam_decoder.py — themselves
cross-checked against transformers' independent SAM implementation); everything else is plumbing:

    lightning.pytorch.LightningModule   -> torch.nn.Module (+ no-op log)
    torchmetrics.classification.*       -> parameter-less nn.Module (state is non-persistent in the reference too)
    addict.Dict                         -> attribute dict, missing key => empty falsy Dict
    rtree.index.Index                   -> brute-force closed-box point index; the id order intersection() returns is selectable
                                           (ORDER = "ascending" | "descending" | "shuffled") to probe order (in)dependence
    cv2                                 -> imread / imwrite / cvtColor / resize / line / circle via PIL + numpy
"""
import sys
import types

import numpy as np
import torch
from torch import nn

RTREE_ORDER = {"mode": "ascending"}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class Dict(dict):
    """addict.Dict behaviour the reference relies on (utils.py:6-9)."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = Dict(v) if isinstance(v, dict) and not isinstance(v, Dict) else v

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            return Dict()

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Dict) else v) for k, v in self.items()}


class _LightningModule(nn.Module):
    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass


class _Metric(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class _RtreeIndex:
    """rtree.index.Index for the one use the path makes of it (inferencer.py:126-130,150): degenerate point boxes inserted
    with consecutive ids, closed-box intersection queries."""

    def __init__(self, *a, **k):
        self.ids, self.boxes = [], []

    def insert(self, i, box):
        self.ids.append(i)
        self.boxes.append(box)

    def intersection(self, q):
        b = np.asarray(self.boxes, dtype=np.float64).reshape(-1, 4)
        x0, y0, x1, y1 = q
        hit = (b[:, 2] >= x0) & (b[:, 0] <= x1) & (b[:, 3] >= y0) & (b[:, 1] <= y1)
        ids = [self.ids[j] for j in np.nonzero(hit)[0]]
        mode = RTREE_ORDER["mode"]
        if mode == "descending":
            ids = ids[::-1]
        elif mode == "shuffled":
            rs = np.random.RandomState(len(ids) * 7919 + int(x0) * 31 + int(y0))
            ids = [ids[j] for j in rs.permutation(len(ids))]
        return iter(ids)


def _cv2_module():
    from PIL import Image, ImageDraw

    def imread(path):
        return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])      # BGR like OpenCV

    def cvtColor(img, code):
        return np.ascontiguousarray(img[:, :, ::-1])                                               # BGR2RGB == RGB2BGR

    def imwrite(path, img):
        img = np.asarray(img)
        Image.fromarray(img if img.ndim == 2 else np.ascontiguousarray(img[:, :, ::-1])).save(path)
        return True

    def resize(img, size):
        return np.array(Image.fromarray(img).resize(size, Image.BILINEAR))

    def line(img, p0, p1, color, thickness):
        im = Image.fromarray(img)
        ImageDraw.Draw(im).line([p0, p1], fill=tuple(color), width=thickness)
        img[...] = np.array(im)
        return img

    def circle(img, c, r, color, thickness):
        im = Image.fromarray(img)
        ImageDraw.Draw(im).ellipse([c[0] - r, c[1] - r, c[0] + r, c[1] + r], fill=tuple(color))
        img[...] = np.array(im)
        return img

    return dict(imread=imread, cvtColor=cvtColor, imwrite=imwrite, resize=resize, line=line, circle=circle,
                COLOR_BGR2RGB=4, COLOR_RGB2BGR=4)


def install():
    """Register the stubs in sys.modules.  Call before ``import model`` / ``import inferencer`` of the reference."""
    from oracle import sam_encoder
    pl = _mod("lightning")
    _mod("lightning.pytorch", LightningModule=_LightningModule)
    assert pl.pytorch.LightningModule is _LightningModule
    _mod("torchmetrics")
    _mod("torchmetrics.classification", BinaryJaccardIndex=_Metric, F1Score=_Metric, BinaryPrecisionRecallCurve=_Metric)
    _mod("wandb")
    _mod("torchvision")
    _mod("torchvision.ops", sigmoid_focal_loss=lambda *a, **k: None)
    _mod("sam")
    _mod("sam.segment_anything")
    _mod("sam.segment_anything.modeling")
    _mod("sam.segment_anything.modeling.image_encoder", ImageEncoderViT=sam_encoder.ImageEncoderViT)
    _mod("sam.segment_anything.modeling.common", LayerNorm2d=sam_encoder.LayerNorm2d)
    try:
        from oracle import sam_decoder
        dec = dict(MaskDecoder=sam_decoder.MaskDecoder, PromptEncoder=sam_decoder.PromptEncoder,
                   TwoWayTransformer=sam_decoder.TwoWayTransformer)
    except ImportError:
        class _Absent(nn.Module):
            def __init__(self, *a, **k):
                raise NotImplementedError("SAM MaskDecoder branch has no oracle restatement")
        dec = dict(MaskDecoder=_Absent, PromptEncoder=_Absent, TwoWayTransformer=_Absent)
    _mod("sam.segment_anything.modeling.mask_decoder", MaskDecoder=dec["MaskDecoder"])
    _mod("sam.segment_anything.modeling.prompt_encoder", PromptEncoder=dec["PromptEncoder"])
    _mod("sam.segment_anything.modeling.transformer", TwoWayTransformer=dec["TwoWayTransformer"])
    _mod("cv2", **_cv2_module())
    _mod("imageio")
    _mod("addict", Dict=Dict)
    _mod("rtree")
    _mod("rtree.index", Index=_RtreeIndex)
    _mod("shapely")
    _mod("shapely.geometry", LineString=object, Point=object)
    _mod("shapely.strtree", STRtree=object)
    _mod("igraph")
    _mod("tcod")
    _mod("skimage")
    _mod("skimage.draw", line=None)

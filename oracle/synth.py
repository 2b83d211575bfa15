"""Seeded synthetic weights / tiles / query sets shared by the oracle, the tests
and bench.py (test infrastructure only).  No checkpoint or dataset exists in the
build environment, so parity runs on synthetic ``state_dict``s fed
identically to the oracle and to the HIP path.  Recipe: SURVEY.md §8(d).
"""
import numpy as np
import torch


def synth_state_dict(model, seed=1234):
    """Fill every parameter of ``model`` (an oracle or product SAMRoad) in key
    order from one seeded generator.  Non-zero qkv bias and rel_pos (so the
    pad-key path is exercised), LN gamma ~ 1+N(0,.1), map_decoder.7.bias = -3
    (sparse masks), TopoNet with a moderately larger scale so its outputs are
    not all ~0.5."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or \
                (k.endswith(".weight") and v.dim() == 1):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif k.endswith(".bias") or k.endswith("in_proj_bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif "rel_pos" in k or k.endswith("pos_embed"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:])) if v.dim() > 1 else shape[0]
            if k.startswith("image_encoder"):
                std = 0.02
            else:  # decoder / toponet: variance-preserving-ish
                if "map_decoder" in k and v.dim() == 4:
                    fan_in = shape[0]  # ConvTranspose2d layout [Cin,Cout,kh,kw]
                std = 1.0 / np.sqrt(fan_in)
            t = (std * torch.randn(shape, generator=g)).clamp_(-2 * std, 2 * std)
        sd[k] = t.to(v.dtype)
    if "map_decoder.7.bias" in sd:
        sd["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], -3.0)
    return sd


def synth_state_dict_keyed(model, seed=1234):
    """Like synth_state_dict, but every tensor is drawn from its own generator seeded by (seed, crc32(key)), so the values do
    not depend on the order in which a module tree registers its parameters (the reference's LoRA surgery, model.py:303-347,
    re-registers qkv inside a wrapper).  LoRA B matrices (zero-initialised by the reference) are made non-zero."""
    import zlib
    sd = {}
    for k, v in model.state_dict().items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 31))
        shape = tuple(v.shape)
        if k.endswith(".weight") and v.dim() == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif k.endswith(".bias") or k.endswith("in_proj_bias") or "rel_pos" in k or k.endswith("pos_embed"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif ".linear_b_" in k or ".linear_a_" in k:
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:])) if v.dim() > 1 else shape[0]
            if "map_decoder" in k and v.dim() == 4:
                fan_in = shape[0]
            std = 0.02 if k.startswith("image_encoder") else 1.0 / np.sqrt(fan_in)
            t = (std * torch.randn(shape, generator=g)).clamp_(-2 * std, 2 * std)
        sd[k] = t.to(v.dtype)
    if "map_decoder.7.bias" in sd:
        sd["map_decoder.7.bias"] = torch.full_like(sd["map_decoder.7.bias"], -3.0)
    return sd


def synth_tiles(batch, patch, seed=0):
    """[B,P,P,3] float32 with u8 values: uniform noise low-pass filtered with an
    8-px box so LayerNorm statistics are image-like."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((batch, 3, patch + 8, patch + 8), generator=g) * 255.0
    x = torch.nn.functional.avg_pool2d(x, 8, stride=1)[:, :, :patch, :patch]
    # restore contrast lost by the box filter.  The mean is taken in float64 by numpy (one thread, pairwise): a parallel f32
    # torch sum depends on the thread count in its last bits, which moved pixels across the .5 rounding boundary — the "same"
    # seeded tile differed by one u8 level in ~0.4 % of its pixels between an 8-thread and a 16-thread host.
    mean = float(x.numpy().astype(np.float64).mean())
    x = (x - mean) * 4.0 + 127.0
    return x.clamp_(0, 255).round_().permute(0, 2, 3, 1).contiguous()


def synth_scene(size, seed=0):
    """[S,S,3] uint8 scene."""
    return synth_tiles(1, size, seed)[0].to(torch.uint8).numpy()


def synth_queries(batch, n_points, patch, k=16, radius=64.0, seed=7):
    """Points (integer pixel x,y), pairs and valid exactly as the reference's
    pass-2 builder makes them (inferencer.py:148-185): KDTree kNN k+1, radius
    bound, missing neighbour == n, invalid targets replaced by the source."""
    import scipy.spatial
    rng = np.random.default_rng(seed)
    pts_l, pairs_l, valid_l = [], [], []
    for b in range(batch):
        n = n_points if b % 3 != 2 else max(1, n_points - 5)  # ragged -> padded collate
        cand = rng.integers(0, patch, size=(n * 6, 2))
        # thin to roughly Poisson-disk spacing
        keep = []
        tree_pts = []
        for p in cand:
            if all((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 >= 12 ** 2 for q in tree_pts[-64:]):
                tree_pts.append(p)
                keep.append(p)
            if len(keep) == n:
                break
        pts = np.array(keep, dtype=np.int64).reshape(-1, 2)
        n = pts.shape[0]
        tree = scipy.spatial.KDTree(pts)
        _, idx = tree.query(pts, k=k + 1, distance_upper_bound=radius)
        idx = idx[:, 1:]
        src = np.tile(np.arange(n)[:, None], (1, k))
        valid = idx < n
        tgt = np.where(valid, idx, src)
        pts_l.append(pts)
        pairs_l.append(np.stack([src, tgt], -1))
        valid_l.append(valid)
    length = max(p.shape[0] for p in pts_l)

    def pad(x):
        return np.pad(x, [(0, length - x.shape[0])] + [(0, 0)] * (x.ndim - 1))
    points = torch.tensor(np.stack([pad(p) for p in pts_l]))
    pairs = torch.tensor(np.stack([pad(p) for p in pairs_l]))
    valid = torch.tensor(np.stack([pad(v) for v in valid_l]))
    return points, pairs, valid

"""Tile-level data parallelism over the GPUs of one node.  One process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process (inferencer.py:243); the exchange steps below are new but minimal —
the path has exactly one exchange between the two passes and one gather at the end:

    broadcast_bytes        the PACKED weight arena of rank 0 (fp16 MFMA operands, ~175 MB for ViT-B), device to device:
                           one large RCCL broadcast (xGMI rings are per-link bound, few large transfers win); the other
                           ranks never read the checkpoint or re-pack (SAMRoad.share_packed_weights)
    broadcast_state_dict   the older form: the f32 state_dict as one flat buffer (kept for callers without a packed model)
    reduce_canvases        sum of the per-rank mask canvases on rank 0 — each rank owns a contiguous chunk of the x-outer
                           tile list = a vertical band of the scene, and ships only that band
    broadcast_points       rank 0 extracts graph points on the host, everyone gets the [N,2] array
    gather_edge_votes      per-rank (src, tgt, score_sum, count) arrays -> rank 0, summed by key
"""
import numpy as np
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_state_dict(sd, src=0, device=None):
    """Broadcast an ordered {name: f32 tensor} dict as ONE flat buffer.  All ranks must hold the same
    keys/shapes (non-source ranks may hold uninitialised tensors)."""
    keys = list(sd.keys())
    sizes = [sd[k].numel() for k in keys]
    dev = device if device is not None else torch.device("cpu")
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        off = 0
        for k, n in zip(keys, sizes):
            flat[off:off + n].copy_(sd[k].reshape(-1))
            off += n
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, n in zip(keys, sizes):
        out[k] = flat[off:off + n].view(sd[k].shape).cpu() if dev.type != "cpu" else flat[off:off + n].view(sd[k].shape).clone()
        off += n
    return out


def broadcast_bytes(buf, src=0, device=None):
    """A uint8 tensor held by `src` (None elsewhere) -> the same bytes on every rank, on `device`: one size broadcast, one
    payload broadcast.  Used for the packed weight arena (SAMRoad.share_packed_weights)."""
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([buf.numel() if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if dist.get_rank() != src:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if buf.numel():
        dist.broadcast(buf, src=src)
    return buf


def tile_bands(tile_xy, patch, world):
    """Column band [x_lo, x_hi) of the scene that each rank's tile chunk (tiling.shard_tiles) touches; (0, 0) for a rank without
    tiles.  tile_xy: int array [n,2] of tile origins (x0, y0) in the reference's x-outer order."""
    from .tiling import shard_tiles
    bands = []
    for r in range(world):
        lo, hi = shard_tiles(len(tile_xy), world, r)
        if hi <= lo:
            bands.append((0, 0))
        else:
            xs = [int(tile_xy[i][0]) for i in range(lo, hi)]
            bands.append((min(xs), max(xs) + int(patch)))
    return bands


def reduce_canvases(kp, road, dst=0, bands=None):
    """In-place SUM of the two f32 scene canvases [S,S] onto `dst`.  With `bands` (tile_bands: every rank touched only the
    columns [x_lo, x_hi) of its own tile chunk) each rank ships just that band — 2 x S x (x_hi - x_lo) floats instead of
    2 x S x S: about 600 of 2048 columns per rank for the CityScale tiling on 8 GPUs — by point-to-point sends to `dst`, which
    adds the bands in rank order.  Without bands: one dense reduce.  Either way a pixel's addends are grouped per rank, so the
    last f32 bit may differ from the single-GPU order (the u8 truncation that follows is compared with +-1 level)."""
    if not is_distributed():
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    if bands is not None:
        order = [b for b in bands if b[1] > b[0]]
        if any(order[i][0] > order[i + 1][0] for i in range(len(order) - 1)):
            bands = None                # tile list not x-outer: the chunks are not vertical bands -> dense reduce
    if bands is None:
        both = torch.stack([kp, road])
        dist.reduce(both, dst=dst, op=dist.ReduceOp.SUM)
        if rank == dst:
            kp.copy_(both[0])
            road.copy_(both[1])
        return
    if rank != dst:
        x0, x1 = bands[rank]
        if _CHECK_BANDS[0]:
            # a contribution outside the band would be dropped silently (an accumulate-into canvas, a tiling that is not x-outer)
            outside = float(kp[:, :x0].abs().sum() + kp[:, x1:].abs().sum() + road[:, :x0].abs().sum() + road[:, x1:].abs().sum())
            if outside != 0.0:
                raise RuntimeError(f"reduce_canvases: rank {rank} holds canvas data outside its band [{x0}, {x1})")
        if x1 > x0:
            band = torch.stack([kp[:, x0:x1], road[:, x0:x1]]).contiguous()
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, band, dst)]):
                q.wait()
        return
    # all receives are posted at once, as ONE batched group (ncclGroupStart / End under RCCL: N - 1 un-batched irecv calls would each
    # be their own communicator operation, serialised in call order); the senders finish pass 1 at about the same time.  The bands are
    # then added in rank order, which fixes the summation order of a pixel shared by several bands
    parts, ops = {}, []
    for r in range(world):
        x0, x1 = bands[r]
        if r == dst or x1 <= x0:
            continue
        parts[r] = torch.empty((2, kp.shape[0], x1 - x0), dtype=kp.dtype, device=kp.device)
        ops.append(dist.P2POp(dist.irecv, parts[r], r))
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    for r in sorted(parts):
        x0, x1 = bands[r]
        kp[:, x0:x1] += parts[r][0]
        road[:, x0:x1] += parts[r][1]


_CHECK_BANDS = [False]       # tests switch this on: every sender checks that its canvas is zero outside its band


def canvas_bytes(bands, S, dst=0):
    """Bytes the banded reduce moves to `dst` for one scene (two f32 canvases, S rows)."""
    return sum(2 * 4 * S * max(0, x1 - x0) for r, (x0, x1) in enumerate(bands) if r != dst)


def broadcast_points(points, src=0, device=None):
    """points: int64 ndarray [N,2] on `src` (ignored elsewhere) -> same array on every rank."""
    if not is_distributed():
        return points
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([points.shape[0] if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    buf = torch.zeros((int(n.item()), 2), dtype=torch.int64, device=dev)
    if dist.get_rank() == src:
        buf.copy_(torch.as_tensor(np.ascontiguousarray(points), dtype=torch.int64))
    if buf.numel():
        dist.broadcast(buf, src=src)
    return buf.cpu().numpy()


def gather_edge_votes(keys, sums, counts, n_points, dst=0, device=None, first=None):
    """Each rank holds unique directed edge keys (src * n_points + tgt, int64) with f64 score sums, counts and (optionally)
    the local position of each key's first vote.  Returns the merged (keys, sums, counts, first) on `dst`, Nones elsewhere;
    merged `first` orders keys as one process would have first seen them (ranks own consecutive tile chunks, so the global
    visiting order is rank-major).
    Exactness: a key voted from tiles of DIFFERENT ranks is summed as (rank 0's partial sum) + (rank 1's partial sum) + ..., not
    in the reference's single visiting order, so such a sum may differ from the one-process result in its last float64 bit (an
    edge whose mean sits within an ulp of TOPO_THRESHOLD can flip).  Keys confined to one rank — and every single-process run —
    are bit-identical.  gather_raw_votes + one accumulation on `dst` (config.EXACT_VOTE_MERGE) is exact at the price of
    shipping every vote."""
    if first is None:
        first = np.zeros(keys.shape[0], dtype=np.int64)
    if not is_distributed():
        return keys, sums, counts, first
    dev = device if device is not None else torch.device("cpu")
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
    all_n = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_n, n_local)
    n_max = max(int(t.item()) for t in all_n)
    pack = torch.zeros((max(n_max, 1), 4), dtype=torch.float64, device=dev)
    if keys.shape[0]:
        pack[:keys.shape[0], 0] = torch.as_tensor(keys.astype(np.float64))   # exact below 2^53
        pack[:keys.shape[0], 1] = torch.as_tensor(sums)
        pack[:keys.shape[0], 2] = torch.as_tensor(counts)
        pack[:keys.shape[0], 3] = torch.as_tensor(first.astype(np.float64) + float(rank) * 2.0 ** 40)
    gathered = [torch.zeros_like(pack) for _ in range(world)] if rank == dst else None
    dist.gather(pack, gathered, dst=dst)
    if rank != dst:
        return None, None, None, None
    parts = [g[:int(n.item())].cpu().numpy() for g, n in zip(gathered, all_n)]
    allp = np.concatenate(parts, axis=0) if parts else np.zeros((0, 4))
    k = allp[:, 0].astype(np.int64)
    uk, inv = np.unique(k, return_inverse=True)
    # np.bincount adds in array order (rank-major, each rank's keys ascending): the same order np.add.at used, ~50x faster
    s = np.bincount(inv, weights=allp[:, 1], minlength=uk.shape[0])
    c = np.bincount(inv, weights=allp[:, 2], minlength=uk.shape[0])
    f = np.full(uk.shape[0], np.inf)
    np.minimum.at(f, inv, allp[:, 3])
    return uk, s, c, f


def gather_raw_votes(keys, scores, dst=0, device=None):
    """The exact alternative to gather_edge_votes: every rank's RAW votes (key int64, score float64, in its visiting order) are
    concatenated rank-major on `dst` — which is the one-process visiting order, ranks owning consecutive tile chunks — so that a
    single srh_edge_vote_accumulate there reproduces the one-process sums bit for bit.  Returns (keys, scores) on dst, (None,
    None) elsewhere."""
    if not is_distributed():
        return keys, scores
    dev = device if device is not None else torch.device("cpu")
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
    all_n = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_n, n_local)
    n_max = max(int(t.item()) for t in all_n)
    pack = torch.zeros((max(n_max, 1), 2), dtype=torch.float64, device=dev)
    if keys.shape[0]:
        pack[:keys.shape[0], 0] = torch.as_tensor(keys.astype(np.float64))   # exact below 2^53
        pack[:keys.shape[0], 1] = torch.as_tensor(scores)
    gathered = [torch.zeros_like(pack) for _ in range(world)] if rank == dst else None
    dist.gather(pack, gathered, dst=dst)
    if rank != dst:
        return None, None
    parts = [g[:int(n.item())].cpu().numpy() for g, n in zip(gathered, all_n)]
    allp = np.concatenate(parts, axis=0) if parts else np.zeros((0, 2))
    return np.ascontiguousarray(allp[:, 0].astype(np.int64)), np.ascontiguousarray(allp[:, 1])

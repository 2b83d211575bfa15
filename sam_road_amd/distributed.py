"""Tile-level data parallelism over the GPUs of one node.  One process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process (inferencer.py:243); the exchange steps below are new but minimal —
the path has exactly one exchange between the two passes and one gather at the end:

    broadcast_state_dict   packed into one flat buffer (one large RCCL broadcast instead of ~230 small
                           ones: xGMI rings are per-link bound, few large transfers win)
    reduce_canvases        sum of the per-rank mask canvases on rank 0 (ranks own disjoint tile chunks)
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


def reduce_canvases(kp, road, dst=0):
    """In-place SUM of the two f32 scene canvases onto `dst` (each pixel's addends come from disjoint
    tile sets per rank; the cross-rank order is the ring's, so the last f32 bit may differ from the
    single-GPU order — the u8 truncation that follows is compared with +-1 level in the tests)."""
    if not is_distributed():
        return
    both = torch.stack([kp, road])
    dist.reduce(both, dst=dst, op=dist.ReduceOp.SUM)
    if dist.get_rank() == dst:
        kp.copy_(both[0])
        road.copy_(both[1])


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
    visiting order is rank-major)."""
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

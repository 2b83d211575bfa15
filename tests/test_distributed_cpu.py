"""world_size-2 gloo tests of the N>1 exchange steps (sam_road_amd/distributed.py) on CPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sam_road_amd import distributed as D
    from sam_road_amd.tiling import shard_tiles
    try:
        # weights: one flat broadcast
        g = torch.Generator().manual_seed(3)
        ref = {"a.weight": torch.randn(5, 7, generator=g), "b.bias": torch.randn(11, generator=g)}
        sd = ref if rank == 0 else {k: torch.empty_like(v) for k, v in ref.items()}
        got = D.broadcast_state_dict(sd, src=0)
        assert all(torch.equal(got[k], ref[k]) for k in ref)
        # canvases: each rank adds its own tiles, reduce == single-process sum
        S, P = 96, 32
        tiles = [(x, y) for x in (0, 21, 43, 64) for y in (0, 21, 43, 64)]
        rng = np.random.default_rng(0)
        patches = rng.random((len(tiles), P, P, 2)).astype(np.float32)
        full_kp = np.zeros((S, S), np.float32); full_rd = np.zeros((S, S), np.float32)
        for (x, y), pch in zip(tiles, patches):
            full_kp[y:y + P, x:x + P] += pch[..., 0]; full_rd[y:y + P, x:x + P] += pch[..., 1]
        lo, hi = shard_tiles(len(tiles), world, rank)
        kp = torch.zeros((S, S)); rd = torch.zeros((S, S))
        for (x, y), pch in zip(tiles[lo:hi], patches[lo:hi]):
            kp[y:y + P, x:x + P] += torch.tensor(pch[..., 0]); rd[y:y + P, x:x + P] += torch.tensor(pch[..., 1])
        D.reduce_canvases(kp, rd, dst=0)
        if rank == 0:
            np.testing.assert_allclose(kp.numpy(), full_kp, atol=1e-5)
            np.testing.assert_allclose(rd.numpy(), full_rd, atol=1e-5)
        # points broadcast (incl. empty)
        pts = np.array([[3, 4], [50, 60], [7, 7]], dtype=np.int64)
        np.testing.assert_array_equal(D.broadcast_points(pts if rank == 0 else None, src=0), pts)
        assert D.broadcast_points(np.zeros((0, 2), np.int64) if rank == 0 else None, src=0).shape == (0, 2)
        # edge votes: overlapping keys across ranks are summed
        n_pts = 100
        if rank == 0:
            k, s, c = np.array([5, 17, 230], np.int64), np.array([0.5, 1.5, 0.25]), np.array([1.0, 2.0, 1.0])
            f = np.array([7, 0, 3], np.int64)
        else:
            k, s, c = np.array([17, 999], np.int64), np.array([0.5, 0.75]), np.array([1.0, 1.0])
            f = np.array([1, 0], np.int64)
        uk, us, uc, uf = D.gather_edge_votes(k, s, c, n_pts, dst=0, first=f)
        if rank == 0:
            np.testing.assert_array_equal(uk, [5, 17, 230, 999])
            np.testing.assert_allclose(us, [0.5, 2.0, 0.25, 0.75])
            np.testing.assert_allclose(uc, [1.0, 3.0, 1.0, 1.0])
            # first-vote order: rank-major, then the rank's local position => 17 (rank 0, pos 0), 230, 5, 999 (rank 1)
            np.testing.assert_array_equal(uk[np.argsort(uf)], [17, 230, 5, 999])
        else:
            assert uk is None
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_exchange_steps_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res

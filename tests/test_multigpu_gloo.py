"""World-size-2 CPU test of the multi-GPU path's host logic over gloo: tile ownership,
per-rank candidate lists, the all-gather exchange and the merge. The compute on each rank
is stood in for by the oracle restricted to that rank's tiles (tests may use the oracle;
on a GPU box the same tiles are walked by k_allpairs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tile_pairs(oracle, db, tiles, max_dist):
    """Oracle pairs restricted to a set of tiles (row0,row1,col0,col1)."""
    parts = []
    for row0, row1, col0, col1 in tiles:
        p = oracle.allpairs(db, max_dist, rows=(row0, row1))
        parts.append(p[(p["j"] >= col0) & (p["j"] < col1)])
    from hvd_amd._lib import PAIR_DTYPE

    return np.concatenate(parts) if parts else np.zeros(0, PAIR_DTYPE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import hvd_amd  # noqa: F401
    from hvd_amd import multigpu as M
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "hamming_db.npz"))
        db = g["db"]
        ex = M.TorchDistExchange()
        # bootstrap channel used for the RCCL unique id on a GPU box
        token = ex.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128, src=0)
        assert token == bytes(range(128))
        mine = _tile_pairs(O, db, list(M.tiles_of_rank(len(db), rank, world)), 31)
        merged = M.merge_pairs([ex.allgather_pairs(mine)])
        ok = np.array_equal(merged, g["pairs"])
        # a rank with an empty contribution must not break the exchange
        only0 = ex.allgather_pairs(mine if rank == 0 else mine[:0])
        ok = ok and len(only0) == ex_len0(dist, len(mine) if rank == 0 else 0)
        q.put((rank, bool(ok), len(mine), len(merged)))
    finally:
        dist.destroy_process_group()


def ex_len0(dist, n):
    """Sum of n over ranks (what the gathered list length must be)."""
    import torch

    t = torch.tensor([n], dtype=torch.int64)
    dist.all_reduce(t)
    return int(t.item())


@pytest.mark.timeout(300)
def test_sharded_allpairs_world2_gloo(oracle):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = len(load_golden("hamming_db.npz")["pairs"])
    assert all(ok for _, ok, _, _ in res), res
    assert sum(n for _, _, n, _ in res) == total, "ranks' candidate lists must partition the pair set"
    assert all(m == total for _, _, _, m in res)

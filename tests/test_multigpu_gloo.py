"""World-size-2 CPU tests of the multi-GPU path's host logic: tile ownership, per-rank candidate
lists, the all-gather exchange and the merge -- once over the product's own control channel
(hvd_amd.rendezvous: TCP on loopback, no torch) and once over a torch.distributed gloo group
(the launcher the driver uses). The compute on each rank is stood in for by the oracle restricted
to that rank's tiles (tests may use the oracle; on a GPU box the same tiles are walked by
k_allpairs_mfma)."""
import os
import socket
import sys

import numpy as np
import pytest

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


class GlooExchange:
    """all-gather of candidate pairs over a torch.distributed gloo group (test-side helper)."""

    def __init__(self):
        import torch.distributed as dist

        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_pairs(self, records):
        import torch

        from hvd_amd._lib import PAIR_DTYPE

        records = np.ascontiguousarray(records, dtype=PAIR_DTYPE)
        cnt = torch.tensor([records.size], dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(counts, cnt)
        counts = [int(c.item()) for c in counts]
        mx = max(counts + [1])
        send = torch.zeros(mx * 4, dtype=torch.int32)
        if records.size:
            send[: records.size * 4] = torch.from_numpy(records.view(np.int32).copy())
        recv = [torch.zeros(mx * 4, dtype=torch.int32) for _ in range(self.world)]
        self.dist.all_gather(recv, send)
        parts = [r.numpy()[: c * 4].copy().view(PAIR_DTYPE) for r, c in zip(recv, counts)]
        return np.concatenate(parts)


def _worker(rank, world, port, q, kind):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hvd_amd  # noqa: F401
    from hvd_amd import multigpu as M
    from hvd_amd.rendezvous import Rendezvous
    from oracle import oracle as O

    dist = None
    if kind == "gloo":
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        ex = GlooExchange()
        rd = None
    else:
        rd = Rendezvous(rank, world, timeout=120)
        ex = M.HostExchange(rd)
        # the control channel carries the RCCL unique id on a GPU box
        token = rd.broadcast(bytes(range(128)) if rank == 0 else None, src=0)
        assert token == bytes(range(128))
        assert rd.allreduce_max([float(rank), 1.5]) == [float(world - 1), 1.5]
        assert rd.allreduce_min([float(rank)]) == [0.0]
        rd.barrier()
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "hamming_db.npz"))
        db = g["db"]
        mine = _tile_pairs(O, db, list(M.tiles_of_rank(len(db), rank, world)), 31)
        merged = M.merge_pairs([ex.allgather_pairs(mine)])
        ok = np.array_equal(merged, g["pairs"])
        # a rank with an empty contribution must not break the exchange
        only0 = ex.allgather_pairs(mine if rank == 0 else mine[:0])
        n0 = len(mine) if rank == 0 else 0
        if rd is not None:
            total0 = sum(int.from_bytes(p, "little") for p in rd.allgather(n0.to_bytes(8, "little")))
        else:
            import torch

            t = torch.tensor([n0], dtype=torch.int64)
            dist.all_reduce(t)
            total0 = int(t.item())
        ok = ok and len(only0) == total0
        # config-5 sharding of the hashing stage: contiguous video ranges that partition the library
        from hvd_amd.pipeline import video_range_of_rank

        ranges = [video_range_of_rank(1001, r, world) for r in range(world)]
        ok = ok and ranges[0][0] == 0 and ranges[-1][1] == 1001 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        q.put((rank, bool(ok), len(mine), len(merged)))
    finally:
        if dist is not None:
            dist.destroy_process_group()
        if rd is not None:
            rd.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind,world", [("tcp", 2), ("tcp", 3), ("gloo", 2)])
def test_sharded_allpairs_multi_process(oracle, kind, world):
    port = _free_port()
    # torch is imported HERE, not at module level: pytest imports every test module at collection, and a `-m gpu` process
    # that had torch loaded would hand torch's bundled libamdhip64 / librccl to libhvd_mi355x.so (the RCCL banner of the
    # round-4 GPU run named torch's librccl for exactly that reason)
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
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


def test_product_does_not_import_torch():
    """north_star: Python host code + ctypes, no PyTorch. (tests and the launcher may use it.)"""
    import re

    pkg = os.path.join(ROOT, "hydrus-video-deduplicator_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(import|from)\s+torch\b", src, flags=re.M), f
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"^\s*(import|from)\s+torch\b", src, flags=re.M), "bench.py"

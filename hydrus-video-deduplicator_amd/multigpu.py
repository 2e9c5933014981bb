"""All-pairs search sharded over the GPUs of one node: one process per GPU.

The hash DB is replicated in every GPU's HBM (10 M hashes = 320 MB, nothing next to
288 GB); the strict upper triangle of the pair matrix is cut into tiles and tile
(rb, cb) belongs to rank (rb + cb) % world, which balances the triangle to within one
tile per tile-row. There is no data-path collective while comparing; the single exchange
step is an all-gather of each rank's candidate pairs (RCCL over xGMI, through the
C-ABI), after which every rank holds the same sorted pair list.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PAIR_DTYPE


def tile_geometry(n: int, variant: int = 9) -> tuple[int, int]:
    """(rows_per_block, col_chunk) of the all-pairs launch for n hashes (host-only)."""
    lib = _lib.load()
    r, c = C.c_uint32(0), C.c_uint32(0)
    _lib.check(lib.hvd_allpairs_tile_geometry(n, variant, C.byref(r), C.byref(c)))
    return r.value, c.value


def tile_owner(rb: int, cb: int, world: int) -> int:
    """Ownership rule implemented by the kernel (k_hamming.hip)."""
    return (rb + cb) % world


def tiles_of_rank(n: int, rank: int, world: int, variant: int = 9):
    """Yield (row0, row1, col0, col1) of the tiles rank owns that intersect the strict
    upper triangle -- the host-side statement of the kernel's tile walk."""
    rows, chunk = tile_geometry(n, variant)
    n_rb = (n + rows - 1) // rows
    n_cb = (n + chunk - 1) // chunk
    for rb in range(n_rb):
        row0 = rb * rows
        for cb in range(n_cb):
            col0, col1 = cb * chunk, min((cb + 1) * chunk, n)
            if col1 <= row0 + 1:
                continue
            if tile_owner(rb, cb, world) != rank:
                continue
            yield row0, min(row0 + rows, n), col0, col1


def rank_work_shares(n: int, world: int, variant: int = 9, tiles: bool = False) -> np.ndarray:
    """Work of every rank in one all-pairs pass over n hashes, in (row block x 128-candidate super-panel) steps -- what
    k_allpairs_mfma walks: tile (rb, cb) belongs to rank (rb + cb) % world and starts at the super-panel that holds its
    first row + 1 (candidates at or below the diagonal are skipped 128 at a time). Host arithmetic only; the exact
    statement of the load balance a sharded pass can reach. tiles=True: the number of workgroups with work per rank."""
    rows, chunk = tile_geometry(n, variant)
    n_pad = (max(n, 1) + 1023) // 1024 * 1024
    n_rb, n_cb = (n + rows - 1) // rows, (n_pad + chunk - 1) // chunk
    rb = np.arange(n_rb, dtype=np.int64)[:, None]
    cb = np.arange(n_cb, dtype=np.int64)[None, :]
    row0, col0 = rb * rows, cb * chunk
    col1 = np.minimum(col0 + chunk, n_pad)
    j0 = np.maximum(col0, (row0 + 1) // 128 * 128)
    steps = np.where(np.minimum(col1, n) <= row0 + 1, 0, np.maximum(col1 - j0, 0) // 128)
    owner = (rb + cb) % world
    if tiles:
        return np.bincount(owner.ravel(), weights=(steps.ravel() > 0).astype(np.float64), minlength=world)
    return np.bincount(owner.ravel(), weights=steps.ravel().astype(np.float64), minlength=world)


# What the first N > 1 run is to be held against: measured at N = 1 on driver-class boxes (round 6:
# BENCH line + profiles/r06_bench_kernel_stats.csv), nothing here is fitted to a multi-GPU run -- none exists.
SCALING_MODEL = {
    "kernel_ms_per_1e11_cmp": 3.62,   # 18.1 ms per 4.999995e11 comparisons (17.6 - 19.0 by box: power-limited clock)
    "expand_ms_per_1e6_hashes": 0.028,  # k_expand_fp4: every rank rebuilds the image of the WHOLE replicated DB
    "probe_and_empty_launches_ms": 0.13,  # probe 0.048 + the two unchosen forms' empty launches 0.051 + 0.027 + context 0.004
    "readback_ms": 0.03,              # pair count + this rank's records
    "exchange_ms": 0.08,              # ONE all-gather of a 16 KiB slot per rank + read-back of world x 16 KiB (RCCL, small messages)
    "host_ms": 0.10,                  # Python, ctypes, launch latency (host_ms of the N = 1 line)
    "resident_workgroups": 768,       # 3 per CU (168 VGPRs): a launch ends with about half a round of them draining
    "tail_rounds": 0.5,
}


def predict_step(n: int, world: int, model: dict | None = None, variant: int = 9) -> dict:
    """Predicted time of one all-pairs step over n hashes on `world` GPUs from the N = 1 measurements above and the exact
    tile partition: the slowest rank's kernel share + the fixed per-step pieces."""
    m = dict(SCALING_MODEL, **(model or {}))
    shares = rank_work_shares(n, world, variant)
    total_cmp = n * (n - 1) / 2.0
    worst = float(shares.max() / shares.sum()) if shares.sum() else 1.0 / world
    # the constant was measured on 1 M hashes at N = 1 (60 k workgroups = 78 rounds of the resident 768): a rank with fewer
    # workgroups pays the same half round of tail over fewer rounds
    tail = lambda wgs: 1.0 + m["tail_rounds"] * m["resident_workgroups"] / max(wgs, 1.0)  # noqa: E731
    wgs = float(rank_work_shares(n, world, variant, tiles=True).max())
    kernel = m["kernel_ms_per_1e11_cmp"] * total_cmp / 1e11 * worst * tail(wgs) / tail(60_000.0)
    fixed = (m["expand_ms_per_1e6_hashes"] * n / 1e6 + m["probe_and_empty_launches_ms"] + m["readback_ms"] + m["host_ms"] +
             (m["exchange_ms"] if world > 1 else 0.0))
    return {"n_gpus": world, "n_hashes": n, "ms_per_step": round(kernel + fixed, 3), "kernel_ms": round(kernel, 3),
            "fixed_ms": round(fixed, 3), "imbalance": round(worst * world, 4), "workgroups_of_slowest_rank": int(wgs),
            "comparisons_per_s": float(f"{total_cmp / ((kernel + fixed) * 1e-3):.4g}")}


def predict_scaling(n1: int = 1_000_000, worlds=(1, 2, 4, 8), mode: str = "weak", model: dict | None = None) -> list:
    """The curve bench.py's `--mode weak|strong` would trace: weak = n1 * sqrt(N) hashes (comparisons per GPU fixed), strong =
    n1 hashes at every N. `efficiency` = value(N) / (N x value(1))."""
    import math

    out = []
    base = None
    for w in worlds:
        n = n1 if mode == "strong" or w == 1 else int(round(n1 * math.sqrt(w) / 1024.0)) * 1024
        p = predict_step(n, w, model)
        base = p if base is None else base
        p["mode"] = mode
        p["efficiency"] = round(p["comparisons_per_s"] / (w * base["comparisons_per_s"]), 3)
        out.append(p)
    return out


def merge_pairs(parts) -> np.ndarray:
    """Concatenate per-rank records and sort by (i, j). Ranks own disjoint tiles, so there
    are no duplicates to remove; this asserts it."""
    parts = [np.asarray(p, dtype=PAIR_DTYPE) for p in parts]
    allp = np.concatenate(parts) if parts else np.zeros(0, dtype=PAIR_DTYPE)
    allp = allp[np.lexsort((allp["j"], allp["i"]))]
    if allp.size > 1:
        same = (allp["i"][1:] == allp["i"][:-1]) & (allp["j"][1:] == allp["j"][:-1])
        if same.any():
            raise AssertionError("a pair was reported by two ranks: tile ownership is not a partition")
    return allp


class RcclExchange:
    """All-gather of candidate pairs through the library's RCCL communicator."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        self.rank, self.world = rank, world
        lib = _lib.ensure()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(lib.hvd_comm_init(buf, rank, world))

    @staticmethod
    def create_unique_id() -> bytes:
        lib = _lib.load()
        buf = (C.c_uint8 * 128)()
        _lib.check(lib.hvd_comm_unique_id(buf))
        return bytes(buf)

    def allgather_pairs_dev(self, d_pairs_ptr: int, count: int) -> np.ndarray:
        lib = _lib.ensure()
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=PAIR_DTYPE)
            total = C.c_int64(0)
            rc = lib.hvd_comm_allgather_pairs(d_pairs_ptr, count, out.ctypes.data, cap, C.byref(total))
            if rc == _lib.HVD_ERR_OVERFLOW:
                cap = int(total.value)
                continue
            _lib.check(rc)
            return out[: total.value].copy()

    def allgather_bytes_dev(self, d_send_ptr: int, d_recv_ptr: int, bytes_per_rank: int) -> None:
        """RCCL all-gather of equally sized device buffers (hash shards produced on-device, config 5)."""
        _lib.check(_lib.ensure().hvd_comm_allgather_bytes(d_send_ptr, d_recv_ptr, bytes_per_rank))

    def close(self) -> None:
        _lib.check(_lib.load().hvd_comm_destroy())

    def abort(self) -> None:
        """Drop a communicator that not every rank managed to join (no collective handshake)."""
        _lib.check(_lib.load().hvd_comm_abort())


class GroupExchange(RcclExchange):
    """The exchange of the IN-PROCESS device group (hvd_init_devices / HVD_DEVICES): one thread per context calls the same
    entry points, which all-gather over the group's communicators (ncclCommInitAll) -- or through host memory where the
    group lists a device twice. Nothing to set up and nothing to tear down: the communicators belong to the library."""

    def __init__(self, rank: int, world: int):  # noqa: super().__init__ would create a communicator
        self.rank, self.world = rank, world

    def close(self) -> None:
        pass

    def abort(self) -> None:
        pass


def run_on_contexts(fn, world: int | None = None) -> list:
    """fn(rank, world) on every context of the library's device group, one Python thread per context (ctypes releases the
    GIL during the calls; each thread selects its context first). -> [fn's result per rank]; the first exception is
    re-raised after every thread has finished."""
    import threading

    world = _lib.context_count() if world is None else world
    if world <= 1:
        _lib.set_context(0)
        return [fn(0, 1)]
    # An earlier call that failed on one rank abandoned the group's exchange to release its peers (below). Nobody is inside a
    # group call now: re-arm the host barrier / re-create aborted communicators, so that one failure -- a ValueError on every
    # rank, a KeyboardInterrupt -- does not leave the group dead for the rest of the process.
    _lib.group_rearm()
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            _lib.set_context(r)
            out[r] = fn(r, world)
        except BaseException as exc:  # noqa: BLE001 - re-raised below
            err[r] = exc
            # the other ranks may be waiting for this one in an exchange step it will never reach: break the
            # group's barrier / abort its communicators, so that they fail instead of blocking join() for ever
            try:
                _lib.group_abort()
            except Exception:  # noqa: BLE001 - the original exception is what gets reported
                pass

    ts = [threading.Thread(target=body, args=(r,)) for r in range(1, world)]
    for t in ts:
        t.start()
    body(0)
    for t in ts:
        t.join()
    _lib.set_context(0)
    first = next((e for e in err if e is not None and not (isinstance(e, _lib.HvdError) and "abandoned" in str(e))), None)
    for e in ([first] if first is not None else err):  # (the rank that failed first, not a peer it released)
        if e is not None:
            raise e
    return out


class HostExchange:
    """The same all-gather of candidate pairs over the control channel (hvd_amd.rendezvous, plain TCP on the
    loopback interface): what the world_size-2 CPU tests run, and the degraded path of bench.py when the RCCL
    bootstrap fails (reported as such in its JSON). Not used when RCCL is up."""

    def __init__(self, rdzv):
        self.rdzv = rdzv
        self.rank, self.world = rdzv.rank, rdzv.world

    def allgather_pairs(self, records: np.ndarray) -> np.ndarray:
        records = np.ascontiguousarray(records, dtype=PAIR_DTYPE)
        parts = [np.frombuffer(p, dtype=PAIR_DTYPE) for p in self.rdzv.allgather(records.tobytes())]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=PAIR_DTYPE)


def connect_rccl(rdzv, timeout: float = 120.0):
    """Collective: rank 0 creates the RCCL unique id, the control channel hands it to every rank, all ranks
    join the communicator. ncclCommInitRank is bounded by `timeout`; the ranks then agree (min over ranks) on
    whether RCCL is usable. -> (RcclExchange | None, reason, stuck) -- stuck = a bootstrap thread is still
    inside RCCL and cannot be joined (the caller should leave with os._exit)."""
    import threading

    uid = rdzv.broadcast(RcclExchange.create_unique_id() if rdzv.rank == 0 else None, src=0)
    box = {}

    def _init():
        try:
            box["ex"] = RcclExchange(rdzv.rank, rdzv.world, uid)
        except Exception as exc:  # noqa: BLE001 - reported to the caller
            box["err"] = exc

    th = threading.Thread(target=_init, daemon=True)
    th.start()
    th.join(timeout=timeout)
    ok = 1.0 if ("ex" in box and not th.is_alive()) else 0.0
    if rdzv.allreduce_min([ok])[0] >= 1.0:
        return box["ex"], "rccl", False
    why = "timed out" if th.is_alive() else repr(box.get("err", "failed on another rank"))
    if "ex" in box and not th.is_alive():
        # this rank did join, another did not: the half-formed communicator must not survive -- a later
        # hvd_comm_destroy / hvd_shutdown on it can hang, and a world > 1 video search would run collectives on it
        try:
            box["ex"].abort()
        except Exception as exc:  # noqa: BLE001 - the fallback path must go on
            why += f" (abort: {exc!r})"
    return None, why, th.is_alive()


def preflight_rccl(rdzv, exchange: RcclExchange, timeout: float = 60.0):
    """Collective: one tiny all-gather of candidate pairs through the freshly formed communicator, under a deadline, and
    the ranks' verdicts combined over the control channel. -> (ok, reason, stuck). A communicator that initialises but
    cannot move data (fabric / IPC configuration) must degrade the exchange path, not hang the first real pass."""
    import threading

    box = {}

    def _go():
        try:
            d = _lib.DeviceBuffer(16)
            d.zero()
            got = exchange.allgather_pairs_dev(d.ptr, 1)
            box["ok"] = len(got) == exchange.world
            d.free()
        except Exception as exc:  # noqa: BLE001
            box["err"] = exc

    th = threading.Thread(target=_go, daemon=True)
    th.start()
    th.join(timeout)
    mine = 1.0 if (box.get("ok") and not th.is_alive()) else 0.0
    if rdzv.allreduce_min([mine])[0] >= 1.0:
        return True, "ok", False
    why = "timed out" if th.is_alive() else repr(box.get("err", "wrong record count" if "ok" in box else "failed on another rank"))
    return False, why, th.is_alive()


def launch_allpairs(lib, d_db_ptr: int, d_img_ptr: int | None, n: int, d_group_ptr, max_dist: int, rank: int,
                    world: int, d_pairs_ptr: int, cap: int, d_cnt_ptr: int, variant: int) -> None:
    """Enqueue one all-pairs pass of this rank's tiles (popcount variants 0..6 on the packed DB,
    FP4-MFMA variants 8..11 on its FP4 image)."""
    if variant >= 8:
        if d_img_ptr is None:
            raise ValueError("FP4-MFMA variants need the FP4 image (expand_fp4)")
        _lib.check(lib.hvd_dev_allpairs_hamming256_mfma(d_db_ptr, d_img_ptr, n, d_group_ptr, max_dist, rank, world,
                                                        d_pairs_ptr, cap, d_cnt_ptr, variant))
    else:
        _lib.check(lib.hvd_dev_allpairs_hamming256(d_db_ptr, n, d_group_ptr, max_dist, rank, world, d_pairs_ptr, cap,
                                                   d_cnt_ptr, variant))


def expand_fp4(d_db_ptr: int, n: int) -> "_lib.DeviceBuffer":
    """FP4 image of a DB resident in HBM (128 bytes per hash, see k_hamming_mfma.hip)."""
    lib = _lib.ensure()
    sz = C.c_size_t(0)
    _lib.check(lib.hvd_fp4_image_bytes(n, C.byref(sz)))
    d_img = _lib.DeviceBuffer(sz.value)
    _lib.check(lib.hvd_dev_expand_fp4(d_db_ptr, n, d_img.ptr))
    return d_img


def sharded_allpairs(d_db_ptr: int, n: int, rank: int, world: int, exchange: RcclExchange | None,
                     max_dist: int = 31, d_group_ptr: int | None = None, variant: int | None = None,
                     cap: int = 1 << 20) -> np.ndarray:
    """Run this rank's tiles on its GPU, exchange, return the full sorted pair list.
    The DB (and group map) must already be resident in this rank's HBM."""
    from .search import DEFAULT_VARIANT

    variant = DEFAULT_VARIANT if variant is None else variant
    lib = _lib.ensure()
    d_img = expand_fp4(d_db_ptr, n) if variant >= 8 else None
    d_pairs = _lib.DeviceBuffer(16 * cap)
    d_cnt = _lib.DeviceBuffer(8)
    try:
        while True:
            d_cnt.zero()
            launch_allpairs(lib, d_db_ptr, d_img.ptr if d_img else None, n, d_group_ptr, max_dist, rank, world,
                            d_pairs.ptr, cap, d_cnt.ptr, variant)
            count = int(d_cnt.to_array(np.uint64, 1)[0])
            if count <= cap:
                break
            cap = count  # overflow is reported, never truncated
            d_pairs.free()
            d_pairs = _lib.DeviceBuffer(16 * cap)
        if world == 1 or exchange is None:
            recs = d_pairs.to_array(PAIR_DTYPE, count)
        else:
            recs = exchange.allgather_pairs_dev(d_pairs.ptr, count)
        return merge_pairs([recs])
    finally:
        d_pairs.free()
        d_cnt.free()
        if d_img is not None:
            d_img.free()

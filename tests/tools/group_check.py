"""Run by tests/test_gpu_group.py in a process of its own (the library's device group is process state): the in-process
multi-GPU mode -- hvd_init_devices -- on whatever devices the argument lists, against the CPU oracle. `0,0` (one GPU listed
twice: two contexts, two streams, exchange through host memory) is what a 1-GPU box can run; on a node with several GPUs
the same checks run over RCCL. usage: python tests/tools/group_check.py 0,0"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hvd_amd  # noqa: E402
from hvd_amd import _lib as L, multigpu as M, pipeline, search, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (the checker)

devs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,0").split(",")]
O.build()
lib = L.init_devices(devs)
W = len(devs)
assert L.context_count() == W
assert L.group_exchange() == ("host" if len(set(devs)) < W else "rccl"), L.group_exchange()
assert L.init_devices(devs) is lib  # idempotent for the same list
try:
    L.init_devices(devs + [devs[0]])
    raise SystemExit("a second, different group was accepted")
except L.HvdError as e:
    assert e.code == L.HVD_ERR_STATE


def sorted_pairs(p):
    return p[np.lexsort((p["j"], p["i"]))]


# ---- K2 through the drop-in entry (hvd_allpairs_hamming256 fans out by itself) -------------------------------------------
db, _ = synth.hash_db(40_000, seed=61, plant_fraction=0.02)
want = O.allpairs(db, 31, num_threads=8)
assert len(want) > 500
got = hvd_amd.allpairs_hamming(db, 31)  # first try with a buffer that is too small: the true total comes back, then all of it
assert np.array_equal(got, want)
# a buffer that is too small: every context sees the same TRUE total, none truncates it, nobody is left in the exchange
small = np.zeros(10, dtype=L.PAIR_DTYPE)
cnt = C.c_int64(0)
rc = lib.hvd_allpairs_hamming256(db.ctypes.data, len(db), None, 31, small.ctypes.data, 10, C.byref(cnt))
assert rc == L.HVD_ERR_OVERFLOW and cnt.value == len(want), (rc, cnt.value, len(want))
grp = np.sort(np.random.default_rng(62).integers(0, 5000, len(db)).astype(np.int32))
assert np.array_equal(hvd_amd.allpairs_hamming(db, 31, group=grp), O.allpairs(db, 31, group=grp, num_threads=8))
# every tile belongs to exactly one context: each context's own share, launched by hand, partitions the result
parts = []


def share(rank, world):
    d_db = L.DeviceBuffer.from_array(db)
    d_img = M.expand_fp4(d_db.ptr, len(db))
    d_pairs, d_cnt = L.DeviceBuffer(16 << 16), L.DeviceBuffer(8)
    d_cnt.zero()
    M.launch_allpairs(lib, d_db.ptr, d_img.ptr, len(db), None, 31, rank, world, d_pairs.ptr, 1 << 16, d_cnt.ptr, search.DEFAULT_VARIANT)
    out = d_pairs.to_array(L.PAIR_DTYPE, int(d_cnt.to_array(np.uint64, 1)[0]))
    for b in (d_db, d_img, d_pairs, d_cnt):
        b.free()
    return out


parts = M.run_on_contexts(share)
assert len(parts) == W and all(len(p) > 0 for p in parts)
assert np.array_equal(M.merge_pairs(parts), want)  # merge_pairs asserts that no pair came twice

# ---- K1 through the drop-in entry: contiguous frame ranges per context ---------------------------------------------------
fr = synth.frames_gray(3001, seed=63)
h, q = hvd_amd.vpdq.hash_frames(fr)
wh, wq = O.hash_frames(fr)
assert np.array_equal(h, wh) and np.array_equal(q, wq)
rgb = synth.frames_rgb(9, seed=64)
h, q = hvd_amd.vpdq.hash_frames(rgb)
wh, wq = O.hash_frames(rgb)
assert np.array_equal(h, wh) and np.array_equal(q, wq)

# ---- K3: video search, symmetric and cross, key sets exchanged between the contexts --------------------------------------
frames, offsets, _ = synth.video_hashes(900, seed=65, frames_per_video=(1, 24), copy_fraction=0.2)
assert len(frames) >= 4096
wantv = O.match_videos(frames, offsets, 31)
assert len(wantv) > 50
assert np.array_equal(hvd_amd.match_videos(frames, offsets, 31), wantv)
q_sel = np.arange(0, 900, 3)
lengths = np.diff(offsets)
q_off = np.zeros(q_sel.size + 1, dtype=np.int64)
np.cumsum(lengths[q_sel], out=q_off[1:])
q_frames = np.concatenate([frames[offsets[v]:offsets[v + 1]] for v in q_sel])
gotx = search.match_videos_cross(q_frames, q_off, frames, offsets, ids_q=q_sel.astype(np.int32), ids_t=np.arange(900, dtype=np.int32))
L.set_context(0)
one = []
for qi, v in enumerate(q_sel[:60]):  # oracle on a part of the queries
    a = frames[offsets[v]:offsets[v + 1]].tobytes()
    for t in range(900):
        if t != v:
            qh, th = O.match_two(a, frames[offsets[t]:offsets[t + 1]].tobytes(), 31)
            if qh or th:
                one.append((qi, t, qh, th))
assert [r for r in gotx.tolist() if r[0] < 60] == one and len(one) > 3
# a rank that fails on its own must not strand the others in the exchange
L.set_context(W - 1)
L.check(lib.hvd_debug_set(b"vmatch_fail_rank", W))
L.set_context(0)
try:
    hvd_amd.match_videos(frames, offsets, 31)
    raise SystemExit("injected failure was not reported")
except L.HvdError as e:
    assert "rank" in str(e), str(e)
L.set_context(W - 1)
L.check(lib.hvd_debug_set(b"vmatch_fail_rank", 0))
L.set_context(0)
assert np.array_equal(hvd_amd.match_videos(frames, offsets, 31), wantv)

# ---- the drop-in surfaces above the entry points: SQLite adapter and the VpTreeManager facade on a library large enough to
#      be sharded (>= 4096 frames), against the oracle standing in for the matcher ------------------------------------------
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_sqlite_adapter import OracleMatcher, build_db  # noqa: E402
from test_vptree_facade import reference_search_loop, unordered  # noqa: E402

conn_g, blobs_g = build_db(hvd_amd, n_videos=420, seed=83)
conn_o, blobs_o = build_db(hvd_amd, n_videos=420, seed=83)
assert blobs_g == blobs_o and sum(len(b) for b in blobs_g) // 32 >= 4096
pairs_gpu, cnt_gpu = hvd_amd.sqlite_adapter.find_potential_duplicates(conn_g, 50.0)
pairs_orc, cnt_orc = hvd_amd.sqlite_adapter.find_potential_duplicates(conn_o, 50.0, matcher=OracleMatcher(O))
assert pairs_gpu == pairs_orc and cnt_gpu == cnt_orc and len(pairs_gpu) >= 5
conn_t, _ = build_db(hvd_amd, n_videos=420, seed=83)
directed, _ = reference_search_loop(conn_t, hvd_amd.vptree.VpTreeManager(conn_t), 50.0, hvd_amd)
assert unordered(conn_t, directed) == {(a, b) for a, b, _ in pairs_orc}

# ---- a streaming hasher lives on the context it was created on, whoever calls it -------------------------------------------
L.set_context(W - 1)
vh = hvd_amd.VideoHasher(1, 64, 64, 0)
L.set_context(0)
for f in fr[:700]:
    vh.hash_frame(f.tobytes())
wh, wq = O.hash_frames(fr[:700])
assert vh.finish().bytes == wh[wq >= 31].tobytes()

# ---- BASELINE config 5 chained, in process: hash shards all-gathered, video search sharded -------------------------------
V, F = 1200, 64
rng = np.random.default_rng(66)
copy_of = np.full(V, -1, dtype=np.int32)
dst = rng.choice(np.arange(V // 2, V), V // 40, replace=False)
copy_of[dst] = rng.integers(0, V // 2, dst.size)
raw_off = np.arange(V + 1, dtype=np.int64) * F
keep = []


def frames_of_rank(rank, world):
    lo, hi = pipeline.video_range_of_rank(V, rank, world)
    d_copy = L.DeviceBuffer.from_array(copy_of)
    d_fr = L.DeviceBuffer(max(1, (hi - lo) * F * 4096))
    L.check(lib.hvd_dev_synth_video_frames(d_fr.ptr, lo, hi - lo, F, 5, d_copy.ptr))
    keep.extend([d_copy, d_fr])
    return d_fr.ptr


pairs_g, recs_g = pipeline.dedupe_frames_in_process(frames_of_rank, raw_off, 64, 64, 1, 50.0)
L.set_context(0)
d_all = L.DeviceBuffer(V * F * 4096)
d_copy0 = L.DeviceBuffer.from_array(copy_of)
L.check(lib.hvd_dev_synth_video_frames(d_all.ptr, 0, V, F, 5, d_copy0.ptr))
pairs_1, recs_1, _ = pipeline.dedupe_frames_on_device(d_all.ptr, raw_off, 64, 64, 1, 50.0)
assert np.array_equal(recs_g, recs_1) and np.array_equal(pairs_g, pairs_1) and len(pairs_1) >= V // 50
planted = {(int(min(s, d)), int(max(s, d))) for d, s in enumerate(copy_of) if s >= 0}
assert planted <= {tuple(p) for p in pairs_g.tolist()}

# ---- a rank that dies BEFORE an exchange step must not strand the others, and the group must not stay dead
#      afterwards: run_on_contexts abandons the exchange (host barrier broken / communicators aborted) to release
#      the peers, and the NEXT run_on_contexts re-arms it (hvd_group_rearm: communicators re-created) -- host and RCCL groups
import threading


def _raises(fn):
    try:
        fn()
    except BaseException as exc:  # noqa: BLE001
        return exc
    return None


def one_record(rank, world):
    d = L.DeviceBuffer(16)
    d.zero()
    try:
        return M.GroupExchange(rank, world).allgather_pairs_dev(d.ptr, 1)
    finally:
        d.free()


def dies_early(rank, world):
    if rank == world - 1:
        raise ValueError("rank failed before the exchange")
    return one_record(rank, world)


for attempt in range(2):  # (twice: the recovery itself must be repeatable)
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("exc", _raises(lambda: M.run_on_contexts(dies_early))), daemon=True)
    th.start()
    th.join(120)
    assert not th.is_alive(), "the surviving ranks are still waiting for the rank that failed"
    assert isinstance(box["exc"], ValueError), repr(box["exc"])
    # a caller that drives the contexts from its own threads gets a working group back ...
    got = M.run_on_contexts(one_record)
    assert all(len(g_) == W for g_ in got), [len(g_) for g_ in got]
    assert L.group_exchange() == ("host" if len(set(devs)) < W else "rccl")
    # ... and so does the next library-driven group call
    L.set_context(0)
    assert np.array_equal(hvd_amd.allpairs_hamming(db, 31), want)

# a failure every rank leaves in lock-step (the injected local failure of the video search: reported through the agreement
# all-gather) abandons nothing: the very next sharded search runs on the same communicators, no re-arm in between
L.check(lib.hvd_debug_set(b"vmatch_fail_rank", W))
try:
    exc = _raises(lambda: hvd_amd.match_videos(frames, offsets, 31))
    assert isinstance(exc, L.HvdError), repr(exc)
finally:
    L.check(lib.hvd_debug_set(b"vmatch_fail_rank", 0))
assert np.array_equal(hvd_amd.match_videos(frames, offsets, 31), wantv)

L.shutdown()
print("GROUP_OK", devs, "exchange", "host" if len(set(devs)) < W else "rccl")

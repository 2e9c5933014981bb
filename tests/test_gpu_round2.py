"""GPU parity tests of the round-2 surface: device-side video-level reduction (K3), quality
compaction + CSR on the device, the chained BASELINE config 5 pipeline, the streaming hasher's
exact-multiple batches and zero-copy feed, comparator policy. All through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ streaming hasher ----------

@pytest.mark.parametrize("batch,n", [(5, 15), (5, 20), (5, 5), (5, 10), (1, 3), (1, 7), (4, 16)])
def test_hasher_finish_keeps_push_order_at_exact_batch_multiples(gpu, hvd, oracle, batch, n):
    """with n a multiple of the batch size (>= 3 batches) the oldest in-flight batch sits in the
    CURRENT slot; finish() must collect it first. Also batch = 1 (frames above 32 MiB get that)."""
    fr = hvd.synth.frames_gray(n, seed=81)
    ho, qo = oracle.hash_frames(fr)
    lib = gpu.load()
    hdl = C.c_void_p()
    gpu.check(lib.hvd_hasher_create(64, 64, 1, batch, C.byref(hdl)))
    try:
        for rounds in range(2):
            for f in fr:
                gpu.check(lib.hvd_hasher_push(hdl, f.ctypes.data))
            hh = np.zeros((n, 32), np.uint8)
            qq = np.zeros(n, np.int32)
            got = C.c_int64(0)
            gpu.check(lib.hvd_hasher_finish(hdl, hh.ctypes.data, qq.ctypes.data, n, C.byref(got)))
            assert got.value == n
            assert np.array_equal(qq, qo) and np.array_equal(hh, ho)
    finally:
        lib.hvd_hasher_destroy(hdl)


def test_hasher_zero_copy_acquire_commit(gpu, hvd, oracle):
    """hvd_hasher_acquire/commit: the decoder writes into the pinned slot; same result as hash_frame(bytes)."""
    fr = hvd.synth.frames_rgb(9, seed=82, h=96, w=80)
    ho, qo = oracle.hash_frames(fr)
    hasher = hvd.VideoHasher(1, 80, 96, 0, batch_bytes=fr[0].nbytes * 2)
    for f in fr:
        slot = hasher.acquire_frame(3)
        assert slot.shape == (96, 80, 3) and slot.flags.writeable
        np.copyto(slot, f)
        hasher.commit_frame()
    assert hasher.finish().bytes == ho[qo >= 31].tobytes()
    lib = gpu.load()
    hdl = C.c_void_p()
    gpu.check(lib.hvd_hasher_create(64, 64, 1, 4, C.byref(hdl)))
    try:
        assert lib.hvd_hasher_commit(hdl) == gpu.HVD_ERR_STATE  # commit without acquire
    finally:
        lib.hvd_hasher_destroy(hdl)


def test_hasher_from_another_thread_keeps_the_device(gpu, hvd, oracle):
    import threading

    fr = hvd.synth.frames_gray(30, seed=83)
    ho, qo = oracle.hash_frames(fr)
    hasher = hvd.VideoHasher(1, 64, 64, 0, batch_bytes=4096 * 4)
    box = {}

    def work():
        try:
            for f in fr:
                hasher.hash_frame(f)
            box["ph"] = hasher.finish()
        except Exception as exc:  # pragma: no cover
            box["err"] = exc

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert "err" not in box and box["ph"].bytes == ho[qo >= 31].tobytes()


# ------------------------------------------------------------------ comparator policy ---------

def test_comparator_policy_at_exactly_the_tolerance(gpu, hvd, monkeypatch):
    """A frame pair at Hamming distance exactly 31: a hit under "le", a miss under "lt" (ADVICE r1)."""
    rng = np.random.default_rng(84)
    a = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    k = np.array([31])
    b31 = hvd.synth.flip_bits(a, k, rng)
    b30 = hvd.synth.flip_bits(a, np.array([30]), rng)
    A, B31, B30 = a.tobytes(), b31.tobytes(), b30.tobytes()
    assert hvd.vpdq.MATCH_COMPARATOR == "le"
    assert hvd.matchHashBytes(A, B31, 31) == 100.0 and hvd.matchHashBytes(A, B30, 31) == 100.0
    monkeypatch.setattr(hvd.vpdq, "MATCH_COMPARATOR", "lt")
    assert hvd.matchHashBytes(A, B31, 31) == 0.0 and hvd.matchHashBytes(A, B30, 31) == 100.0
    assert hvd.matchHashBytes(A, A, 0) == 0.0  # nothing is < 0
    vh = [hvd.VpdqHash(A), hvd.VpdqHash(B31), hvd.VpdqHash(B30)]
    assert hvd.find_potential_duplicates(vh, 50.0) == [(0, 2)]  # 1-2 are 61 apart at most... and 0-1 is excluded
    monkeypatch.setattr(hvd.vpdq, "MATCH_COMPARATOR", "le")
    got = hvd.find_potential_duplicates(vh, 50.0)
    assert (0, 1) in got and (0, 2) in got


# ------------------------------------------------------------------ K3 on the device ----------

def _device_match(hvd, frames, offsets, max_dist=31, **kw):
    lib_ = hvd.pipeline.DeviceLibrary.from_host(frames, offsets)
    try:
        return lib_.match_videos(max_dist, **kw)
    finally:
        lib_.free()


@pytest.mark.parametrize("fpv", [1, 64, (0, 40), (50, 200)])
def test_k3_device_resident_vs_oracle(gpu, hvd, oracle, fpv):
    V = 300 if fpv != (50, 200) else 80
    frames, offsets, planted = hvd.synth.video_hashes(V, seed=85, frames_per_video=fpv, copy_fraction=0.1)
    want = oracle.match_videos(frames, offsets, 31)
    assert np.array_equal(_device_match(hvd, frames, offsets), want)
    assert np.array_equal(hvd.match_videos(frames, offsets, 31), want)


@pytest.mark.parametrize("max_dist", [0, 10, 31, 63, 64, 100, 127])
def test_k3_device_thresholds(gpu, hvd, oracle, max_dist):
    frames, offsets, _ = hvd.synth.video_hashes(120, seed=86, frames_per_video=(1, 30), copy_fraction=0.2, max_flips=130)
    assert np.array_equal(_device_match(hvd, frames, offsets, max_dist), oracle.match_videos(frames, offsets, max_dist))


def test_k3_two_long_near_duplicate_videos(gpu, hvd, oracle):
    """The case the host reduction could not scale to: two long videos whose frames ALL match each other
    (a static scene): n^2 frame hits, 2n distinct (frame, video) facts, ONE record back."""
    rng = np.random.default_rng(87)
    n = 3000
    base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    va = hvd.synth.flip_bits(np.repeat(base, n, 0), rng.integers(0, 8, n), rng)
    vb = hvd.synth.flip_bits(np.repeat(base, n, 0), rng.integers(0, 8, n), rng)
    other = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    frames = np.concatenate([va, other[:250], vb, other[250:]])
    offsets = np.array([0, n, n + 100, n + 250, 2 * n + 250, 2 * n + 500], dtype=np.int64)
    got = _device_match(hvd, frames, offsets)
    assert got.tolist() == [(0, 3, n, n)]
    assert np.array_equal(hvd.match_videos(frames, offsets, 31), got)
    # a partial overlap: the second half of vb replaced by noise
    frames2 = frames.copy()
    frames2[n + 250 + n // 2: 2 * n + 250] = rng.integers(0, 256, (n - n // 2, 32), dtype=np.uint8)
    assert _device_match(hvd, frames2, offsets).tolist() == [(0, 3, n, n // 2)]


def test_k3_tables_regrow_instead_of_truncating(gpu, hvd, oracle):
    frames, offsets, _ = hvd.synth.video_hashes(400, seed=88, frames_per_video=(1, 20), copy_fraction=0.5)
    want = oracle.match_videos(frames, offsets, 31)
    assert len(want) > 100
    lib = gpu.load()
    gpu.check(lib.hvd_debug_set(b"vmatch_slots_log2", 4))  # 16 slots: every table overflows and is rebuilt larger
    try:
        assert np.array_equal(_device_match(hvd, frames, offsets), want)
        assert np.array_equal(_device_match(hvd, frames, offsets, cap=3), want)  # record buffer too small: re-emit only
        assert np.array_equal(hvd.match_videos(frames, offsets, 31, cap=2), want)
    finally:
        gpu.check(lib.hvd_debug_set(b"vmatch_slots_log2", 0))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_k3_rank_tile_sets_cover_the_truth(gpu, hvd, oracle, world):
    """Kernel-side sharding of the video search: every rank's pass (exchange switched off by the debug key, so each
    returns only what its own tiles saw) reports a subset of the truth, the union over ranks is the truth's pair
    set, and per pair max-over-ranks <= truth <= sum-over-ranks (a frame can find its partner video on two ranks)."""
    frames, offsets, _ = hvd.synth.video_hashes(3000, seed=89, frames_per_video=(4, 12), copy_fraction=0.2)
    want = {(int(r["a"]), int(r["b"])): (int(r["q_hits"]), int(r["t_hits"])) for r in oracle.match_videos(frames, offsets, 31)}
    dl = hvd.pipeline.DeviceLibrary.from_host(frames, offsets)
    lib = gpu.load()
    try:
        d_cnt = gpu.DeviceBuffer(8)
        d_out = gpu.DeviceBuffer(16 * 65536)
        # world > 1 without a communicator is refused, never silently partial
        assert lib.hvd_dev_vpdq_match_videos(dl.image().ptr, dl.n_frames, dl.d_video.ptr, 31, 0, world, d_out.ptr, 65536,
                                             d_cnt.ptr) == gpu.HVD_ERR_STATE
        gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 2))
        parts = []
        for r in range(world):
            parts.append(dl.match_videos(31, rank=r, world=world))
    finally:
        gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 0))
        dl.free()
    mx, sm = {}, {}
    for p in parts:
        for r in p:
            k = (int(r["a"]), int(r["b"]))
            q, t = int(r["q_hits"]), int(r["t_hits"])
            mx[k] = (max(mx.get(k, (0, 0))[0], q), max(mx.get(k, (0, 0))[1], t))
            sm[k] = (sm.get(k, (0, 0))[0] + q, sm.get(k, (0, 0))[1] + t)
    assert set(mx) == set(want) and len(want) > 100
    for k, (q, t) in want.items():
        assert mx[k][0] <= q <= sm[k][0] and mx[k][1] <= t <= sm[k][1]
    assert sum(len(p) > 0 for p in parts) == world  # every rank owns tiles with hits


def test_k3_key_exchange_path_on_one_rank(gpu, hvd, oracle):
    """The cross-rank union (set -> list -> RCCL all-gather -> de-duplicating set) forced on with a world-1
    communicator: same records as without it."""
    frames, offsets, _ = hvd.synth.video_hashes(500, seed=92, frames_per_video=(1, 20), copy_fraction=0.3)
    want = oracle.match_videos(frames, offsets, 31)
    ex = hvd.multigpu.RcclExchange(0, 1, hvd.multigpu.RcclExchange.create_unique_id())
    lib = gpu.load()
    try:
        gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 1))
        assert np.array_equal(_device_match(hvd, frames, offsets), want)
        gpu.check(lib.hvd_debug_set(b"vmatch_slots_log2", 5))
        assert np.array_equal(_device_match(hvd, frames, offsets), want)
    finally:
        gpu.check(lib.hvd_debug_set(b"vmatch_slots_log2", 0))
        gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 0))
        ex.close()


def test_k3_cross_device_reduction_vs_oracle(gpu, hvd, oracle):
    """Query library x target library with id exclusion, against the oracle's pairwise counters."""
    frames, offsets, _ = hvd.synth.video_hashes(260, seed=90, frames_per_video=(0, 24), copy_fraction=0.3)
    q_sel = np.arange(0, 260, 3)
    lengths = np.diff(offsets)
    q_off = np.zeros(q_sel.size + 1, dtype=np.int64)
    np.cumsum(lengths[q_sel], out=q_off[1:])
    q_frames = np.concatenate([frames[offsets[v]:offsets[v + 1]] for v in q_sel])
    got = hvd.search.match_videos_cross(q_frames, q_off, frames, offsets, ids_q=q_sel.astype(np.int32),
                                        ids_t=np.arange(260, dtype=np.int32))
    want = []
    for qi, v in enumerate(q_sel):
        a = frames[offsets[v]:offsets[v + 1]].tobytes()
        for t in range(260):
            if t == v:
                continue
            b = frames[offsets[t]:offsets[t + 1]].tobytes()
            q, th = oracle.match_two(a, b, 31)
            if q or th:
                want.append((qi, t, q, th))
    assert got.tolist() == want and len(want) > 10


# ------------------------------------------------------------------ quality compaction --------

@pytest.mark.parametrize("n,V", [(0, 0), (0, 3), (1, 1), (1023, 7), (1024, 1), (5000, 300), (70000, 1500)])
def test_compact_kept_vs_numpy(gpu, hvd, n, V):
    """VideoHasher.finish for a whole library on the device: kept hashes in order, CSR, frame->video map;
    empty videos (at the start, in the middle, at the end) and all-dropped videos included."""
    rng = np.random.default_rng(91 + n)
    hashes = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    quality = rng.integers(0, 101, n).astype(np.int32)
    if n > 100:
        quality[10:60] = 0  # a video with nothing kept
    cuts = np.sort(rng.integers(0, n + 1, max(V - 1, 0)))
    raw_off = np.concatenate([[0], cuts, [n]]).astype(np.int64) if V else np.array([0], dtype=np.int64)
    if V == 0:
        raw_off = np.array([0], dtype=np.int64)
    d_h = gpu.DeviceBuffer.from_array(hashes) if n else gpu.DeviceBuffer(1)
    d_q = gpu.DeviceBuffer.from_array(quality) if n else gpu.DeviceBuffer(1)
    lib_ = hvd.pipeline.DeviceLibrary.from_raw_hashes(d_h.ptr, d_q.ptr, n, raw_off)
    keep = quality >= 31
    assert lib_.n_frames == int(keep.sum()) and lib_.n_videos == raw_off.size - 1
    assert np.array_equal(lib_.hashes(), hashes[keep])
    want_off = np.concatenate([[0], np.cumsum(keep)])[raw_off]
    assert np.array_equal(lib_.offsets(), want_off)
    vid_raw = np.searchsorted(raw_off, np.arange(n), side="right") - 1
    got_vid = lib_.d_video.to_array(np.int32, lib_.n_frames)
    assert np.array_equal(got_vid, vid_raw[keep])
    lib_.free()


# ------------------------------------------------------------------ config 5, chained ---------

def _copy_map(V, fraction, seed):
    """copy_of[v] = source video or -1; sources are never copies themselves."""
    rng = np.random.default_rng(seed)
    copy_of = np.full(V, -1, dtype=np.int32)
    m = int(round(V * fraction))
    dst = rng.choice(np.arange(1, V), size=m, replace=False)
    is_dst = np.zeros(V, dtype=bool)
    is_dst[dst] = True
    srcs = np.flatnonzero(~is_dst)
    copy_of[dst] = rng.choice(srcs, size=m)
    return copy_of


def _run_config5(gpu, hvd, oracle, V, F, sample):
    lib = gpu.load()
    copy_of = _copy_map(V, 0.02, 5)
    d_copy = gpu.DeviceBuffer.from_array(copy_of)
    n = V * F
    d_frames = gpu.DeviceBuffer(n * 4096)
    gpu.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, d_copy.ptr))
    raw_off = np.arange(V + 1, dtype=np.int64) * F
    pairs, recs, library = hvd.pipeline.dedupe_frames_on_device(d_frames.ptr, raw_off, 64, 64, 1, threshold=50.0,
                                                                keep_library=True)
    try:
        # (1) hashes + quality filter vs the oracle on a sample of whole videos (frames read back from HBM)
        rng = np.random.default_rng(6)
        vids = np.sort(rng.choice(V, size=sample // F, replace=False))
        off = library.offsets()
        kept_h = library.hashes()
        for v0 in range(0, len(vids), 64):
            vs = vids[v0:v0 + 64]
            fr = np.stack([np.frombuffer(
                gpu_read(gpu, d_frames.ptr + int(v) * F * 4096, F * 4096), dtype=np.uint8).reshape(F, 64, 64) for v in vs])
            ho, qo = oracle.hash_frames(fr.reshape(-1, 64, 64), num_threads=8)
            for k, v in enumerate(vs):
                sel = slice(k * F, (k + 1) * F)
                want = ho[sel][qo[sel] >= 31]
                assert np.array_equal(kept_h[off[v]:off[v + 1]], want), f"video {v}: kept hashes differ from the oracle"
        lengths = np.diff(off)
        assert lengths.max() <= F and lengths.sum() == library.n_frames
        assert (lengths < F).mean() > 0.3  # ~5 % constant frames are dropped: many videos lose some
        # (2) every record re-verified on the host by brute force popcount over the two videos
        got = {(int(r["a"]), int(r["b"])): (int(r["q_hits"]), int(r["t_hits"])) for r in recs}
        for (a, b), (q, t) in list(got.items())[:: max(1, len(got) // 300)]:
            A, B = kept_h[off[a]:off[a + 1]], kept_h[off[b]:off[b + 1]]
            d = np.unpackbits(A[:, None, :] ^ B[None, :, :], axis=2).sum(2)
            assert ((d <= 31).any(1).sum(), (d <= 31).any(0).sum()) == (q, t)
        # (3) planted near-copies (+-2 per pixel) are found
        planted = [(int(min(s, d)), int(max(s, d))) for d, s in enumerate(copy_of) if s >= 0]
        found = {tuple(p) for p in pairs.tolist()}
        recall = sum(p in found for p in planted) / len(planted)
        assert recall >= 0.90, recall
        # (4) the record set on a sub-library (all planted videos + as many others) equals the oracle's
        sub = np.unique(np.concatenate([np.flatnonzero(copy_of >= 0), copy_of[copy_of >= 0],
                                        rng.choice(V, size=min(V, 2 * len(planted)), replace=False)]))[:3000]
        sub_frames = np.concatenate([kept_h[off[v]:off[v + 1]] for v in sub])
        sub_off = np.concatenate([[0], np.cumsum(lengths[sub])]).astype(np.int64)
        want_sub = oracle.match_videos(sub_frames, sub_off, 31)
        idx = {int(v): k for k, v in enumerate(sub)}
        got_sub = sorted((idx[a], idx[b], q, t) for (a, b), (q, t) in got.items() if a in idx and b in idx)
        assert got_sub == [tuple(int(x) for x in r) for r in want_sub.tolist()]
        checksum = int(np.bitwise_xor.reduce(recs.view(np.uint32).astype(np.uint64) * np.arange(1, recs.size * 4 + 1,
                                                                                                 dtype=np.uint64)))
        return {"pairs": len(pairs), "records": len(recs), "recall": recall, "checksum": checksum, "kept": library.n_frames}
    finally:
        library.free()
        d_frames.free()
        d_copy.free()


def gpu_read(gpu, ptr, nbytes):
    out = np.empty(nbytes, dtype=np.uint8)
    gpu.check(gpu.load().hvd_memcpy_d2h(out.ctypes.data, C.c_void_p(ptr), nbytes))
    return out


def test_config5_chained_small(gpu, hvd, oracle):
    r = _run_config5(gpu, hvd, oracle, V=2000, F=16, sample=8000)
    assert r["pairs"] >= 30


def test_config5_chained_full_size(gpu, hvd, oracle):
    """BASELINE configs[4] on one GPU, chained and device-resident: 50 000 videos x 64 DISTINCT synthetic frames
    (13.1 GB generated in HBM) -> PDQ hashes -> quality filter + CSR -> FP4 image -> 50k x 50k video search with
    the counters reduced on the GPU. Checked: kept hashes of >= 10k frames against the oracle, records against a
    host popcount, planted-copy recall, the record set of a 3000-video sub-library against the oracle; the run is
    repeated and must reproduce its checksum."""
    r1 = _run_config5(gpu, hvd, oracle, V=50_000, F=64, sample=10_240)
    assert r1["pairs"] >= 900 and r1["kept"] > 2_500_000
    lib = gpu.load()
    copy_of = _copy_map(50_000, 0.02, 5)
    d_copy = gpu.DeviceBuffer.from_array(copy_of)
    d_frames = gpu.DeviceBuffer(50_000 * 64 * 4096)
    gpu.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, 50_000, 64, 5, d_copy.ptr))
    pairs, recs, _ = hvd.pipeline.dedupe_frames_on_device(d_frames.ptr, np.arange(50_001, dtype=np.int64) * 64, 64, 64, 1)
    checksum = int(np.bitwise_xor.reduce(recs.view(np.uint32).astype(np.uint64) * np.arange(1, recs.size * 4 + 1, dtype=np.uint64)))
    assert checksum == r1["checksum"] and len(pairs) == r1["pairs"]
    d_frames.free()
    d_copy.free()


# ------------------------------------------------------------------ K2: data-dependent kernel form --------

def _auto_form(gpu):
    v = C.c_int(0)
    gpu.check(gpu.load().hvd_debug_get(b"mfma_auto_form", C.byref(v)))
    s = C.c_int(0)
    gpu.check(gpu.load().hvd_debug_get(b"mfma_probe_survivors", C.byref(s)))
    return v.value, s.value


def test_k2_auto_variant_picks_the_form_from_the_data(gpu, hvd, oracle):
    """Variant 13 probes how often the first 128 bits of unrelated hashes agree and runs the fetch form (9) on uniform
    hashes, the register form (12) on structured ones; the pair list is the oracle's either way."""
    n = 30000
    uni, _ = hvd.synth.hash_db(n, seed=93, plant_fraction=0.01)
    assert np.array_equal(hvd.allpairs_hamming(uni, 31), oracle.allpairs(uni, 31, num_threads=8))
    form, surv = _auto_form(gpu)
    assert form == 9, (form, surv)
    # structured: the first 128 bits come from 64 prototypes, the last 128 bits are random -> the 128-bit first stage
    # passes 1/64 of all pairs, none of which is a hit
    rng = np.random.default_rng(94)
    proto = rng.integers(0, 256, (64, 16), dtype=np.uint8)
    st = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    st[:, :16] = proto[rng.integers(0, 64, n)]
    st[5000] = st[17]  # and a few real duplicates
    st[20000, :] = st[123, :]
    st[20000, 31] ^= 0x0F
    want = oracle.allpairs(st, 31, num_threads=8)
    assert len(want) >= 2
    assert np.array_equal(hvd.allpairs_hamming(st, 31), want)
    form, surv = _auto_form(gpu)
    half = C.c_int(0)
    gpu.check(gpu.load().hvd_debug_get(b"mfma_auto_half", C.byref(half)))
    # round 3: the probe also picks WHICH 128 bits the first stage sees -- here the random upper half, where survivors
    # are rare again, so the fetch form runs
    assert half.value == 1 and form == 9 and surv > 1000, (form, surv, half.value)
    # both halves structured (64 prototypes each): no half helps, the register form takes the survivors
    st2 = st.copy()
    st2[:, 16:] = rng.integers(0, 256, (64, 16), dtype=np.uint8)[rng.integers(0, 64, n)]
    st2[:, 31] ^= rng.integers(0, 256, n, dtype=np.uint8)  # ... so that not 1/4096 of all pairs are exact duplicates
    st2[5000] = st2[17]
    want2 = oracle.allpairs(st2, 31, num_threads=8, cap=1 << 22)
    assert np.array_equal(hvd.allpairs_hamming(st2, 31), want2)
    form, surv = _auto_form(gpu)
    gpu.check(gpu.load().hvd_debug_get(b"mfma_auto_half", C.byref(half)))
    # (round 5: the probe's third selection, bits 0..63 + 192..255, needs BOTH prototypes to agree -- 1/4096 of the pairs instead
    # of 1/64 -- and is what it picks; still far too many survivors for anything but the register form)
    assert form == 12 and half.value == 2 and surv > 1000, (form, surv, half.value)
    # every explicit form agrees on the structured DB too (the fetch forms go through their survivor path all the time)
    from test_gpu_parity import _run_variant

    for v in (8, 9, 12, 18):
        assert np.array_equal(_run_variant(gpu, hvd, st, v), want), v
    # video mode on structured frames
    off = np.arange(0, n + 1, 30, dtype=np.int64)
    assert np.array_equal(hvd.match_videos(st, off, 31), oracle.match_videos(st, off, 31))


def test_k1_quality_all_byte_pairs(gpu, hvd, oracle):
    """The gray quality metric's one-multiply form (k_pdq.hip grad_term_gray) through the whole hash kernel, strict and
    fma, against the oracle's reference form. (a) Every ordered pair of byte values as vertical and as horizontal
    neighbours (these frames saturate the metric at 100: a smoke test of the clamp); (b) UNSATURATED frames: a constant A
    with 45 isolated pixels B -- 180 neighbour pairs, so quality = 2 * term(A, B) exactly -- for 3000 random (A, B) and
    for every pair whose difference is a multiple of 51, where (u - v) * 100 lands on a multiple of 255 and the luma
    rounding errors decide the truncation."""
    rng = np.random.default_rng(95)
    pairs = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).reshape(-1, 2).astype(np.uint8)
    rng.shuffle(pairs)
    frames = np.zeros((32, 64, 64), dtype=np.uint8)  # 32 frames x 32 row pairs x 64 columns = 65536 vertical pairs
    frames[:, 0::2, :] = pairs[:, 0].reshape(32, 32, 64)
    frames[:, 1::2, :] = pairs[:, 1].reshape(32, 32, 64)
    ab = [(a, b) for a in range(256) for b in range(256) if a != b and abs(a - b) % 51 == 0]
    ab += [tuple(x) for x in rng.integers(0, 256, (3000, 2))]
    ab = np.array(ab, dtype=np.uint8)
    iso = np.repeat(ab[:, 0], 4096).reshape(-1, 64, 64).copy()
    ys, xs = np.meshgrid(np.arange(2, 62, 4), np.arange(2, 62, 4), indexing="ij")  # 15 x 15 interior sites, 4 apart
    sites = np.stack([ys.ravel(), xs.ravel()], 1)[:45]
    iso[:, sites[:, 0], sites[:, 1]] = ab[:, 1:2]
    both = np.concatenate([frames, frames.transpose(0, 2, 1), iso])  # transposed: the same pairs as horizontal neighbours
    ho, qo = oracle.hash_frames(both, num_threads=8)
    h, q = hvd.vpdq.hash_frames(both)
    assert np.array_equal(q, qo) and np.array_equal(h, ho)
    qi = q[64:]
    assert len(set(qi.tolist())) > 40 and qi.max() == 100 and (qi % 2 == 0).all()  # unsaturated: 2 * term
    hvd.vpdq.set_dct_mode("fma")
    try:
        hf, qf = hvd.vpdq.hash_frames(both)
        hof, qof = oracle.hash_frames(both, num_threads=8, fma=True)
        assert np.array_equal(qf, qo) and np.array_equal(qof, qo) and np.array_equal(hf, hof)
    finally:
        hvd.vpdq.set_dct_mode("strict")


def test_k2_device_api_from_two_threads(gpu, hvd, oracle):
    """The all-pairs launch writes a launch-uniform context into device memory right before the kernel that reads it;
    two host threads enqueueing passes concurrently (GUI worker + main thread) must each get their own pairs."""
    import threading

    dbs = [hvd.synth.hash_db(6000 + 500 * k, seed=96 + k, plant_fraction=0.02)[0] for k in range(2)]
    want = [oracle.allpairs(db, 31, num_threads=4) for db in dbs]
    out, errs = [None, None], []

    def work(k):
        try:
            lib = gpu.load()
            d_db = gpu.DeviceBuffer.from_array(dbs[k])
            d_img = hvd.multigpu.expand_fp4(d_db.ptr, len(dbs[k]))
            d_pairs, d_cnt = gpu.DeviceBuffer(16 * 4096), gpu.DeviceBuffer(8)
            for _ in range(30):
                d_cnt.zero()
                hvd.multigpu.launch_allpairs(lib, d_db.ptr, d_img.ptr, len(dbs[k]), None, 31, 0, 1, d_pairs.ptr, 4096,
                                             d_cnt.ptr, 13 if k == 0 else 9)
                cnt = int(d_cnt.to_array(np.uint64, 1)[0])
                got = hvd.multigpu.merge_pairs([d_pairs.to_array(gpu.PAIR_DTYPE, cnt)])
                if not np.array_equal(got, want[k]):
                    raise AssertionError(f"thread {k}: wrong pair list")
            out[k] = True
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and out == [True, True], errs


class _LoopbackAllGather:
    """All ranks of a sharded run simulated one after the other on this GPU: sweep 1 registers every rank's send
    buffers, sweep 2 serves the all-gathers from them (test-side stand-in for hvd_comm_allgather_bytes)."""

    def __init__(self, gpu, world):
        self.gpu, self.world = gpu, world
        self.sent = {}   # (call index, rank) -> DeviceBuffer copy of the send buffer
        self.rank, self.call, self.record = 0, 0, True

    def start(self, rank, record):
        self.rank, self.call, self.record = rank, 0, record

    def allgather_bytes_dev(self, d_send_ptr, d_recv_ptr, nbytes):
        lib = self.gpu.load()
        if self.record:
            keep = self.gpu.DeviceBuffer(nbytes)
            self.gpu.check(lib.hvd_memcpy_d2d(keep.ptr, d_send_ptr, nbytes))
            self.sent[(self.call, self.rank)] = keep
        else:
            for r in range(self.world):
                self.gpu.check(lib.hvd_memcpy_d2d(d_recv_ptr + r * nbytes, self.sent[(self.call, r)].ptr, nbytes))
        self.gpu.check(lib.hvd_dev_sync())
        self.call += 1


@pytest.mark.parametrize("world,V", [(2, 301), (3, 100), (8, 50), (8, 5)])
def test_config5_rank_sharding_simulated(gpu, hvd, oracle, world, V):
    """The multi-rank config-5 path (disjoint video ranges hashed per rank, padded shards all-gathered, padding squeezed
    out, whole library compacted on every rank, tiles searched per rank) with the ranks run one after the other on this
    GPU and a loopback all-gather: every rank must assemble the single-GPU library, and the ranks' partial searches must
    cover the single-GPU records (key exchange off: max <= truth <= sum per counter). V = 5 < world: ranks with no video."""
    F = 6
    lib = gpu.load()
    copy_of = np.full(V, -1, dtype=np.int32)
    if V > 10:
        copy_of[V // 2] = 1
        copy_of[V - 1] = 3
    d_copy = gpu.DeviceBuffer.from_array(copy_of)
    raw_off = np.arange(V + 1, dtype=np.int64) * F
    d_all = gpu.DeviceBuffer(V * F * 4096)
    gpu.check(lib.hvd_dev_synth_video_frames(d_all.ptr, 0, V, F, 7, d_copy.ptr))
    _, want_recs, want_lib = hvd.pipeline.dedupe_frames_on_device(d_all.ptr, raw_off, 64, 64, 1, keep_library=True)
    want_hashes, want_off = want_lib.hashes(), want_lib.offsets()
    want_lib.free()
    loop = _LoopbackAllGather(gpu, world)
    shards = {}
    for sweep in (0, 1):
        for r in range(world):
            loop.start(r, record=(sweep == 0))
            lo, hi = hvd.pipeline.video_range_of_rank(V, r, world)
            d_fr = gpu.DeviceBuffer(max(1, (hi - lo) * F * 4096))  # this rank generates only its own videos
            gpu.check(lib.hvd_dev_synth_video_frames(d_fr.ptr, lo, hi - lo, F, 7, d_copy.ptr))
            d_h, d_q = hvd.pipeline.hash_frames_on_device(d_fr.ptr, (hi - lo) * F, 64, 64, 1)
            d_fh, d_fq = hvd.pipeline.gather_hash_shards(d_h, d_q, raw_off, r, world, loop)
            if sweep == 1:
                library = hvd.pipeline.DeviceLibrary.from_raw_hashes(d_fh.ptr, d_fq.ptr, V * F, raw_off)
                assert np.array_equal(library.hashes(), want_hashes) and np.array_equal(library.offsets(), want_off)
                gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 2))
                try:
                    shards[r] = library.match_videos(rank=r, world=world)
                finally:
                    gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 0))
                library.free()
            for b in (d_fr, d_h, d_q, d_fh, d_fq):
                b.free()
    want = {(int(x["a"]), int(x["b"])): (int(x["q_hits"]), int(x["t_hits"])) for x in want_recs}
    mx, sm = {}, {}
    for part in shards.values():
        for x in part:
            k = (int(x["a"]), int(x["b"]))
            q, t = int(x["q_hits"]), int(x["t_hits"])
            mx[k] = (max(mx.get(k, (0, 0))[0], q), max(mx.get(k, (0, 0))[1], t))
            sm[k] = (sm.get(k, (0, 0))[0] + q, sm.get(k, (0, 0))[1] + t)
    assert set(mx) == set(want)
    for k, (q, t) in want.items():
        assert mx[k][0] <= q <= sm[k][0] and mx[k][1] <= t <= sm[k][1]
    if V > 10:
        assert (1, V // 2) in want and (3, V - 1) in want

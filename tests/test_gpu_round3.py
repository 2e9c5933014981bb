"""Round-3 GPU parity tests: the per-workgroup pair collection of the FP4-MFMA kernel (dense hit regimes, overflow of
the LDS buffer), the 128 -> 192 -> 256 cascade on structured hashes, the re-emit entry point, the agreement step of the
sharded video search, the parked streaming-hasher slot sets and the pinned host allocator -- all through the C-ABI,
against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sorted_pairs(p):
    return p[np.lexsort((p["j"], p["i"]))]


@pytest.mark.parametrize("variant", [9, 12, 13, 8, 18])
def test_k2_dense_clusters_overflow_the_workgroup_buffer(gpu, hvd, oracle, variant):
    """One workgroup tile holds far more than 512 hits (the LDS pair buffer): 3 clusters of 120 identical-ish hashes
    placed next to each other -> ~21k pairs inside a few tiles. Overflowing hits take the direct append; nothing may be
    lost or duplicated, distances exact."""
    rng = np.random.default_rng(41)
    db, _ = hvd.synth.hash_db(6000, seed=40)
    for c in range(3):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        rows = np.tile(base, (120, 1))
        rows = hvd.synth.flip_bits(rows, rng.integers(0, 9, 120), rng)
        db[1000 + 130 * c: 1000 + 130 * c + 120] = rows
    want = oracle.allpairs(db, 31, cap=1 << 18)
    d_db = gpu.DeviceBuffer.from_array(db)
    d_img = hvd.multigpu.expand_fp4(d_db.ptr, len(db))
    cap = 1 << 18
    d_pairs, d_cnt = gpu.DeviceBuffer(16 * cap), gpu.DeviceBuffer(8)
    d_cnt.zero()
    hvd.multigpu.launch_allpairs(gpu.load(), d_db.ptr, d_img.ptr, len(db), None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, variant)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    got = _sorted_pairs(d_pairs.to_array(gpu.PAIR_DTYPE, cnt))
    assert cnt == len(want) > 20000
    assert np.array_equal(got, want)
    # a buffer that is too small: the count is still exact, records up to cap are valid pairs
    d_cnt.zero()
    hvd.multigpu.launch_allpairs(gpu.load(), d_db.ptr, d_img.ptr, len(db), None, 31, 0, 1, d_pairs.ptr, 1000, d_cnt.ptr, variant)
    assert int(d_cnt.to_array(np.uint64, 1)[0]) == len(want)
    part = d_pairs.to_array(gpu.PAIR_DTYPE, 1000)
    truth = {(int(r["i"]), int(r["j"])): int(r["dist"]) for r in want}
    assert all(truth.get((int(r["i"]), int(r["j"]))) == int(r["dist"]) for r in part)
    assert len({(int(r["i"]), int(r["j"])) for r in part}) == 1000
    for b in (d_db, d_img, d_pairs, d_cnt):
        b.free()


def test_k2_cascade_on_structured_hashes_with_near_misses(gpu, hvd, oracle):
    """The register form's 128 -> 192 -> 256 cascade: pairs that agree on the first 128 bits but not on the rest (false
    survivors of stage one), pairs that pass 192 bits and fail at 256, and pairs exactly at the tolerance."""
    rng = np.random.default_rng(43)
    n = 5000
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = rng.choice(n // 2, 600, replace=False)
    dst = n // 2 + np.arange(600)
    db[dst] = db[src]
    kind = np.arange(600) % 6
    for k, d in zip(kind, dst):
        bits = np.unpackbits(db[d], bitorder="little")
        if k == 0:    # identical first 128 bits, random rest: survives stage one only
            bits[128:] = rng.integers(0, 2, 128)
        elif k == 1:  # identical first 192 bits, 40 flips in the last 64: survives 192, fails 256
            bits[192 + rng.choice(64, 40, replace=False)] ^= 1
        elif k == 2:  # exactly 31 flips spread over all bits: a hit at the tolerance
            bits[rng.choice(256, 31, replace=False)] ^= 1
        elif k == 3:  # exactly 32: a miss by one
            bits[rng.choice(256, 32, replace=False)] ^= 1
        elif k == 4:  # 31 flips, all in the last 64 bits
            bits[192 + rng.choice(64, 31, replace=False)] ^= 1
        else:         # 20 in the first half + 12 in the second: first stage passes (20 <= 31), total 32 misses
            bits[rng.choice(128, 20, replace=False)] ^= 1
            bits[128 + rng.choice(128, 12, replace=False)] ^= 1
        db[d] = np.packbits(bits, bitorder="little")
    want = oracle.allpairs(db, 31)
    assert len(want) == 200  # kinds 2 and 4; the other four kinds are near misses that exercise the cascade
    for variant in (12, 9, 13, 18):
        got = hvd.search.allpairs_hamming(db, 31) if variant == 13 else None
        if got is None:
            d_db = gpu.DeviceBuffer.from_array(db)
            d_img = hvd.multigpu.expand_fp4(d_db.ptr, n)
            d_pairs, d_cnt = gpu.DeviceBuffer(16 * 4096), gpu.DeviceBuffer(8)
            d_cnt.zero()
            hvd.multigpu.launch_allpairs(gpu.load(), d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, 4096, d_cnt.ptr, variant)
            got = _sorted_pairs(d_pairs.to_array(gpu.PAIR_DTYPE, int(d_cnt.to_array(np.uint64, 1)[0])))
            for b in (d_db, d_img, d_pairs, d_cnt):
                b.free()
        assert np.array_equal(got, want), variant


def test_k3_too_small_record_buffer_only_repeats_the_emit(gpu, hvd, oracle):
    """DeviceLibrary.match_videos used to re-run the whole O(n^2) pass (and, sharded, the key exchange) when
    there were more video pairs than room. Now hvd_dev_vpdq_emit_again re-emits from the pair map of the last pass."""
    frames, offsets, _ = hvd.synth.video_hashes(400, seed=95, frames_per_video=(1, 12), copy_fraction=0.5)
    want = oracle.match_videos(frames, offsets, 31)
    assert len(want) > 64
    lib_ = hvd.pipeline.DeviceLibrary.from_host(frames, offsets)
    try:
        assert np.array_equal(lib_.match_videos(31, cap=7), want)      # overflow -> re-emit
        assert np.array_equal(lib_.match_videos(31), want)
        # the entry point on its own: same records again, into a fresh buffer
        cap = len(want) + 5
        d_out, d_cnt = gpu.DeviceBuffer(16 * cap), gpu.DeviceBuffer(8)
        gpu.check(gpu.load().hvd_dev_vpdq_emit_again(d_out.ptr, cap, d_cnt.ptr))
        cnt = int(d_cnt.to_array(np.uint64, 1)[0])
        recs = d_out.to_array(gpu.VMATCH_DTYPE, cnt)
        assert np.array_equal(recs[np.lexsort((recs["b"], recs["a"]))], want)
        d_out.free()
        d_cnt.free()
    finally:
        lib_.free()


def test_k3_a_failing_rank_does_not_strand_the_exchange(gpu, hvd, oracle):
    """a rank that failed before the key all-gathers returned while its peers blocked for ever. The local
    phase's result code now rides in the first all-gather. World-1 communicator with the exchange forced on and an
    injected failure: the call must RETURN the error (having gone through the collective), and the next call must work."""
    frames, offsets, _ = hvd.synth.video_hashes(200, seed=96, frames_per_video=(1, 10), copy_fraction=0.3)
    want = oracle.match_videos(frames, offsets, 31)
    ex = hvd.multigpu.RcclExchange(0, 1, hvd.multigpu.RcclExchange.create_unique_id())
    lib = gpu.load()
    lib_ = hvd.pipeline.DeviceLibrary.from_host(frames, offsets)
    try:
        gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 1))
        gpu.check(lib.hvd_debug_set(b"vmatch_fail_rank", 1))
        with pytest.raises(gpu.HvdError, match="injected failure"):
            lib_.match_videos(31)
        gpu.check(lib.hvd_debug_set(b"vmatch_fail_rank", 0))
        assert np.array_equal(lib_.match_videos(31), want)
    finally:
        gpu.check(lib.hvd_debug_set(b"vmatch_fail_rank", 0))
        gpu.check(lib.hvd_debug_set(b"vmatch_exchange", 0))
        lib_.free()
        ex.close()


def test_comm_abort_is_idempotent_and_leaves_the_library_usable(gpu, hvd, oracle):
    ex = hvd.multigpu.RcclExchange(0, 1, hvd.multigpu.RcclExchange.create_unique_id())
    ex.abort()
    ex.abort()
    frames, offsets, _ = hvd.synth.video_hashes(50, seed=97, frames_per_video=8, copy_fraction=0.3)
    assert np.array_equal(hvd.match_videos(frames, offsets, 31), oracle.match_videos(frames, offsets, 31))
    # and a new communicator can be formed afterwards
    ex2 = hvd.multigpu.RcclExchange(0, 1, hvd.multigpu.RcclExchange.create_unique_id())
    ex2.close()


def test_videohasher_reuses_parked_slots_across_videos_and_geometries(gpu, hvd, oracle):
    """One VideoHasher per video (vpdqpy/vpdqpy.py:113): the slot sets are parked at finish() and taken over by the next
    hasher of the same geometry. Results must not depend on what the previous video left in the slots."""
    rgb = hvd.synth.frames_rgb(7, seed=50)
    gray = hvd.synth.frames_gray(300, seed=51)
    want_rgb_h, want_rgb_q = oracle.hash_frames(rgb)
    want_g_h, want_g_q = oracle.hash_frames(gray)
    for rnd in range(3):
        for nframes in (7, 3, 1):
            h = hvd.VideoHasher(1, 512, 512, 0)
            for f in rgb[:nframes]:
                h.hash_frame(f.tobytes())
            assert h.finish().bytes == want_rgb_h[:nframes][want_rgb_q[:nframes] >= 31].tobytes()
        for nframes in (300, 129, 5):
            h = hvd.VideoHasher(1, 64, 64, 0)
            for f in gray[:nframes]:
                np.copyto(h.acquire_frame(1), f)
                h.commit_frame()
            assert h.finish().bytes == want_g_h[:nframes][want_g_q[:nframes] >= 31].tobytes()
        h = hvd.VideoHasher(1, 64, 64, 0)  # a hasher that is dropped without finish() must not poison the parked set
        h.hash_frame(gray[0].tobytes())
        h.close()
    assert hvd.VideoHasher(1, 64, 64, 0).finish().bytes == b""


def test_pinned_host_allocation_round_trip(gpu):
    lib = gpu.load()
    p = C.c_void_p()
    gpu.check(lib.hvd_host_malloc(C.byref(p), 1 << 20))
    src = np.frombuffer((C.c_uint8 * (1 << 20)).from_address(p.value), dtype=np.uint8)
    src[:] = np.arange(1 << 20, dtype=np.uint32).astype(np.uint8)
    d = gpu.DeviceBuffer(1 << 20)
    gpu.check(lib.hvd_memcpy_h2d(d.ptr, p, 1 << 20))
    assert np.array_equal(d.to_array(np.uint8, 1 << 20), src)
    d.free()
    gpu.check(lib.hvd_host_free(p))


def test_k2_probe_moves_the_first_stage_to_the_more_selective_half(gpu, hvd, oracle):
    """Frame hashes are not uniform: when one half of the bits barely separates unrelated hashes (here: bits 0..127
    nearly constant across the DB), the probe puts the 128-bit first stage on the OTHER half. Same pair list either way."""
    rng = np.random.default_rng(47)
    n = 6000
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    common = rng.integers(0, 256, 16, dtype=np.uint8)
    low = np.tile(common, (n, 1))
    low = hvd.synth.flip_bits(np.concatenate([low, np.zeros((n, 16), np.uint8)], axis=1), rng.integers(0, 9, n), rng)[:, :16]
    db[:, :16] = low  # bits 0..127: a common pattern with 0..8 flips -> any two agree within 16 bits there
    src = rng.choice(n // 2, 300, replace=False)
    db[n // 2: n // 2 + 300] = hvd.synth.flip_bits(db[src], rng.integers(0, 40, 300), rng)
    want = oracle.allpairs(db, 31)
    assert 150 < len(want) < 400
    lib = gpu.load()

    def auto_half(d):
        got = hvd.search.allpairs_hamming(d, 31)
        v = C.c_int(0)
        gpu.check(lib.hvd_debug_get(b"mfma_auto_half", C.byref(v)))
        lo, hi = C.c_int(0), C.c_int(0)
        gpu.check(lib.hvd_debug_get(b"mfma_probe_survivors", C.byref(lo)))
        gpu.check(lib.hvd_debug_get(b"mfma_probe_survivors_hi", C.byref(hi)))
        return got, v.value, lo.value, hi.value

    got, half, lo, hi = auto_half(db)
    assert np.array_equal(got, want)
    assert half == 1 and lo > 100 * max(hi, 1)
    # mirrored DB (the degenerate half on top): the first stage stays on bits 0..127
    mirrored = np.ascontiguousarray(np.concatenate([db[:, 16:], db[:, :16]], axis=1))
    got_m, half_m, lo_m, hi_m = auto_half(mirrored)
    assert np.array_equal(got_m, oracle.allpairs(mirrored, 31)) and half_m == 0 and hi_m > 100 * max(lo_m, 1)
    # uniform DB: no preference, bits 0..127
    uni, _ = hvd.synth.hash_db(6000, seed=48)
    got_u, half_u, _, _ = auto_half(uni)
    assert half_u == 0 and np.array_equal(got_u, oracle.allpairs(uni, 31))
    # and the video-level search on the degenerate library
    off = np.arange(0, n + 1, 20, dtype=np.int64)
    assert np.array_equal(hvd.match_videos(db, off, 31), oracle.match_videos(db, off, 31))


def test_rccl_preflight_passes_on_a_world_1_communicator(gpu, hvd):
    """bench.py's guard in front of the first real exchange: one 16-byte all-gather under a deadline."""
    from hvd_amd.rendezvous import Rendezvous

    ex = hvd.multigpu.RcclExchange(0, 1, hvd.multigpu.RcclExchange.create_unique_id())
    try:
        ok, why, stuck = hvd.multigpu.preflight_rccl(Rendezvous(0, 1), ex, timeout=60.0)
        assert ok and not stuck, why
    finally:
        ex.close()

"""GPU parity tests (run with -m gpu on an MI355X): every HIP path, called through the
C-ABI, against the CPU oracle on the same seeded inputs and against the committed golden
fixtures. Bit-exact: hashes, qualities, pair lists and match counters are integers/bytes."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ K1: PDQ hashing --

def test_native_library_is_loaded_and_on_gfx950(gpu):
    assert gpu.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libhvd_mi355x.so" in maps


def test_dct_matrix_on_box_matches_oracle(gpu, oracle):
    d = np.zeros((16, 64), np.float32)
    gpu.check(gpu.load().hvd_dct_matrix(d.ctypes.data))
    assert np.array_equal(d.view(np.uint32), oracle.dct_matrix().view(np.uint32))


def test_k1_golden_gray64(gpu, hvd):
    g = load_golden("pdq_gray64.npz")
    h, q = hvd.vpdq.hash_frames(g["frames"])
    assert np.array_equal(h, g["hashes"])
    assert np.array_equal(q, g["quality"])


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 10000])
def test_k1_gray64_vs_oracle(gpu, hvd, oracle, n):
    """BASELINE config 2 at n=10000; ragged counts exercise the partial last workgroup."""
    fr = hvd.synth.frames_gray(n, seed=2)
    h, q = hvd.vpdq.hash_frames(fr)
    ho, qo = oracle.hash_frames(fr, num_threads=8)
    assert np.array_equal(q, qo), f"{int((q != qo).sum())} quality mismatches"
    assert np.array_equal(h, ho), f"{int((h != ho).any(1).sum())} hash mismatches"


def test_k1_gray64_extreme_frames(gpu, hvd, oracle):
    rng = np.random.default_rng(11)
    fr = np.stack([
        np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8),
        rng.integers(0, 256, (64, 64), dtype=np.uint8),                 # white noise
        (rng.integers(0, 2, (64, 64)) * 255).astype(np.uint8),          # binary noise, max gradients
        np.tile(np.array([0, 255], np.uint8), (64, 32)),                # column stripes
        np.tile(np.array([[0], [255]], np.uint8), (32, 64)),            # row stripes
        np.tri(64, dtype=np.uint8) * 200,
    ])
    h, q = hvd.vpdq.hash_frames(fr)
    ho, qo = oracle.hash_frames(fr)
    assert np.array_equal(q, qo) and np.array_equal(h, ho)


@pytest.fixture
def fma_mode(hvd):
    hvd.vpdq.set_dct_mode("fma")
    try:
        yield
    finally:
        hvd.vpdq.set_dct_mode("strict")


def test_k1_fma_mode_golden(gpu, hvd, fma_mode):
    """Opt-in DCT mode on the matrix cores (v_mfma_f32_16x16x4_f32 = an fmaf chain): bit-exact against
    the frozen fma-mode vectors, 64x64 and through the down-sampler."""
    assert hvd.vpdq.get_dct_mode() == "fma"
    g = load_golden("pdq_gray64.npz")
    h, q = hvd.vpdq.hash_frames(g["frames"])
    assert np.array_equal(q, g["quality"])
    assert np.array_equal(h, g["hashes_fma"]), f"{int((h != g['hashes_fma']).any(1).sum())} hash mismatches"
    r = load_golden("pdq_rgb512.npz")
    h, q = hvd.vpdq.hash_frames(r["frames"])
    assert np.array_equal(h, r["hashes_fma"]) and np.array_equal(q, r["quality"])


@pytest.mark.parametrize("n", [1, 5, 1023, 10000])
def test_k1_fma_mode_vs_oracle(gpu, hvd, oracle, fma_mode, n):
    fr = hvd.synth.frames_gray(n, seed=21)
    h, q = hvd.vpdq.hash_frames(fr)
    ho, qo = oracle.hash_frames(fr, num_threads=8, fma=True)
    assert np.array_equal(q, qo)
    assert np.array_equal(h, ho), f"{int((h != ho).any(1).sum())} hash mismatches"
    rgb = hvd.synth.frames_rgb(min(n, 40), seed=22, h=64, w=64)
    h, q = hvd.vpdq.hash_frames(rgb)
    ho, qo = oracle.hash_frames(rgb, fma=True)
    assert np.array_equal(h, ho) and np.array_equal(q, qo)


def test_k1_dct_mode_default_is_strict_and_modes_stay_close(gpu, hvd, oracle):
    assert hvd.vpdq.get_dct_mode() == "strict"
    with pytest.raises(ValueError):
        hvd.vpdq.set_dct_mode("fast")
    assert hvd._lib.load().hvd_set_pdq_dct_mode(7) == -1
    fr = hvd.synth.frames_gray(2000, seed=23)
    hs, qs = hvd.vpdq.hash_frames(fr)
    hvd.vpdq.set_dct_mode("fma")
    try:
        hf, qf = hvd.vpdq.hash_frames(fr)
    finally:
        hvd.vpdq.set_dct_mode("strict")
    h2, _ = hvd.vpdq.hash_frames(fr)
    assert np.array_equal(h2, hs) and np.array_equal(qs, qf)
    good = qs >= 31                                     # frames the reference keeps (DedupeDB.py:535-559)
    d = np.unpackbits(hs[good] ^ hf[good], axis=1).sum(1)
    assert d.max() <= 31, "a mode flip alone must never break a frame match at the default tolerance"


def test_k1_empty_batch(gpu, hvd):
    h, q = hvd.vpdq.hash_frames(np.zeros((0, 64, 64), np.uint8))
    assert h.shape == (0, 32) and q.shape == (0,)


def test_k1_golden_rgb512_and_misc(gpu, hvd):
    g = load_golden("pdq_rgb512.npz")
    h, q = hvd.vpdq.hash_frames(g["frames"])
    assert np.array_equal(h, g["hashes"]) and np.array_equal(q, g["quality"])
    m = load_golden("pdq_rgb_misc.npz")
    h, q = hvd.vpdq.hash_frames(m["frames_odd"])
    assert np.array_equal(h, m["hashes_odd"]) and np.array_equal(q, m["quality_odd"])
    h, q = hvd.vpdq.hash_frames(m["frames_64"])
    assert np.array_equal(h, m["hashes_64"]) and np.array_equal(q, m["quality_64"])


@pytest.mark.parametrize("shape", [(5, 512, 512, 3), (3, 64, 200, 3), (2, 257, 64, 3), (2, 130, 190)])
def test_k1_downsampler_vs_oracle(gpu, hvd, oracle, shape):
    n, h, w = shape[:3]
    fr = hvd.synth.frames_rgb(n, seed=21, h=h, w=w) if len(shape) == 4 else hvd.synth.frames_gray(n, 22, h, w)
    hh, q = hvd.vpdq.hash_frames(fr)
    ho, qo = oracle.hash_frames(fr, num_threads=4)
    assert np.array_equal(q, qo) and np.array_equal(hh, ho)


def test_k1_bad_geometry_is_an_error(gpu, hvd):
    with pytest.raises(gpu.HvdError):
        hvd.vpdq.hash_frames(np.zeros((1, 32, 64), np.uint8))
    with pytest.raises(ValueError):
        hvd.vpdq.hash_frames(np.zeros((1, 64, 64, 4), np.uint8))


def test_videohasher_dropin_surface(gpu, hvd, oracle):
    """vpdqpy.py:113-119 call shape: VideoHasher(1, w, h, n).hash_frame(bytes)...finish()."""
    fr = hvd.synth.frames_rgb(6, seed=31, h=512, w=512)
    fr[2] = 40  # a flat frame: quality 0 -> dropped by finish()
    hasher = hvd.VideoHasher(1, 512, 512, 4)
    for f in fr:
        hasher.hash_frame(bytes(f))
    ph = hasher.finish()
    ho, qo = oracle.hash_frames(fr, num_threads=4)
    want = ho[qo >= 31].tobytes()
    assert isinstance(ph, hvd.VpdqHash) and ph.bytes == want
    assert len(ph.bytes) % hvd.VpdqHash.bytesPerPdqHash == 0 and len(ph) == int((qo >= 31).sum()) < 6
    # facade: compute_phash on decoded frames, string round trip
    ph2 = hvd.compute_phash(fr, num_threads=-2)
    assert ph2 == ph and hvd.decode_phash_from_str(hvd.encode_phash_to_str(ph2)) == ph
    # all frames low quality -> empty hash, legal (dedup.py:82)
    e = hvd.compute_phash(np.full((3, 64, 64), 9, np.uint8))
    assert len(e) == 0 and e.bytes == b""
    with pytest.raises(ValueError):
        hasher2 = hvd.VideoHasher(1, 64, 64, 0)
        hasher2.hash_frame(b"\0" * 100)


def test_gray_and_rgb_entries_agree(gpu, hvd):
    g = hvd.synth.frames_gray(16, seed=33)
    hg, qg = hvd.vpdq.hash_frames(g)
    hr, qr = hvd.vpdq.hash_frames(np.repeat(g[..., None], 3, axis=3))
    assert np.array_equal(hg, hr) and np.array_equal(qg, qr)


# ------------------------------------------------------------ K2: all-pairs Hamming --

def test_k2_golden(gpu, hvd):
    g = load_golden("hamming_db.npz")
    got = hvd.allpairs_hamming(g["db"], 31)
    assert np.array_equal(got, g["pairs"])


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 256, 257, 1024, 1025, 4097, 30000])
def test_k2_vs_oracle_sizes(gpu, hvd, oracle, n):
    db, _ = hvd.synth.hash_db(n, seed=40 + n % 7, plant_fraction=0.02)
    got = hvd.allpairs_hamming(db, 31)
    want = oracle.allpairs(db, 31, num_threads=8) if n else got[:0]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("max_dist", [0, 1, 30, 31, 32, 100, 256])
def test_k2_thresholds(gpu, hvd, oracle, max_dist):
    db, _ = hvd.synth.hash_db(600, seed=50, plant_fraction=0.05)
    got = hvd.allpairs_hamming(db, max_dist)
    want = oracle.allpairs(db, max_dist, cap=600 * 600)
    assert np.array_equal(got, want)
    if max_dist == 256:
        assert len(got) == 600 * 599 // 2


def test_k2_duplicates_and_collisions(gpu, hvd, oracle):
    db, _ = hvd.synth.hash_db(2000, seed=51, plant_fraction=0.0)
    db[100:140] = db[7]          # 41 identical hashes -> 820 distance-0 pairs
    db[1999] = ~db[0]            # distance 256
    got = hvd.allpairs_hamming(db, 31)
    assert np.array_equal(got, oracle.allpairs(db, 31))
    assert int((got["dist"] == 0).sum()) == 41 * 40 // 2


def test_k2_group_filter(gpu, hvd, oracle):
    db, _ = hvd.synth.hash_db(5000, seed=52, plant_fraction=0.05)
    grp = (np.arange(5000) // 3).astype(np.int32)
    got = hvd.allpairs_hamming(db, 31, group=grp)
    assert np.array_equal(got, oracle.allpairs(db, 31, group=grp, num_threads=8))


def test_k2_overflow_is_reported_not_truncated(gpu, hvd):
    db = np.zeros((300, 32), np.uint8)
    out = np.zeros(10, gpu.PAIR_DTYPE)
    cnt = C.c_int64(0)
    rc = gpu.load().hvd_allpairs_hamming256(db.ctypes.data, 300, None, 31, out.ctypes.data, 10, C.byref(cnt))
    assert rc == gpu.HVD_ERR_OVERFLOW and cnt.value == 300 * 299 // 2
    assert "too small" in gpu.last_error()
    assert len(hvd.allpairs_hamming(db, 31, cap=10)) == 300 * 299 // 2  # wrapper retries with the exact size


def _run_variant(gpu, hvd, db, variant, max_dist=31, group=None, cap=1 << 16):
    n = len(db)
    lib = gpu.load()
    d_db = gpu.DeviceBuffer.from_array(db)
    d_img = hvd.multigpu.expand_fp4(d_db.ptr, n) if variant >= 8 else None
    d_grp = gpu.DeviceBuffer.from_array(group) if group is not None else None
    d_pairs = gpu.DeviceBuffer(16 * cap)
    d_cnt = gpu.DeviceBuffer(8)
    d_cnt.zero()
    hvd.multigpu.launch_allpairs(lib, d_db.ptr, d_img.ptr if d_img else None, n, d_grp.ptr if d_grp else None,
                                 max_dist, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, variant)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    assert cnt <= cap
    return hvd.multigpu.merge_pairs([d_pairs.to_array(gpu.PAIR_DTYPE, cnt)])


@pytest.mark.parametrize("variant", [0, 1, 8, 9, 12, 13, 18])
def test_k2_all_kernel_variants_agree(gpu, hvd, oracle, variant):
    """Popcount (0, 1) and FP4-MFMA (8, 9, 12, 18, auto 13) forms produce the identical pair list; the forms pruned in round 6
    are refused."""
    n = 20000
    db, _ = hvd.synth.hash_db(n, seed=53, plant_fraction=0.01)
    want = oracle.allpairs(db, 31, num_threads=8)
    assert np.array_equal(_run_variant(gpu, hvd, db, variant), want)
    grp = (np.arange(n) // 5).astype(np.int32)
    assert np.array_equal(_run_variant(gpu, hvd, db, variant, group=grp), oracle.allpairs(db, 31, group=grp, num_threads=8))


def test_k2_pruned_forms_are_refused(gpu, hvd):
    """Round 6 removed the measured-and-lost forms (popcount 2..6, MFMA 10, 11, 14..17, 19): selecting one is an argument
    error at the C-ABI, never a silent substitute."""
    db, _ = hvd.synth.hash_db(3000, seed=54)
    for variant in (2, 3, 4, 5, 6, 7, 10, 11, 14, 15, 16, 17, 19, 20):
        with pytest.raises(gpu.HvdError):
            _run_variant(gpu, hvd, db, variant)
    lib = gpu.load()
    for v in (10, 15, 19):
        assert lib.hvd_debug_set(b"vmatch_variant", v) == gpu.HVD_ERR_ARG


@pytest.mark.parametrize("variant", [8, 9, 12, 13, 18])
@pytest.mark.parametrize("max_dist", [0, 31, 63, 64, 127, 128, 256])
def test_k2_mfma_threshold_routing(gpu, hvd, oracle, variant, max_dist):
    """dot >= 256-2*max_dist is the popcount predicate for every tolerance, including the ones
    where the prefilter (>= 64) or the sign trick (>= 128) must hand over to another kernel."""
    db, _ = hvd.synth.hash_db(700, seed=56, plant_fraction=0.05)
    want = oracle.allpairs(db, max_dist, cap=700 * 700)
    assert np.array_equal(_run_variant(gpu, hvd, db, variant, max_dist=max_dist, cap=700 * 700), want)


def test_fp4_image_layout(gpu, hvd):
    """Every hash bit b becomes the e2m1 nibble 0x2 (+1.0) or 0xA (-1.0) in chunk (bit/32),
    stored at slot chunk ^ ((row>>1)&7); padding rows are zero."""
    n = 300
    db, _ = hvd.synth.hash_db(n, seed=57, plant_fraction=0.0)
    d_db = gpu.DeviceBuffer.from_array(db)
    d_img = hvd.multigpu.expand_fp4(d_db.ptr, n)
    img = d_img.to_array(np.uint8, d_img.nbytes).reshape(-1, 8, 16)
    assert img.shape[0] == 1024 and not img[n:].any()
    bits = np.unpackbits(db, axis=1, bitorder="little").reshape(n, 8, 32)
    for row in (0, 1, 2, 3, 17, 299):
        for chunk in range(8):
            raw = img[row, chunk ^ ((row >> 1) & 7)]
            nib = np.stack([raw & 15, raw >> 4], axis=1).reshape(-1)  # low nibble first
            assert np.array_equal(nib, np.where(bits[row, chunk] == 1, 0xA, 0x2))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_k2_rank_sharding_on_one_gpu(gpu, hvd, oracle, world):
    """Every rank's tile set run on this GPU in turn: the union must be the exact pair list
    and the per-rank lists disjoint (merge_pairs asserts it)."""
    n = 40000
    db, _ = hvd.synth.hash_db(n, seed=54, plant_fraction=0.01)
    d_db = gpu.DeviceBuffer.from_array(db)
    parts = []
    for r in range(world):
        parts.append(hvd.multigpu.sharded_allpairs(d_db.ptr, n, r, world, None))
    got = hvd.multigpu.merge_pairs(parts)
    assert np.array_equal(got, oracle.allpairs(db, 31, num_threads=8))
    assert min(len(p) for p in parts) > 0


def test_k2_rccl_exchange_single_rank(gpu, hvd):
    """world=1 communicator: exercises ncclCommInitRank + both all-gathers on the box."""
    n = 5000
    db, _ = hvd.synth.hash_db(n, seed=55, plant_fraction=0.02)
    ex = hvd.multigpu.RcclExchange(0, 1, hvd.multigpu.RcclExchange.create_unique_id())
    try:
        d_db = gpu.DeviceBuffer.from_array(db)
        d_pairs = gpu.DeviceBuffer(16 * 4096)
        d_cnt = gpu.DeviceBuffer(8)
        d_cnt.zero()
        d_img = hvd.multigpu.expand_fp4(d_db.ptr, n)
        hvd.multigpu.launch_allpairs(gpu.load(), d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, 4096, d_cnt.ptr,
                                     hvd.search.DEFAULT_VARIANT)
        cnt = int(d_cnt.to_array(np.uint64, 1)[0])
        got = hvd.multigpu.merge_pairs([ex.allgather_pairs_dev(d_pairs.ptr, cnt)])
        assert np.array_equal(got, hvd.allpairs_hamming(db, 31))
        assert len(ex.allgather_pairs_dev(d_pairs.ptr, 0)) == 0
        # all-gather of hash shards produced on-device (config 5: hash on every GPU, replicate the DB)
        d_out = gpu.DeviceBuffer(32 * n)
        gpu.check(gpu.load().hvd_comm_allgather_bytes(d_db.ptr, d_out.ptr, 32 * n))
        gpu.check(gpu.load().hvd_dev_sync())
        assert np.array_equal(d_out.to_array(np.uint8, 32 * n).reshape(-1, 32), db)
    finally:
        ex.close()


def test_k2_full_size_1m_properties(gpu, hvd):
    """BASELINE config 3 (1M hashes, ~5e11 comparisons): size-independent properties.
    Planted pairs within tolerance must all be found with their exact distance; every
    reported pair must verify on the host; nothing else is expected from uniform random
    hashes (P(dist<=31) ~ 8e-38 per pair)."""
    n = 1_000_000
    db, planted = hvd.synth.hash_db(n, seed=3)
    got = hvd.allpairs_hamming(db, 31)
    x = np.unpackbits(db[got["i"]] ^ db[got["j"]], axis=1).sum(1)
    assert np.array_equal(x, got["dist"]) and (got["dist"] <= 31).all() and (got["i"] < got["j"]).all()
    found = set(zip(got["i"].tolist(), got["j"].tolist()))
    d_pl = np.unpackbits(db[planted[:, 0]] ^ db[planted[:, 1]], axis=1).sum(1)
    want = {(int(min(s, d)), int(max(s, d))) for (s, d, _), dd in zip(planted, d_pl) if dd <= 31}
    assert want <= found
    assert len(found) == len(got)  # no duplicates
    # chains (copy of a copy) can add a few pairs beyond the directly planted ones, never many
    assert len(found) - len(want) <= len(want) // 10 + 5


# ----------------------------------------------------------- K3: video-level match --

def test_match_two_vs_oracle(gpu, hvd, oracle):
    g = load_golden("video_match.npz")
    off, fb = g["offsets"], g["frames"]
    rng = np.random.default_rng(60)
    for _ in range(40):
        a, b = rng.integers(0, len(off) - 1, 2)
        ba, bb = fb[off[a]:off[a + 1]].tobytes(), fb[off[b]:off[b + 1]].tobytes()
        assert hvd.vpdq.match_counts(ba, bb, 31) == oracle.match_two(ba, bb, 31)


@pytest.mark.parametrize("na,nb", [(1, 1), (1, 300), (257, 3), (700, 733), (717, 717), (800, 800), (5, 3000)])
def test_match_two_small_and_large_operand_paths(gpu, hvd, oracle, na, nb):
    """hvd_match_two compares out of LDS, straight from pinned host memory, while 40 (na + nb) <= 56 KiB (1433 frames)
    and through device buffers beyond; both must give the oracle's counters, at several tolerances."""
    fr, _ = hvd.synth.hash_db(na + nb, seed=na * 7 + nb, plant_fraction=0.3)
    a, b = fr[:na].tobytes(), fr[na:].tobytes()
    for tol in (0, 31, 90):
        assert hvd.vpdq.match_counts(a, b, tol) == oracle.match_two(a, b, tol), (na, nb, tol)
    assert hvd.vpdq.match_counts(a, a, 0) == (na, na) or len(set(map(bytes, fr[:na]))) < na


def test_match_hash_semantics(gpu, hvd):
    g = load_golden("video_match.npz")
    v = hvd.VpdqHash(g["frames"][:12].tobytes())
    e = hvd.VpdqHash(b"")
    assert hvd.matchHash(v, v, 31) == 100.0
    assert hvd.Vpdq.is_similar(v, v) == (True, 100.0)
    assert hvd.matchHash(v, e, 31) == 0.0 and hvd.matchHash(e, v, 31) == 0.0 and hvd.matchHash(e, e, 31) == 0.0
    assert hvd.Vpdq.is_similar(e, e)[0] is False  # an empty hash is not even similar to itself (DedupeDB.py:555-557)
    assert hvd.calculate_distance(v.bytes, v.bytes) == 1 and hvd.calculate_distance(v.bytes, b"") == 101
    w = hvd.VpdqHash(g["frames"][:6].tobytes() + g["frames"][100:106].tobytes())
    s = hvd.get_phash_similarity(v, w)
    assert s == 50.0 and hvd.Vpdq.is_similar(v, w, threshold=50.0) == (True, 50.0)
    assert hvd.Vpdq.is_similar(v, w)[0] is False  # default threshold 75
    # large inputs: more frames than one workgroup pass
    big = hvd.synth.video_hashes(2, seed=61, frames_per_video=700, copy_fraction=0.0)[0]
    a, b = big[:700].tobytes(), big[350:1050].tobytes()
    assert hvd.vpdq.match_counts(a, b, 31) == (350, 350)


def test_k3_golden(gpu, hvd):
    g = load_golden("video_match.npz")
    got = hvd.match_videos(g["frames"], g["offsets"], 31)
    assert np.array_equal(got, g["records"])


@pytest.mark.parametrize("fpv", [1, 64, (0, 40), (50, 200)])
def test_k3_vs_oracle(gpu, hvd, oracle, fpv):
    V = 300 if fpv != (50, 200) else 80
    frames, offsets, planted = hvd.synth.video_hashes(V, seed=62, frames_per_video=fpv, copy_fraction=0.1)
    got = hvd.match_videos(frames, offsets, 31)
    want = oracle.match_videos(frames, offsets, 31)
    assert np.array_equal(got, want)
    if fpv != (0, 40):
        assert len(got) >= len(planted) > 0


def test_find_potential_duplicates_pair_set(gpu, hvd, oracle):
    """dedup.py:445-502 semantics: {A,B}: int(sim) >= int(threshold), from brute force."""
    frames, offsets, planted = hvd.synth.video_hashes(200, seed=63, frames_per_video=(1, 64), copy_fraction=0.15)
    blobs = [frames[offsets[v]:offsets[v + 1]].tobytes() for v in range(200)]
    got = hvd.find_potential_duplicates([hvd.VpdqHash(b) for b in blobs], threshold=50.0)
    want = []
    for a in range(200):
        for b in range(a + 1, 200):
            q, t = oracle.match_two(blobs[a], blobs[b], 31)
            sim = min(q * 100.0 / (len(blobs[a]) // 32), t * 100.0 / (len(blobs[b]) // 32))
            if int(sim) >= 50:
                want.append((a, b))
    assert got == want and len(got) > 0
    # every reported pair is confirmed by the legacy per-pair entry point, both directions
    for a, b in got[:10]:
        assert hvd.calculate_distance(blobs[a], blobs[b]) <= hvd.fix_vpdq_similarity(50.0)
        assert hvd.calculate_distance(blobs[b], blobs[a]) <= hvd.fix_vpdq_similarity(50.0)


def test_end_to_end_hash_then_search(gpu, hvd, oracle):
    """BASELINE config 5 in miniature: videos of 64x64 frames -> hash on GPU -> video search;
    near-copies (per-pixel noise +-2) must be found, and everything equals the oracle."""
    rng = np.random.default_rng(64)
    V, F = 40, 16
    base = hvd.synth.frames_gray(V * F, seed=65, const_fraction=0.0).reshape(V, F, 64, 64)
    copies = {5: 2, 17: 9, 30: 29}
    for dst, src in copies.items():
        noisy = base[src].astype(np.int16) + rng.integers(-2, 3, base[src].shape)
        base[dst] = np.clip(noisy, 0, 255).astype(np.uint8)
    phashes = [hvd.compute_phash(base[v]) for v in range(V)]
    ho, qo = oracle.hash_frames(base.reshape(-1, 64, 64), num_threads=8)
    for v in range(V):
        sel = slice(v * F, (v + 1) * F)
        assert phashes[v].bytes == ho[sel][qo[sel] >= 31].tobytes()
    dup = hvd.find_potential_duplicates(phashes, threshold=50.0)
    want = []
    for a in range(V):
        for b in range(a + 1, V):
            q, t = oracle.match_two(phashes[a].bytes, phashes[b].bytes, 31)
            if len(phashes[a]) and len(phashes[b]) and int(min(q * 100.0 / len(phashes[a]),
                                                               t * 100.0 / len(phashes[b]))) >= 50:
                want.append((a, b))
    assert dup == want
    planted = {(min(s_, d_), max(s_, d_)) for d_, s_ in copies.items()}
    assert len(planted & set(dup)) >= 2  # noisy copies of high-contrast videos stay within tolerance


def test_pipeline_hash_videos_matches_per_video_path(gpu, hvd, oracle):
    rng = np.random.default_rng(66)
    lens = [0, 1, 7, 16, 3]
    frames = hvd.synth.frames_gray(sum(lens), seed=67)
    vids, pos = [], 0
    for n in lens:
        vids.append(frames[pos:pos + n])
        pos += n
    ph = hvd.hash_videos(vids)
    for v, p in zip(vids, ph):
        assert p == (hvd.compute_phash(v) if len(v) else hvd.VpdqHash(b""))
    _, pairs = hvd.dedupe_videos(vids + [vids[3].copy()], threshold=50.0)
    if len(ph[3]):
        assert (3, 5) in pairs


def test_k3_full_size_config5_properties(gpu, hvd):
    """BASELINE config 5 search half at full size: 50k videos x 64 frame hashes = 3.2M frames,
    ~5.1e12 frame comparisons, 1.25e9 video pairs. Size-independent properties: every planted
    near-copy is reported with full counters, every record verifies against the per-pair entry
    point on a sample, nothing is reported between unrelated (uniform random) videos."""
    V, F = 50_000, 64
    frames, offsets, planted = hvd.synth.video_hashes(V, seed=5, frames_per_video=F, copy_fraction=0.02, max_flips=24)
    recs = hvd.match_videos(frames, offsets, 31)
    got = {(int(r["a"]), int(r["b"])): (int(r["q_hits"]), int(r["t_hits"])) for r in recs}
    assert len(planted) >= 900
    full = 0
    for idx, (s_, d_) in enumerate(planted):
        q, t = got[(int(s_), int(d_))]  # KeyError = a planted copy was missed
        if idx % 2 == 0:
            assert (q, t) == (F, F)
            full += 1
        else:
            assert q >= F // 2 and t >= F // 2
    assert full > 400
    # chains (copy of a copy) add a few records; unrelated videos add none
    assert len(planted) <= len(got) <= len(planted) + len(planted) // 5
    for (a, b), (q, t) in list(got.items())[:20]:
        assert hvd.vpdq.match_counts(frames[offsets[a]:offsets[a + 1]].tobytes(),
                                     frames[offsets[b]:offsets[b + 1]].tobytes(), 31) == (q, t)
    pairs = hvd.search.similar_video_pairs(recs, np.diff(offsets), 50.0)
    assert len(pairs) >= len(planted) * 0.95


@pytest.mark.parametrize("geom", [(64, 64, 1, 1000), (64, 64, 3, 300), (512, 512, 3, 150), (130, 190, 1, 40)])
def test_streaming_hasher_ring(gpu, hvd, oracle, geom):
    """f4: the native streaming hasher with a tiny batch so that the ring wraps many times
    (slot reuse, back-pressure, partial last batch); results must come back in push order."""
    w, h, ch, n = geom
    fr = hvd.synth.frames_rgb(n, seed=71, h=h, w=w) if ch == 3 else hvd.synth.frames_gray(n, 72, h, w)
    ho, qo = oracle.hash_frames(fr, num_threads=8)
    for batch_bytes in (fr[0].nbytes * 7, 64 << 20):
        hasher = hvd.VideoHasher(1, w, h, 0, batch_bytes=batch_bytes)
        for k, f in enumerate(fr):
            hasher.hash_frame(bytes(f) if k % 2 else f)  # bytes objects and buffer objects alike
        ph = hasher.finish()
        assert ph.bytes == ho[qo >= 31].tobytes()
    lib = gpu.load()
    hdl = C.c_void_p()
    gpu.check(lib.hvd_hasher_create(w, h, ch, 5, C.byref(hdl)))
    try:
        for rounds in range(2):  # reusable after finish()
            for f in fr[:23]:
                gpu.check(lib.hvd_hasher_push(hdl, f.ctypes.data))
            pend = C.c_int64(0)
            gpu.check(lib.hvd_hasher_pending(hdl, C.byref(pend)))
            assert pend.value == 23
            hh = np.zeros((23, 32), np.uint8)
            qq = np.zeros(23, np.int32)
            got = C.c_int64(0)
            gpu.check(lib.hvd_hasher_finish(hdl, hh.ctypes.data, qq.ctypes.data, 23, C.byref(got)))
            assert got.value == 23 and np.array_equal(hh, ho[:23]) and np.array_equal(qq, qo[:23])
    finally:
        lib.hvd_hasher_destroy(hdl)


def test_k2_full_size_10m_sharded_8_ways(gpu, hvd):
    """BASELINE config 4 (10M hashes, ~5e13 comparisons, 8 ranks): the 8 ranks' tile sets are run
    one after the other on this GPU; the union must contain every planted pair within tolerance
    with its exact distance, ranks must be disjoint and balanced, every record must verify."""
    n, world = 10_000_000, 8
    db, planted = hvd.synth.hash_db(n, seed=4)
    d_db = gpu.DeviceBuffer.from_array(db)
    parts = [hvd.multigpu.sharded_allpairs(d_db.ptr, n, r, world, None) for r in range(world)]
    got = hvd.multigpu.merge_pairs(parts)  # raises if two ranks report the same pair
    x = np.unpackbits(db[got["i"]] ^ db[got["j"]], axis=1).sum(1)
    assert np.array_equal(x, got["dist"]) and (got["dist"] <= 31).all() and (got["i"] < got["j"]).all()
    d_pl = np.unpackbits(db[planted[:, 0]] ^ db[planted[:, 1]], axis=1).sum(1)
    want = {(int(min(s, d)), int(max(s, d))) for (s, d, _), dd in zip(planted, d_pl) if dd <= 31}
    found = set(zip(got["i"].tolist(), got["j"].tolist()))
    assert want <= found and len(found) - len(want) <= len(want) // 10 + 5
    sizes = [len(p) for p in parts]
    assert min(sizes) > 0.5 * max(sizes), sizes  # planted pairs are uniform over the triangle


def test_k1_down512_all_forms_agree(gpu, hvd, oracle):
    """512x512 front-end: the generic 4-launch path, the workgroup-per-frame kernel and the wave-per-frame kernel
    (forced on: by default it only takes batches of >= 704 frames) are interchangeable bit for bit (rgb24 and gray)."""
    lib = gpu.load()
    rgb = hvd.synth.frames_rgb(5, seed=91)
    gray = hvd.synth.frames_gray(5, seed=92, h=512, w=512)
    want_rgb = oracle.hash_frames(rgb, num_threads=8)
    want_gray = oracle.hash_frames(gray, num_threads=8)
    keys = (b"pdq_fused_down512", b"pdq_down512_wave")
    try:
        for cfg in ((0, 0), (1, 0), (1, 2)):
            for k_, v_ in zip(keys, cfg):
                gpu.check(lib.hvd_debug_set(k_, v_))
            for fr, (ho, qo) in ((rgb, want_rgb), (gray, want_gray)):
                h, q = hvd.vpdq.hash_frames(fr)
                assert np.array_equal(h, ho) and np.array_equal(q, qo), (cfg, fr.shape)
        assert lib.hvd_debug_set(b"pdq_down512_systolic", 1) == -1  # dropped variants are unknown keys
    finally:
        for k_, v_ in zip(keys, (1, 1)):
            gpu.check(lib.hvd_debug_set(k_, v_))


@pytest.mark.parametrize("n,grid", [(1, 0), (3, 2), (64, 0), (130, 48), (200, 64)])
@pytest.mark.parametrize("channels", [3, 1])
def test_k1_down512_wave_kernel_vs_oracle(gpu, hvd, oracle, n, grid, channels):
    """k_down512w (one wave per frame, skewed half-wave pipeline): every frame of ragged batches, with fewer waves
    than frames (grid-stride loop, cross-frame prefetch) and on a fresh scratch buffer each time."""
    lib = gpu.load()
    fr = (hvd.synth.frames_rgb(n, seed=300 + n) if channels == 3 else hvd.synth.frames_gray(n, seed=400 + n, h=512, w=512))
    if n >= 64:  # hard content too: noise, saturated blocks, a constant frame
        rng = np.random.default_rng(n)
        fr[1] = rng.integers(0, 256, fr[1].shape, dtype=np.uint8)
        fr[2] = (rng.integers(0, 2, fr[2].shape) * 255).astype(np.uint8)
        fr[3] = 255
        fr[4] = 0
    ho, qo = oracle.hash_frames(fr, num_threads=16)
    try:
        gpu.check(lib.hvd_debug_set(b"pdq_down512_wave", 2))
        gpu.check(lib.hvd_debug_set(b"pdq_down512_wave_grid", grid))
        h, q = hvd.vpdq.hash_frames(fr)
    finally:
        gpu.check(lib.hvd_debug_set(b"pdq_down512_wave", 1))
        gpu.check(lib.hvd_debug_set(b"pdq_down512_wave_grid", 0))
    assert np.array_equal(q, qo), f"{int((q != qo).sum())} quality mismatches"
    assert np.array_equal(h, ho), f"{int((h != ho).any(1).sum())} hash mismatches"


@pytest.mark.parametrize("strip", [32, 64])
@pytest.mark.parametrize("n", [1, 7, 257, 300])
@pytest.mark.parametrize("channels", [3, 1])
def test_k1_down512_workgroup_kernel_both_strip_widths_vs_oracle(gpu, hvd, oracle, n, strip, channels):
    """k_down512<CH, S> (workgroup per frame): S = 32 (two workgroups per CU) and S = 64 (round 5: one per CU, half the strips, the
    low-latency form the dispatcher takes for batches of <= 256 frames), each forced at batch sizes on both sides of that rule --
    more frames than workgroups included (grid-stride loop: the LDS buffers are reused by the next frame) -- with hard content."""
    lib = gpu.load()
    base_n = min(n, 24)
    fr = (hvd.synth.frames_rgb(base_n, seed=500 + n) if channels == 3 else hvd.synth.frames_gray(base_n, seed=600 + n, h=512, w=512))
    if base_n >= 7:
        rng = np.random.default_rng(n + strip)
        fr[1] = rng.integers(0, 256, fr[1].shape, dtype=np.uint8)
        fr[2] = (rng.integers(0, 2, fr[2].shape) * 255).astype(np.uint8)
        fr[3] = 255
        fr[4] = 0
    ho, qo = oracle.hash_frames(fr, num_threads=16)
    idx = np.arange(n) % base_n
    try:
        gpu.check(lib.hvd_debug_set(b"pdq_down512_wave", 0))
        gpu.check(lib.hvd_debug_set(b"pdq_down512_strip", strip))
        h, q = hvd.vpdq.hash_frames(fr[idx])
    finally:
        gpu.check(lib.hvd_debug_set(b"pdq_down512_wave", 1))
        gpu.check(lib.hvd_debug_set(b"pdq_down512_strip", 0))
    assert np.array_equal(q, qo[idx]), f"{int((q != qo[idx]).sum())} quality mismatches"
    assert np.array_equal(h, ho[idx]), f"{int((h != ho[idx]).any(1).sum())} hash mismatches"
    assert lib.hvd_debug_set(b"pdq_down512_strip", 48) == gpu.HVD_ERR_ARG


def test_k1_down512_large_batch_takes_the_wave_kernel(gpu, hvd, oracle):
    """Default dispatch at a batch size that selects k_down512w (>= 704 frames): 768 rgb24 frames = 604 MB."""
    base = hvd.synth.frames_rgb(48, seed=77)
    fr = np.concatenate([base] * 16)
    ho, qo = oracle.hash_frames(base, num_threads=16)
    h, q = hvd.vpdq.hash_frames(fr)
    assert np.array_equal(h, np.concatenate([ho] * 16)) and np.array_equal(q, np.concatenate([qo] * 16))


@pytest.mark.parametrize("seed", range(int(os.environ.get("HVD_SWEEP_SEEDS", "12"))))
def test_randomized_differential_sweep(gpu, hvd, oracle, seed):
    """Seeded random shapes around the padding / tile / super-panel boundaries (127, 128, 1023,
    1024, 1025, 2047 ...), random tolerances, group maps, rank splits and query x target shapes:
    every GPU entry point against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    edges = np.array([2, 31, 32, 33, 127, 128, 129, 255, 257, 1023, 1024, 1025, 2047, 2049, 3000, 4097, 5555])
    n = int(rng.choice(edges)) + int(rng.integers(0, 3))
    md = int(rng.choice([0, 5, 31, 31, 31, 40, 63, 64, 90, 127, 128, 200]))
    db, _ = hvd.synth.hash_db(n, seed=2000 + seed, plant_fraction=0.05, max_flips=int(min(2 * md + 8, 200)))
    grp = None
    if rng.random() < 0.5:
        grp = rng.integers(0, max(2, n // int(rng.integers(1, 9))), n).astype(np.int32)
    want = oracle.allpairs(db, md, group=grp, cap=max(n * n // 2, 16), num_threads=8)
    # host entry (default kernel)
    assert np.array_equal(hvd.allpairs_hamming(db, md, group=grp), want)
    # every device variant
    for variant in (0, 1, 8, 9, 12, 13, 18):
        assert np.array_equal(_run_variant(gpu, hvd, db, variant, max_dist=md, group=grp, cap=max(len(want), 16)), want), variant
    # rank split of the default kernel
    world = int(rng.integers(2, 6))
    d_db = gpu.DeviceBuffer.from_array(db)
    d_grp = gpu.DeviceBuffer.from_array(grp) if grp is not None else None
    parts = [hvd.multigpu.sharded_allpairs(d_db.ptr, n, r, world, None, max_dist=md,
                                           d_group_ptr=d_grp.ptr if d_grp else None) for r in range(world)]
    assert np.array_equal(hvd.multigpu.merge_pairs(parts), want)
    # video level, ragged, and the query x target form against per-pair matching
    V = int(rng.integers(2, 120))
    frames, offsets, _ = hvd.synth.video_hashes(V, seed=3000 + seed, frames_per_video=(0, int(rng.integers(1, 40))),
                                                copy_fraction=0.3)
    want_v = oracle.match_videos(frames, offsets, 31)
    assert np.array_equal(hvd.match_videos(frames, offsets, 31), want_v)
    dl = hvd.pipeline.DeviceLibrary.from_host(frames, offsets)  # device-resident form, tiny record buffer (re-emit)
    try:
        assert np.array_equal(dl.match_videos(31, cap=int(rng.integers(1, 50))), want_v)
    finally:
        dl.free()
    q_sel = rng.choice(V, size=int(rng.integers(1, V + 1)), replace=False)
    blobs = [frames[offsets[p]:offsets[p + 1]] for p in q_sel]
    fq = np.concatenate(blobs) if sum(len(b) for b in blobs) else np.zeros((0, 32), np.uint8)
    oq = np.zeros(len(blobs) + 1, np.int64)
    np.cumsum([len(b) for b in blobs], out=oq[1:])
    got = hvd.search.match_videos_cross(fq, oq, frames, offsets, q_sel.astype(np.int32), np.arange(V, dtype=np.int32))
    want_x = []
    for a, p in enumerate(q_sel):
        for b in range(V):
            if b == p:
                continue
            qh, th = oracle.match_two(frames[offsets[p]:offsets[p + 1]].tobytes(),
                                      frames[offsets[b]:offsets[b + 1]].tobytes(), 31)
            if qh or th:
                want_x.append((a, b, qh, th))
    assert np.array_equal(got, np.array(want_x, dtype=gpu.VMATCH_DTYPE))
    # frames: a random geometry through both entries
    h, w = int(rng.choice([64, 65, 96, 128, 200, 512])), int(rng.choice([64, 70, 128, 333, 512]))
    nfr = int(rng.integers(1, 9))
    fr = hvd.synth.frames_rgb(nfr, seed=4000 + seed, h=h, w=w) if rng.random() < 0.5 else \
        hvd.synth.frames_gray(nfr, 4000 + seed, h, w)
    hh, qq = hvd.vpdq.hash_frames(fr)
    ho, qo = oracle.hash_frames(fr, num_threads=4)
    assert np.array_equal(hh, ho) and np.array_equal(qq, qo), (h, w, fr.shape)


def test_entry_points_from_a_worker_thread(gpu, hvd, oracle):
    """The reference drives this path from a QThread worker (gui/gui.py:195-237): hashing, matching
    and searching must work from a thread other than the one that called hvd_init."""
    import threading

    fr = hvd.synth.frames_gray(64, seed=93)
    db, _ = hvd.synth.hash_db(3000, seed=94, plant_fraction=0.02)
    out = {}

    def work():
        try:
            out["h"] = hvd.vpdq.hash_frames(fr)
            out["p"] = hvd.allpairs_hamming(db, 31)
            hs = hvd.VideoHasher(1, 64, 64, 0)
            for f in fr:
                hs.hash_frame(f)
            out["v"] = hs.finish()
            out["m"] = hvd.matchHashBytes(db[:10].tobytes(), db[:10].tobytes(), 31)
        except Exception as exc:  # surfaced below
            out["err"] = exc

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert "err" not in out, out.get("err")
    ho, qo = oracle.hash_frames(fr)
    assert np.array_equal(out["h"][0], ho) and np.array_equal(out["h"][1], qo)
    assert np.array_equal(out["p"], oracle.allpairs(db, 31))
    assert out["v"].bytes == ho[qo >= 31].tobytes() and out["m"] == 100.0

"""Round 5 GPU tests: FULL-SIZE differential parity at the sizes BASELINE.json names.

Until round 4 the full-size runs were checked through properties only (every reported pair verifies, planted recall,
sub-library vs oracle, reproducible checksum). A true pair lost at one tile position at full size would have passed.
Here the complete result of every BASELINE search config is compared record for record with an independent
implementation:

  configs[2]  1M hashes        every kernel form == the CPU oracle's pair list (AVX-512 scan, a few seconds)
  configs[3]  10M hashes       union of the 8 rank tile sets (auto MFMA form) == integer popcount kernel (variant 1)
  configs[4]  50k x 64 frames  hvd_vmatch records through form 18 == form 8 (256-bit MFMA, no prefilter, no queue)
                               == a HOST fold of the frame-pair list the popcount kernel produces with the group filter

Semantics: dedup.py:445-502 (pair set), db/vptree.py:29-31 (distance), vpdqpy/vpdqpy.py:49-56 (video counters).
"""
import ctypes as C

import numpy as np
import pytest

from bench import host_threads

pytestmark = pytest.mark.gpu


def _pairs_equal(got, want, what):
    assert len(got) == len(want), f"{what}: {len(got)} records, expected {len(want)}"
    for f in ("i", "j", "dist"):
        assert np.array_equal(got[f], want[f]), f"{what}: field {f} differs"


def test_cfg3_full_pair_list_equals_the_oracle_in_every_form(gpu, hvd, oracle):
    """BASELINE configs[2]: the whole sorted pair list of the 1M-hash DB, through the product entry point (auto form) and
    through every explicit kernel form, equals the oracle's brute force over all 4.999995e11 pairs."""
    n = 1_000_000
    db, _ = hvd.synth.hash_db(n, seed=3)
    want = oracle.allpairs(db, 31, num_threads=host_threads())
    assert 500 < len(want) < 2000  # ~0.1 % planted x 32/41 within tolerance (+ a few chains)
    _pairs_equal(hvd.allpairs_hamming(db, 31), want, "hvd_allpairs_hamming256")
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        for v in (0, 1, 8, 9, 12, 13, 15, 18):
            got = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=v)
            _pairs_equal(got, want, f"variant {v}")
    finally:
        d_db.free()


def test_cfg4_union_of_8_rank_tile_sets_equals_the_popcount_kernel(gpu, hvd):
    """BASELINE configs[3]: 10M hashes. The 8 ranks' tile sets of the auto FP4-MFMA form, run one after the other on this
    GPU and merged, equal -- record for record -- the list of the integer popcount kernel (csrc/k_hamming.hip variant 1:
    xor + v_bcnt, no matrix cores, no FP4 image, different tiling) over the whole triangle in one launch."""
    n, world = 10_000_000, 8
    db, _ = hvd.synth.hash_db(n, seed=4)
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        parts = [hvd.multigpu.sharded_allpairs(d_db.ptr, n, r, world, None) for r in range(world)]
        got = hvd.multigpu.merge_pairs(parts)
        want = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=1)
    finally:
        d_db.free()
    assert 5000 < len(want) < 20000
    _pairs_equal(got, want, "union of 8 rank tile sets")
    # and the independent list itself verifies on the host
    x = np.unpackbits(db[want["i"]] ^ db[want["j"]], axis=1).sum(1)
    assert np.array_equal(x, want["dist"]) and (want["i"] < want["j"]).all()


def fold_frame_pairs(pairs, video, n_videos):
    """Host statement of the video-level reduction (vpdqpy/vpdqpy.py:49-56 for every video pair): from frame pairs
    (i < j, frames of different videos) to one record per video pair a < b with q_hits = distinct frames of a that have
    a match in b, t_hits = distinct frames of b that have a match in a."""
    a = video[pairs["i"]].astype(np.int64)
    b = video[pairs["j"]].astype(np.int64)
    assert (a < b).all()  # frames are in video order and pairs inside one video were filtered
    key = a * n_videos + b
    n_fr = np.int64(video.size)
    q = np.unique(key * n_fr + pairs["i"].astype(np.int64)) // n_fr
    t = np.unique(key * n_fr + pairs["j"].astype(np.int64)) // n_fr
    kq, cq = np.unique(q, return_counts=True)
    kt, ct = np.unique(t, return_counts=True)
    assert np.array_equal(kq, kt)
    out = np.zeros(kq.size, dtype=hvd_vmatch_dtype())
    out["a"], out["b"], out["q_hits"], out["t_hits"] = kq // n_videos, kq % n_videos, cq, ct
    return out


def hvd_vmatch_dtype():
    from hvd_amd import _lib

    return _lib.VMATCH_DTYPE


def test_cfg5_full_library_records_equal_form8_and_the_host_fold(gpu, hvd):
    """BASELINE configs[4] search half at full size: 50 000 videos x 64 synthetic frames generated and hashed in HBM.
    The hvd_vmatch records of the product path (auto -> panel-mark queue, form 18) equal those of form 18 forced, of form
    8 (full 256-bit MFMA compare: no 128-bit first stage, no survivor queue) and a host fold of the frame-pair list the
    integer popcount kernel reports under the video group filter."""
    lib = gpu.load()
    V, F = 50_000, 64
    rng = np.random.default_rng(5)
    copy_of = np.full(V, -1, dtype=np.int32)
    m = int(round(V * 0.02))
    dst = rng.choice(np.arange(1, V), size=m, replace=False)
    is_dst = np.zeros(V, dtype=bool)
    is_dst[dst] = True
    copy_of[dst] = rng.choice(np.flatnonzero(~is_dst), size=m)
    d_copy = gpu.DeviceBuffer.from_array(copy_of)
    d_frames = gpu.DeviceBuffer(V * F * 4096)
    gpu.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, d_copy.ptr))
    raw_off = np.arange(V + 1, dtype=np.int64) * F
    _, recs_auto, library = hvd.pipeline.dedupe_frames_on_device(d_frames.ptr, raw_off, 64, 64, 1, keep_library=True)
    d_frames.free()
    d_copy.free()
    try:
        form = C.c_int(0)
        gpu.check(lib.hvd_debug_get(b"mfma_auto_form", C.byref(form)))
        assert form.value in (9, 12, 18)
        by_form = {}
        try:
            for v in (18, 8):
                gpu.check(lib.hvd_debug_set(b"vmatch_variant", v))
                by_form[v] = library.match_videos()
        finally:
            gpu.check(lib.hvd_debug_set(b"vmatch_variant", 0))
        assert len(recs_auto) >= 900
        for v, r in by_form.items():
            assert np.array_equal(r, recs_auto), f"form {v} records differ from the product path's (form {form.value})"
        # the independent path: integer popcount kernel with the group filter -> frame pairs -> host fold
        fp = hvd.multigpu.sharded_allpairs(library.d_hashes.ptr, library.n_frames, 0, 1, None, 31,
                                           d_group_ptr=library.d_video.ptr, variant=1, cap=1 << 22)
        video = library.d_video.to_array(np.int32, library.n_frames)
        assert (video[fp["i"]] != video[fp["j"]]).all()
        want = fold_frame_pairs(fp, video, V)
        assert np.array_equal(recs_auto, want), "video records differ from the host fold of the popcount kernel's frame pairs"
    finally:
        library.free()

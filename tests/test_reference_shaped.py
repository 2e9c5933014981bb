"""Mirror of the reference's own hot-path tests (tests/unit_tests/test_vpdqpy.py), on synthetic
clips because its fixture submodule (tests/testdb: Big Buck Bunny / Sintel clips, known-good hash
texts) is not available offline:

    test_hashing                   :99-101   every clip, odd ones included, hashes to len > 0
    test_hashing_identical         :105-128  recomputed hash == stored known-good text, else
                                             100 - similarity < 1.0
    test_compare_similarity_true   :131-145  is_similar (threshold 75) <=> same SXX_ group

Clips are named like the reference's ("S01_a", "S01_b", ... share a group); members of a group are
re-encodes of one source (per-pixel noise, brightness shift, a few dropped frames)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def source_clip(rng, frames, h=128, w=128):
    """A high-contrast scene that drifts slowly over time (what a real clip looks like to PDQ)."""
    yy, xx = np.meshgrid(np.arange(h) / h, np.arange(w) / w, indexing="ij")
    comps = [(rng.uniform(0.5, 4), rng.uniform(0.5, 4), rng.uniform(0, 6.28), rng.uniform(15, 45), rng.uniform(-0.05, 0.05))
             for _ in range(8)]
    out = np.empty((frames, h, w, 3), np.uint8)
    for t in range(frames):
        img = np.full((h, w), 128.0)
        for fx, fy, ph, amp, drift in comps:
            img += amp * np.cos(2 * np.pi * (fx * xx + fy * yy) + ph + drift * t * 6.28)
        g = np.clip(img, 0, 255)
        out[t, ..., 0] = g
        out[t, ..., 1] = np.clip(g * 0.9 + 10, 0, 255)
        out[t, ..., 2] = np.clip(255 - g * 0.8, 0, 255)
    return out


def make_clips(hvd, n_groups=6, frames=24, seed=500):
    rng = np.random.default_rng(seed)
    clips = {}
    for g in range(n_groups):
        src = source_clip(rng, frames)
        clips[f"S{g:02d}_original"] = src
        noisy = np.clip(src.astype(np.int16) + rng.integers(-3, 4, src.shape), 0, 255).astype(np.uint8)
        clips[f"S{g:02d}_reencode"] = noisy
        bright = np.clip(src.astype(np.int16) + 6, 0, 255).astype(np.uint8)
        clips[f"S{g:02d}_brighter"] = np.delete(bright, [3, 11], axis=0)  # two dropped frames
    # "strange" clips: hash but are not compared (reference test_vpdqpy.py:48-53)
    strange = {
        "tiny_64x64": hvd.synth.frames_rgb(5, seed=seed + 900, h=64, w=64),
        "single_frame": hvd.synth.frames_rgb(1, seed=seed + 901, h=128, w=128),
        "wide_64x300": hvd.synth.frames_rgb(4, seed=seed + 902, h=64, w=300),
    }
    return clips, strange


def similar_group(a: str, b: str) -> bool:
    """reference similar_group (:79-86): same SXX prefix."""
    if a.split("_")[0][0] != "S" or b.split("_")[0][0] != "S":
        return False
    return a.split("_")[0] == b.split("_")[0]


@pytest.fixture(scope="module")
def hashed(gpu, hvd):
    clips, strange = make_clips(hvd)
    calc = lambda d: {name: hvd.Vpdq.computeHash(fr) for name, fr in d.items()}  # noqa: E731
    return calc(clips), calc(strange), clips


def test_hashing(hashed):
    hashes, strange_hashes, _ = hashed
    for name, ph in {**hashes, **strange_hashes}.items():
        assert len(ph) > 0, name


def test_hashing_identical(hvd, hashed, tmp_path, oracle):
    hashes, _, clips = hashed
    for name, ph in hashes.items():
        # "known good" text written by a known-good implementation: here the CPU oracle
        ho, qo = oracle.hash_frames(clips[name], num_threads=4)
        (tmp_path / f"{name}.txt").write_text(str(hvd.VpdqHash(ho[qo >= 31].tobytes())))
        expected = hvd.VpdqHash.from_string((tmp_path / f"{name}.txt").read_text())
        similar, similarity = hvd.Vpdq.is_similar(ph, expected)
        assert 0.0 <= similarity <= 100.0
        if expected != ph:  # the reference tolerates environmental drift below 1.0; here there is none
            assert 1.0 > (100.0 - similarity), name
        assert expected == ph and similar and similarity == 100.0


def test_compare_similarity_true(hvd, hashed):
    hashes, _, _ = hashed
    names = list(hashes)
    for a in names:
        for b in names:
            if a == b:
                continue
            similar, similarity = hvd.Vpdq.is_similar(hashes[a], hashes[b])  # threshold 75
            assert 0.0 <= similarity <= 100.0
            assert similar == similar_group(a, b), (a, b, similarity)


def test_benchmark_shaped_all_pairs_loop_equals_batch_search(hvd, hashed):
    """tests/benchmarks/test_benchmark_vpdqpy.py:62-73 runs is_similar over all j >= i pairs; the batch
    search must give the same verdicts in one pass."""
    hashes, _, _ = hashed
    names = list(hashes)
    loop = {(i, j) for i in range(len(names)) for j in range(i + 1, len(names))
            if hvd.Vpdq.is_similar(hashes[names[i]], hashes[names[j]], threshold=75)[0]}
    batch = set(hvd.find_potential_duplicates([hashes[n] for n in names], threshold=75.0))
    assert batch == loop and len(loop) == 6 * 3

"""CPU tests of the oracle itself: the C restatement against the independent numpy
restatement, against the committed golden fixtures, and against analytic known answers.
(The reference's own vectors are unavailable -- PARITY UNPINNED, oracle/hvd_oracle.c.)"""
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle import pdq_numpy as P


def test_dct_matrix_two_implementations(oracle):
    assert np.array_equal(oracle.dct_matrix().view(np.uint32), P.dct_matrix().view(np.uint32))


def test_dct_matrix_is_orthonormal_rows(oracle):
    d = oracle.dct_matrix().astype(np.float64)
    assert np.allclose(d @ d.T, np.eye(16), atol=1e-6)


def test_golden_gray64(oracle):
    g = load_golden("pdq_gray64.npz")
    h, q, c = oracle.hash_frames(g["frames"], want_coeffs=True)
    assert np.array_equal(h, g["hashes"])
    assert np.array_equal(q, g["quality"])
    assert np.array_equal(c.view(np.uint32), g["coeffs"].view(np.uint32))


def test_golden_gray64_numpy_restatement():
    g = load_golden("pdq_gray64.npz")
    for f in range(0, len(g["frames"]), 3):
        h, q, b = P.hash_gray(g["frames"][f])
        assert h == g["hashes"][f].tobytes()
        assert q == g["quality"][f]
        assert np.array_equal(b.ravel().view(np.uint32), g["coeffs"][f].view(np.uint32))


def test_golden_rgb512(oracle):
    g = load_golden("pdq_rgb512.npz")
    h, q, c = oracle.hash_frames(g["frames"], want_coeffs=True)
    assert np.array_equal(h, g["hashes"]) and np.array_equal(q, g["quality"])
    assert np.array_equal(c.view(np.uint32), g["coeffs"].view(np.uint32))
    hh, qq, _ = P.hash_rgb(g["frames"][0])  # Jarosz path of the second implementation
    assert hh == g["hashes"][0].tobytes() and qq == g["quality"][0]


def test_golden_fma_mode_both_restatements(oracle):
    """The opt-in fused-multiply-add DCT mode: C oracle (libm fmaf) and numpy (TwoSum, no libm) agree
    with the frozen vectors; the quality and the default mode are untouched by it."""
    g = load_golden("pdq_gray64.npz")
    h, q, c = oracle.hash_frames(g["frames"], want_coeffs=True, fma=True)
    assert np.array_equal(h, g["hashes_fma"]) and np.array_equal(q, g["quality"])
    assert np.array_equal(c.view(np.uint32), g["coeffs_fma"].view(np.uint32))
    assert not np.array_equal(g["coeffs_fma"].view(np.uint32), g["coeffs"].view(np.uint32))
    for f in range(1, len(g["frames"]), 5):
        hh, qq, b = P.hash_gray(g["frames"][f], fma=True)
        assert hh == g["hashes_fma"][f].tobytes() and qq == g["quality"][f]
        assert np.array_equal(b.ravel().view(np.uint32), g["coeffs_fma"][f].view(np.uint32))
    h0, _ = oracle.hash_frames(g["frames"])  # the mode does not leak into later calls
    assert np.array_equal(h0, g["hashes"])
    r = load_golden("pdq_rgb512.npz")
    h, q = oracle.hash_frames(r["frames"], fma=True)
    assert np.array_equal(h, r["hashes_fma"]) and np.array_equal(q, r["quality"])


def test_fma32_helper_is_a_single_rounding():
    """_fma32 against exact rational arithmetic, including products that land on float32 ties."""
    from fractions import Fraction
    rng = np.random.default_rng(5)
    a = rng.standard_normal(4000).astype(np.float32)
    b = rng.standard_normal(4000).astype(np.float32)
    c = (rng.standard_normal(4000) * 10.0 ** rng.integers(-6, 3, 4000)).astype(np.float32)
    # engineered ties: a*b = 1 + 2^-24 exactly half an ulp above 1, then c decides the direction
    a[:4] = np.float32(1 + 2.0 ** -12); b[:4] = np.float32(1 - 2.0 ** -12 + 2.0 ** -24)
    c[:4] = np.float32([0.0, 2.0 ** -60, -2.0 ** -60, 2.0 ** -30])
    got = P._fma32(a, b, c)
    for i in range(len(a)):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        r = np.float32(float(got[i]))
        lo, hi = np.nextafter(r, np.float32(-np.inf)), np.nextafter(r, np.float32(np.inf))
        err = abs(Fraction(float(r)) - exact)
        assert err <= abs(Fraction(float(lo)) - exact) and err <= abs(Fraction(float(hi)) - exact), i
        if err == abs(Fraction(float(lo)) - exact) or err == abs(Fraction(float(hi)) - exact):
            assert (int(r.view(np.uint32)) & 1) == 0, f"tie not broken to even at {i}"


@pytest.mark.parametrize("fma,n", [(False, 10000), (True, 3000)])
def test_two_restatements_agree_on_many_frames(oracle, fma, n):
    """C oracle vs the independent numpy restatement, bit for bit (hash, quality and all
    256 coefficients) on 10^4 seeded frames of the bench generator."""
    import hvd_amd
    fr = hvd_amd.synth.frames_gray(n, seed=2)
    h, q, c = oracle.hash_frames(fr, num_threads=8, want_coeffs=True, fma=fma)
    for lo in range(0, n, 2500):
        hp, qp, cp = P.hash_gray64_batch(fr[lo:lo + 2500], fma=fma)
        sl = slice(lo, lo + 2500)
        assert np.array_equal(cp.view(np.uint32), c[sl].view(np.uint32))
        assert np.array_equal(qp, q[sl]) and np.array_equal(hp, h[sl])


def test_mirrored_frame_flips_the_sign_of_even_indexed_frequencies(oracle):
    """Analytic property of the DCT rows: D[i][63-j] = (-1)^(i+1) ... with frequency index
    i+1, so mirroring a frame left-right negates coefficient columns 0,2,4,.. and keeps 1,3,5,..; mirroring
    top-bottom does the same to coefficient rows. Float summation order differs, hence a tolerance."""
    import hvd_amd
    fr = hvd_amd.synth.frames_gray(16, seed=77)
    _, _, c = oracle.hash_frames(fr, want_coeffs=True)
    _, _, ch = oracle.hash_frames(fr[:, :, ::-1], want_coeffs=True)
    _, _, cv = oracle.hash_frames(fr[:, ::-1, :], want_coeffs=True)
    c, ch, cv = (x.reshape(-1, 16, 16).astype(np.float64) for x in (c, ch, cv))
    sgn = np.where(np.arange(16) % 2 == 0, -1.0, 1.0)
    assert np.allclose(ch, c * sgn[None, None, :], atol=2e-2)
    assert np.allclose(cv, c * sgn[None, :, None], atol=2e-2)
    assert np.abs(c).max() > 50  # the tolerance is tiny against the signal


def test_golden_rgb_misc(oracle):
    g = load_golden("pdq_rgb_misc.npz")
    h, q = oracle.hash_frames(g["frames_odd"])
    assert np.array_equal(h, g["hashes_odd"]) and np.array_equal(q, g["quality_odd"])
    h, q = oracle.hash_frames(g["frames_64"])
    assert np.array_equal(h, g["hashes_64"]) and np.array_equal(q, g["quality_64"])


def test_threads_do_not_change_results(oracle):
    g = load_golden("pdq_gray64.npz")
    h1, q1 = oracle.hash_frames(g["frames"], num_threads=1)
    h4, q4 = oracle.hash_frames(g["frames"], num_threads=4)
    assert np.array_equal(h1, h4) and np.array_equal(q1, q4)


def test_gray_entry_equals_rgb_entry_with_equal_channels(oracle):
    g = load_golden("pdq_gray64.npz")["frames"][:8]
    rgb = np.repeat(g[..., None], 3, axis=3)
    hg, qg = oracle.hash_frames(g)
    hr, qr = oracle.hash_frames(rgb)
    assert np.array_equal(hg, hr) and np.array_equal(qg, qr)


def test_constant_frame_has_quality_zero(oracle):
    fr = np.stack([np.full((64, 64), v, np.uint8) for v in (0, 1, 77, 255)])
    _, q = oracle.hash_frames(fr)
    assert q.tolist() == [0, 0, 0, 0]


def test_hash_has_128_bits_set_without_ties(oracle):
    g = load_golden("pdq_gray64.npz")
    c = g["coeffs"]
    for f in range(len(c)):
        if len(np.unique(c[f])) == 256:  # no ties => exactly the 128 largest are set
            assert int(np.unpackbits(g["hashes"][f]).sum()) == 128


def test_single_basis_image_sets_its_coefficient(oracle):
    # frame = 128 + 100 * outer(D[p], D[q]) scaled: coefficient (p,q) must be the largest one
    d = oracle.dct_matrix().astype(np.float64)
    for p, q in [(0, 0), (3, 7), (15, 15), (9, 2)]:
        img = 128 + 800 * np.outer(d[p], d[q])
        fr = np.clip(np.rint(img), 0, 255).astype(np.uint8)[None]
        _, _, c = oracle.hash_frames(fr, want_coeffs=True)
        assert int(np.argmax(np.abs(c[0]))) == p * 16 + q
        k = p * 16 + q
        bit = (oracle.hash_frames(fr)[0][0][k >> 3] >> (k & 7)) & 1
        assert bit == (1 if c[0][k] > 0 else 0)


def test_bit_layout_byte_k_div_8_bit_k_mod_8(oracle):
    g = load_golden("pdq_gray64.npz")
    c, h = g["coeffs"][0], g["hashes"][0]
    med = np.sort(c)[127]
    for k in range(256):
        assert ((h[k >> 3] >> (k & 7)) & 1) == (1 if c[k] > med else 0)


def test_hamming_known_answers(oracle):
    rng = np.random.default_rng(7)
    x = rng.integers(0, 256, 32, dtype=np.uint8)
    assert oracle.hamming256(x, x) == 0
    assert oracle.hamming256(x, ~x) == 256
    for k in range(256):  # every bit position, all byte/word boundaries
        y = x.copy()
        y[k >> 3] ^= 1 << (k & 7)
        assert oracle.hamming256(x, y) == 1
    for nflip in (31, 32):
        y = x.copy()
        for k in range(0, 8 * nflip, 8):
            y[k >> 3] ^= 1
        assert oracle.hamming256(x, y) == nflip
        assert oracle.hamming256(x, y) == P.hamming(x.tobytes(), y.tobytes())


def test_golden_allpairs(oracle):
    g = load_golden("hamming_db.npz")
    got = oracle.allpairs(g["db"], 31)
    assert np.array_equal(got, g["pairs"])
    # multi-threaded oracle and row-range form give the same list
    assert np.array_equal(oracle.allpairs(g["db"], 31, num_threads=4), g["pairs"])
    a = oracle.allpairs(g["db"], 31, rows=(0, 1500))
    b = oracle.allpairs(g["db"], 31, rows=(1500, 3000))
    assert np.array_equal(np.concatenate([a, b]), g["pairs"])


def test_allpairs_boundary_31_vs_32(oracle):
    g = load_golden("hamming_db.npz")
    p31 = oracle.allpairs(g["db"], 31)
    p32 = oracle.allpairs(g["db"], 32)
    assert set(map(tuple, p31[["i", "j"]].tolist())) <= set(map(tuple, p32[["i", "j"]].tolist()))
    assert (p31["dist"] <= 31).all() and (p32["dist"] <= 32).all()
    extra = len(p32) - len(p31)
    assert extra == int((p32["dist"] == 32).sum())


def test_allpairs_group_filter(oracle):
    g = load_golden("hamming_db.npz")
    grp = (np.arange(len(g["db"])) // 7).astype(np.int32)
    got = oracle.allpairs(g["db"], 31, group=grp)
    want = g["pairs"][grp[g["pairs"]["i"]] != grp[g["pairs"]["j"]]]
    assert np.array_equal(got, want)


def test_allpairs_bands_equal_the_full_scan_restricted_to_their_rows(oracle):
    """hvd_cpu_allpairs_hamming256_bands (the checker of the full-size configs[3] test): the rows of several disjoint
    bands in one call == the golden full list restricted to those rows, with and without the group filter, for any thread
    count, through the overflow retry; malformed band lists are refused."""
    g = load_golden("hamming_db.npz")
    db, full = g["db"], g["pairs"]
    n = len(db)
    bands = [(0, 17), (100, 101), (101, 1000), (n - 300, n)]
    inb = np.zeros(n, dtype=bool)
    for a, b in bands:
        inb[a:b] = True
    want = full[inb[full["i"]]]
    assert 0 < len(want) < len(full)
    for threads in (1, 3):
        assert np.array_equal(oracle.allpairs_bands(db, bands, 31, num_threads=threads), want)
    assert np.array_equal(oracle.allpairs_bands(db, bands, 31, cap=1), want)  # overflow -> retried with the true size
    grp = (np.arange(n) // 7).astype(np.int32)
    wg = want[grp[want["i"]] != grp[want["j"]]]
    assert np.array_equal(oracle.allpairs_bands(db, bands, 31, group=grp, num_threads=2), wg)
    assert len(oracle.allpairs_bands(db, [], 31)) == 0
    assert np.array_equal(oracle.allpairs_bands(db, [(0, n)], 31, num_threads=2), full)
    for bad in ([(5, 3)], [(0, n + 1)], [(10, 20), (15, 30)]):
        with pytest.raises(RuntimeError):
            oracle.allpairs_bands(db, bad, 31)


def test_fold_of_group_filtered_frame_pairs_is_the_video_match(oracle):
    """The full-size configs[4] gate (bench.py / tests/test_gpu_round5.py) derives the expected hvd_vmatch records from the
    oracle's frame-pair scan with the video group filter, folded on the host. That derivation must itself equal the oracle's
    direct per-video-pair statement (hvd_cpu_vpdq_match_videos: vpdqpy/vpdqpy.py:49-56 for every pair of videos)."""
    from bench import fold_frame_pairs
    from hvd_amd import synth

    fr, off, _ = synth.video_hashes(300, seed=12, frames_per_video=(0, 30), copy_fraction=0.3)
    V = off.size - 1
    video = np.repeat(np.arange(V, dtype=np.int32), np.diff(off))
    fp = oracle.allpairs(fr, 31, group=video, num_threads=2)
    want = oracle.match_videos(fr, off, 31)
    assert len(want) > 30
    assert np.array_equal(fold_frame_pairs(fp, video, V, oracle.VMATCH_DTYPE), want)


def test_golden_video_match(oracle):
    g = load_golden("video_match.npz")
    got = oracle.match_videos(g["frames"], g["offsets"], 31)
    assert np.array_equal(got, g["records"])
    # each record equals the pairwise matcher, and absent pairs have no hits
    off = g["offsets"]
    fb = g["frames"]
    have = {(int(r["a"]), int(r["b"])): (int(r["q_hits"]), int(r["t_hits"])) for r in got}
    for a in range(0, len(off) - 1, 5):
        for b in range(a + 1, len(off) - 1, 3):
            q, t = oracle.match_two(fb[off[a]:off[a + 1]].tobytes(), fb[off[b]:off[b + 1]].tobytes(), 31)
            assert have.get((a, b), (0, 0)) == (q, t)


def test_match_two_empty_and_self(oracle):
    g = load_golden("video_match.npz")
    v = g["frames"][:10].tobytes()
    assert oracle.match_two(b"", v) == (0, 0)
    assert oracle.match_two(v, b"") == (0, 0)
    assert oracle.match_two(v, v) == (10, 10)


def test_division_free_quality_term(tmp_path):
    """k_pdq_hash64 replaces (int)(x / 255.0f) by a multiply + exact-remainder correction;
    tests/tools/check_div255.c compares the two for floats |x| <= 26000 (every 61st float here,
    all 2.4e9 of them with HVD_EXHAUSTIVE=1; the exhaustive run was done when the kernel was
    written: 0 mismatches)."""
    import os
    import subprocess

    from conftest import ROOT

    exe = tmp_path / "check_div255"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", str(exe),
                           os.path.join(ROOT, "tests", "tools", "check_div255.c"), "-lm"])
    stride = "1" if os.environ.get("HVD_EXHAUSTIVE") == "1" else "61"
    out = subprocess.run([str(exe), stride], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert " 0 mismatches" in out.stdout


def test_quality_term_gray_shortcut_is_exact():
    """k_pdq_hash64 computes the quality term of GRAY BYTE frames as trunc(|u - v| * RN(100/255)) with u, v the lumas
    of two bytes. Exhaustively over all 256 x 256 byte pairs this equals the reference's |(int)(((u - v) * 100) / 255)|
    (pdqhashing.cpp's gradient term as the oracle restates it), all in binary32."""
    f32 = np.float32
    g = np.arange(256, dtype=f32)
    y = (f32(0.299) * g).astype(f32)
    y = (y + (f32(0.587) * g).astype(f32)).astype(f32)
    y = (y + (f32(0.114) * g).astype(f32)).astype(f32)
    u, v = np.meshgrid(y, y, indexing="ij")
    d = (u - v).astype(f32)
    ref = np.abs(np.trunc((((d * f32(100)).astype(f32)) / f32(255)).astype(f32)).astype(np.int64))
    c = np.frombuffer(np.uint32(0x3EC8C8C9).tobytes(), dtype=f32)[0]
    assert c == f32(100.0) / f32(255.0)
    got = np.trunc((np.abs(d) * c).astype(f32)).astype(np.int64)
    assert np.array_equal(got, ref) and ref.max() == 100
    # the neighbouring floats do NOT have the property: the constant is not arbitrary
    for other in (np.nextafter(c, f32(0)), np.nextafter(c, f32(1))):
        assert not np.array_equal(np.trunc((np.abs(d) * other).astype(f32)).astype(np.int64), ref)


def test_avx512_scan_reports_exactly_what_the_scalar_loop_reports(oracle, tmp_path):
    """The CPU baseline's AVX-512 VPOPCNTDQ scan (oracle/hvd_oracle.c) only finds candidates faster; every record still
    comes from the scalar statement. Run the same searches in two processes, one with the vector path forced off."""
    import subprocess
    import sys

    from conftest import ROOT

    script = tmp_path / "run.py"
    script.write_text(
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from oracle import oracle as O\n"
        "from hvd_amd import synth\n"
        "db, _ = synth.hash_db_clustered(5003, 50, 20, seed=8)\n"
        "u, _ = synth.hash_db(20001, seed=3)\n"
        "grp = (np.arange(5003) // 3).astype(np.int32)\n"
        "r = [O.allpairs(db, 31), O.allpairs(db, 31, group=grp), O.allpairs(db, 40, rows=(1001, 3007)),\n"
        "     O.allpairs(u, 31, num_threads=3), O.allpairs(db[:63], 31), O.allpairs(db[:64], 255), O.allpairs(db[:71], 0)]\n"
        "print(int(O.uses_avx512()), O.allpairs_count(u, 31, 2))\n"
        "np.save(sys.argv[1], np.concatenate([x.view(np.uint32).ravel() for x in r]))\n" % ROOT)
    outs = []
    for k, env_extra in enumerate(({}, {"HVD_ORACLE_NO_AVX512": "1"})):
        out = tmp_path / f"r{k}.npy"
        p = subprocess.run([sys.executable, str(script), str(out)], env={**os.environ, **env_extra}, capture_output=True,
                           check=True)
        outs.append((p.stdout.split(), np.load(out)))
    assert outs[1][0][0] == b"0"  # the switch works
    assert outs[0][0][1] == outs[1][0][1]
    assert np.array_equal(outs[0][1], outs[1][1]) and outs[0][1].size > 1000

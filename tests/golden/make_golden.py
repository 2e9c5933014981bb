"""Generates tests/golden/*.npz.

The reference's own golden vectors live in its un-checked-out `tests/testdb` submodule and
its arithmetic in the absent wheel `hvdaccelerators==0.4.0`, so neither can produce vectors
here (PARITY UNPINNED, see oracle/hvd_oracle.c). These fixtures therefore freeze the output
of the C oracle, and this script refuses to write them unless the independent numpy
restatement (oracle/pdq_numpy.py) reproduces every hash, quality and DCT coefficient
bit-for-bit. They guard the oracle (and through it the GPU path) against regressions.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hvd_amd  # noqa: E402,F401
from hvd_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import pdq_numpy as P  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def check_frames(frames, hashes, quality, coeffs, fma=False):
    for f in range(len(frames)):
        h, q, b = (P.hash_rgb if frames.ndim == 4 else P.hash_gray)(frames[f], fma)
        assert h == hashes[f].tobytes(), f"hash mismatch frame {f}"
        assert q == quality[f], f"quality mismatch frame {f}"
        assert np.array_equal(b.ravel().view(np.uint32), coeffs[f].view(np.uint32)), f"coeff mismatch frame {f}"


def main():
    # 1) gray 64x64 frames spanning the quality range, plus hand-made edge frames
    fr = synth.frames_gray(40, seed=101)
    edge = np.zeros((8, 64, 64), np.uint8)
    edge[0] = 0
    edge[1] = 255
    edge[2] = np.arange(64, dtype=np.uint8)[None, :] * 4          # horizontal ramp
    edge[3] = np.arange(64, dtype=np.uint8)[:, None] * 4          # vertical ramp
    edge[4] = ((np.indices((64, 64)).sum(0) % 2) * 255).astype(np.uint8)  # checkerboard
    edge[5] = np.where(np.indices((64, 64))[1] < 32, 0, 255)      # step edge: gradient exactly 255
    edge[6] = np.where(np.indices((64, 64))[0] % 8 < 4, 51, 102)  # differences of exactly 51 (=> 20.0)
    edge[7] = (np.indices((64, 64))[1] * 51 // 16).astype(np.uint8)
    fr = np.concatenate([fr, edge])
    h, q, c = O.hash_frames(fr, want_coeffs=True)
    check_frames(fr, h, q, c)
    # the opt-in "fma" DCT mode (upstream's arm64 numerics), verified the same way
    hf, qf, cf = O.hash_frames(fr, want_coeffs=True, fma=True)
    check_frames(fr, hf, qf, cf, fma=True)
    assert np.array_equal(q, qf)
    np.savez_compressed(os.path.join(OUT, "pdq_gray64.npz"), frames=fr, hashes=h, quality=q, coeffs=c,
                        hashes_fma=hf, coeffs_fma=cf)

    # 2) rgb24 frames that need the Jarosz down-sampler: the reference's 512x512 plus odd sizes
    rgb512 = synth.frames_rgb(2, seed=102, h=512, w=512)
    h5, q5, c5 = O.hash_frames(rgb512, want_coeffs=True)
    check_frames(rgb512, h5, q5, c5)
    h5f, q5f, c5f = O.hash_frames(rgb512, want_coeffs=True, fma=True)
    check_frames(rgb512, h5f, q5f, c5f, fma=True)
    np.savez_compressed(os.path.join(OUT, "pdq_rgb512.npz"), frames=rgb512, hashes=h5, quality=q5, coeffs=c5,
                        hashes_fma=h5f, coeffs_fma=c5f)
    rgbodd = synth.frames_rgb(3, seed=103, h=100, w=333)
    ho, qo, co = O.hash_frames(rgbodd, want_coeffs=True)
    check_frames(rgbodd, ho, qo, co)
    rgb64 = synth.frames_rgb(3, seed=104, h=64, w=64)
    h6, q6, c6 = O.hash_frames(rgb64, want_coeffs=True)
    check_frames(rgb64, h6, q6, c6)
    np.savez_compressed(os.path.join(OUT, "pdq_rgb_misc.npz"), frames_odd=rgbodd, hashes_odd=ho, quality_odd=qo,
                        frames_64=rgb64, hashes_64=h6, quality_64=q6)

    # 3) hash DB with planted near-duplicates and its exact pair list at tolerance 31
    db, planted = synth.hash_db(3000, seed=105, plant_fraction=0.02)
    pairs = O.allpairs(db, 31)
    brute = [(i, j, P.hamming(db[i].tobytes(), db[j].tobytes())) for i, j, _ in planted[:, :3]]
    for (i, j, d), k in zip(brute, planted[:, 2]):
        assert d <= k, "planted distance exceeds its flip count"
    want = sorted((int(i), int(j)) for (i, j, d) in brute if d <= 31)
    got = sorted((int(p["i"]), int(p["j"])) for p in pairs)
    assert set(want) <= set(got), "oracle misses a planted pair"
    np.savez_compressed(os.path.join(OUT, "hamming_db.npz"), db=db, planted=planted, pairs=pairs)

    # 4) video library with ragged lengths (0-frame videos included) and its match records
    frames, offsets, vplanted = synth.video_hashes(60, seed=106, frames_per_video=(0, 24), copy_fraction=0.2)
    recs = O.match_videos(frames, offsets, 31)
    np.savez_compressed(os.path.join(OUT, "video_match.npz"), frames=frames, offsets=offsets, planted=vplanted,
                        records=recs)
    print("golden fixtures written:", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
    print("gray64 quality:", q.tolist())
    print("pairs:", len(pairs), "video records:", len(recs))


if __name__ == "__main__":
    main()

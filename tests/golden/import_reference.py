#!/usr/bin/env python
"""Pin the oracle to the REAL reference -- one command, for a build container that has what this one lacks.

The arithmetic of the hot path lives in the PyPI wheel `hvdaccelerators==0.4.0` (reference pyproject.toml:36,
uv.lock:186-189), which is not installable offline; the reference's golden hashes live in the un-checked-out
submodule `tests/testdb`. Where both exist, run

    python tests/golden/import_reference.py [--reference /root/reference] [--write]

It imports the reference's own binding, measures the three things this repo had to declare as policies, checks the
CPU oracle (oracle/) against the reference bit for bit, and with --write stores the reference's answers as fixtures
(tests/golden/reference_pinned.npz + .json: inputs and expected outputs only -- no reference source travels).
Exit codes: 0 = oracle and policies agree with the reference ("parity pinned"); 1 = a mismatch (printed; fix the
default or the oracle); 3 = the wheel is not importable here (nothing checked).

What is measured (reference call sites in brackets):
  comparator   matchHash on two one-frame hashes exactly 30 / 31 / 32 bits apart, tolerance 31      [vpdqpy/vpdqpy.py:49-56]
               -> "le" (31 is a hit) or "lt"; compared with hvd_amd.vpdq.MATCH_COMPARATOR
  reduction    matchHash on an asymmetric pair (query 4 frames, 1 of which matches; target 2 frames, both match)
               -> which of min / max / query / target reproduces the number; compared with MATCH_POLICY
  str grammar  str(VpdqHash) of a 2-frame hash and from_string(str(.)) round trip                    [hashing.py:30,40]
               -> compared with hvd_amd.vpdq.VpdqHash's text form (the third declared assumption)
  boundary     30 pairs of one-frame hashes at exactly 30 / 31 / 32 bits + 6 multi-frame pairs whose similarity hinges on
               a frame pair at exactly 31: inputs and the reference's matchHash answers are stored (--write) as
               tests/golden/reference_boundary.npz, which tests/test_gpu_round4.py replays through the GPU path
  frame hashes VideoHasher(1, 64|512, ...).hash_frame on this repo's seeded synthetic frames         [vpdqpy/vpdqpy.py:113-119]
               -> bit-exact vs oracle.hash_frames (strict and fma DCT modes), quality filter included
  testdb       (if <reference>/tests/testdb/videos exists and PyAV is importable) the reference's Vpdq.computeHash on
               its clips vs its stored hash texts, and the SXX_ similarity truth table          [tests/unit_tests/test_vpdqpy.py:105-145]
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--write", action="store_true", help="store the reference's answers under tests/golden/")
    args = ap.parse_args()
    try:
        from hvdaccelerators import vpdq as ref
    except Exception as exc:  # noqa: BLE001
        print(f"hvdaccelerators is not importable here ({exc!r}): nothing can be pinned. PARITY STAYS UNPINNED.")
        return 3

    import hvd_amd  # host logic only; no GPU needed
    from hvd_amd import synth
    from oracle import oracle as O

    report, ok = {}, True

    def frames_to_hash(rows: np.ndarray):
        return ref.VpdqHash.from_string(rows.tobytes().hex())

    # ---- comparator -------------------------------------------------------------------------------------------
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    sims = {}
    for d in (30, 31, 32):
        other = synth.flip_bits(base, np.array([d]), rng)
        assert O.hamming256(base.tobytes(), other.tobytes()) == d
        sims[d] = float(ref.matchHash(frames_to_hash(base), frames_to_hash(other), 31))
    comparator = "le" if sims[31] > 0 else "lt"
    report["comparator"] = {"similarity_at_distance": sims, "reference": comparator, "ours": hvd_amd.vpdq.MATCH_COMPARATOR}
    if sims[30] <= 0 or sims[32] > 0:
        print("UNEXPECTED: the reference does not threshold frame distance at 31:", sims)
        ok = False
    if comparator != hvd_amd.vpdq.MATCH_COMPARATOR:
        print(f"POLICY MISMATCH: comparator is {comparator!r} upstream, default here is {hvd_amd.vpdq.MATCH_COMPARATOR!r} "
              "(set HVD_MATCH_COMPARATOR or change vpdq.MATCH_COMPARATOR)")
        ok = False

    # ---- reduction of the two percentages ---------------------------------------------------------------------------
    q = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    t = np.stack([synth.flip_bits(q[:1], np.array([3]), rng)[0], synth.flip_bits(q[:1], np.array([5]), rng)[0]])
    got = float(ref.matchHash(frames_to_hash(q), frames_to_hash(t), 31))
    got_rev = float(ref.matchHash(frames_to_hash(t), frames_to_hash(q), 31))
    cand = {"min": 25.0, "max": 100.0, "query": 25.0, "target": 100.0}  # q matched 1/4, t matched 2/2
    cand_rev = {"min": 25.0, "max": 100.0, "query": 100.0, "target": 25.0}
    fits = [p for p in cand if abs(cand[p] - got) < 1e-6 and abs(cand_rev[p] - got_rev) < 1e-6]
    report["reduction"] = {"matchHash(q,t)": got, "matchHash(t,q)": got_rev, "reference": fits, "ours": hvd_amd.vpdq.MATCH_POLICY}
    if hvd_amd.vpdq.MATCH_POLICY not in fits:
        print(f"POLICY MISMATCH: reduction upstream is one of {fits}, default here is {hvd_amd.vpdq.MATCH_POLICY!r}")
        ok = False

    # ---- str(VpdqHash): the text form the reference stores and parses (hashing.py:30,40) -------------------------------
    two = rng.integers(0, 256, (2, 32), dtype=np.uint8)
    ref_text = str(frames_to_hash(two))
    ours_text = str(hvd_amd.vpdq.VpdqHash(two.tobytes()))
    round_trip = ref.VpdqHash.from_string(ref_text).bytes == two.tobytes()
    parses_ours = None
    try:
        parses_ours = ref.VpdqHash.from_string(ours_text).bytes == two.tobytes()
    except Exception as exc:  # noqa: BLE001
        parses_ours = repr(exc)
    report["str_grammar"] = {"reference": ref_text, "ours": ours_text, "equal": ref_text == ours_text,
                             "reference_round_trip": round_trip, "reference_parses_ours": parses_ours,
                             "ours_parses_reference": hvd_amd.vpdq.VpdqHash.from_string(ref_text).bytes == two.tobytes()}
    if ref_text != ours_text:
        print(f"GRAMMAR MISMATCH: str(VpdqHash) upstream is {ref_text[:80]!r}..., here {ours_text[:80]!r}...")
        ok = False

    # ---- boundary vectors: what the GPU suite replays ------------------------------------------------------------------
    bq, bt, bsim, bdist = [], [], [], []
    for d in (30, 31, 32):
        for _ in range(10):
            x = rng.integers(0, 256, (1, 32), dtype=np.uint8)
            y = synth.flip_bits(x, np.array([d]), rng)
            bq.append(x)
            bt.append(y)
            bdist.append(d)
            bsim.append(float(ref.matchHash(frames_to_hash(x), frames_to_hash(y), 31)))
    mq, mt, msim = [], [], []
    for k in range(6):  # multi-frame pairs: k+1 of the query's 6 frames have their only partner at exactly 31 bits
        qv = rng.integers(0, 256, (6, 32), dtype=np.uint8)
        tv = rng.integers(0, 256, (5, 32), dtype=np.uint8)
        for f in range(min(k + 1, 5)):
            tv[f] = synth.flip_bits(qv[f:f + 1], np.array([31]), rng)[0]
        mq.append(qv)
        mt.append(tv)
        msim.append(float(ref.matchHash(frames_to_hash(qv), frames_to_hash(tv), 31)))
    boundary = {"one_q": np.concatenate(bq), "one_t": np.concatenate(bt), "one_dist": np.array(bdist, dtype=np.int32),
                "one_similarity": np.array(bsim), "multi_q": np.stack(mq), "multi_t": np.stack(mt),
                "multi_similarity": np.array(msim), "tolerance": np.array([31], dtype=np.int32)}
    report["boundary"] = {"one_frame": dict(zip(map(str, bdist), bsim)), "multi_frame_similarity": msim}
    if args.write:
        np.savez_compressed(os.path.join(HERE, "reference_boundary.npz"), **boundary)

    # ---- frame hashes -------------------------------------------------------------------------------------------------
    store = {}
    for name, fr in (("gray64", synth.frames_gray(200, seed=2)), ("rgb512", synth.frames_rgb(12, seed=6))):
        h, w = fr.shape[1:3]
        rgb = fr if fr.ndim == 4 else np.repeat(fr[..., None], 3, axis=3)  # gray is defined as R=G=B
        hasher = ref.VideoHasher(1, w, h, 1)
        for f in rgb:
            hasher.hash_frame(bytes(np.ascontiguousarray(f)))
        ref_bytes = hasher.finish().bytes
        res = {}
        for mode, fma in (("strict", False), ("fma", True)):
            ho, qo = O.hash_frames(fr, fma=fma)
            res[mode] = ho[qo >= 31].tobytes() == ref_bytes
        report[f"frames_{name}"] = res
        store[f"{name}_reference_hash"] = np.frombuffer(ref_bytes, dtype=np.uint8)
        if not res["strict"]:
            print(f"ORACLE MISMATCH on {name}: strict mode {res['strict']}, fma mode {res['fma']} "
                  "(if only fma agrees, this platform contracts the DCT: HVD_PDQ_DCT_MODE=fma is the pinned mode here)")
            ok = ok and res["fma"]

    # ---- the reference's own fixtures -----------------------------------------------------------------------------------
    vids = os.path.join(args.reference, "tests", "testdb", "videos")
    if os.path.isdir(vids):
        try:
            sys.path.insert(0, os.path.join(args.reference, "src"))
            from hydrusvideodeduplicator.vpdqpy.vpdqpy import Vpdq  # needs PyAV

            table = {}
            names = sorted(f for f in os.listdir(vids) if not f.startswith("."))
            hashes = {n: Vpdq.computeHash(os.path.join(vids, n)) for n in names}
            for a in names:
                for b in names:
                    table[f"{a}|{b}"] = float(Vpdq.match_hash(hashes[a], hashes[b]))
            store["testdb_names"] = np.array(names)
            report["testdb"] = {"videos": len(names)}
            if args.write:
                json.dump({"hashes": {n: str(h) for n, h in hashes.items()}, "similarity": table},
                          open(os.path.join(HERE, "reference_testdb.json"), "w"))
        except Exception as exc:  # noqa: BLE001
            report["testdb"] = {"skipped": repr(exc)}
    else:
        report["testdb"] = {"skipped": "tests/testdb submodule not checked out"}

    print(json.dumps(report, indent=1, default=str))
    if args.write:
        np.savez_compressed(os.path.join(HERE, "reference_pinned.npz"), **store)
        json.dump(report, open(os.path.join(HERE, "reference_pinned.json"), "w"), indent=1, default=str)
    print("PARITY PINNED: oracle and policies agree with hvdaccelerators" if ok else "MISMATCH: see above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

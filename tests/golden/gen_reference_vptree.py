#!/usr/bin/env python
"""Generates tests/golden/reference_vptree.npz by running the REFERENCE's own Python code in this container.

`/root/reference/src/hydrusvideodeduplicator/db/{DedupeDB,vptree}.py` are pure Python and import cleanly once the one
absent native module is stubbed. This script stubs `hvdaccelerators.vpdq.matchHashBytes` with the CPU oracle (frame
comparator `<=`, reduction `min`: this repo's declared policies -- the arithmetic is NOT the reference's, the wheel is
absent) and then lets the reference do everything else itself:

  * `DedupeDb.create_tables / add_file / add_perceptual_hash / associate_file_with_perceptual_hash` build the database
    and, through `VpTreeManager.add_leaf`, the real vantage-point tree (db/DedupeDB.py:241-324, db/vptree.py:155-283);
  * `VpTreeManager.search_file(hash_id, fix_vpdq_similarity(threshold))` is run for every file at thresholds 50 and 75
    (db/vptree.py:865-902, the call of dedup.py:475);
  * `DedupeDb.upgrade_db()` converts a pre-0.10.0 database of JSON-format hashes (db/DedupeDB.py:528-584).

The fixture holds inputs and the reference's outputs only (data, no source). tests/test_reference_fixture.py replays
the inputs through this repo's facade / adapter / converter and compares. Run:  python tests/golden/gen_reference_vptree.py
"""
import json
import os
import random
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("HVD_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "src"))

from oracle import oracle as O  # noqa: E402

import hvd_amd  # noqa: E402  (host-side helpers only: synth, no GPU)
from hvd_amd import synth  # noqa: E402


def match_hash_bytes(a: bytes, b: bytes, tol: int) -> float:
    na, nb = len(a) // 32, len(b) // 32
    if na == 0 or nb == 0:
        return 0.0
    q, t = O.match_two(a, b, int(tol))
    return min(q * 100.0 / na, t * 100.0 / nb)


stub = types.ModuleType("hvdaccelerators")
vp = types.ModuleType("hvdaccelerators.vpdq")
vp.matchHashBytes = match_hash_bytes
stub.vpdq = vp
sys.modules["hvdaccelerators"] = stub
sys.modules["hvdaccelerators.vpdq"] = vp

from hydrusvideodeduplicator.db import DedupeDB, vptree  # noqa: E402


def build_library(n_videos=90, seed=31):
    frames, offsets, _ = synth.video_hashes(n_videos, seed=seed, frames_per_video=(1, 14), copy_fraction=0.35)
    blobs = [frames[offsets[v]:offsets[v + 1]].tobytes() for v in range(n_videos)]
    blobs[9] = blobs[4]      # two files, one perceptual hash
    blobs[40] = blobs[4]     # ... three
    blobs[17] = b""          # empty hashes (every frame below the quality bar)
    blobs[18] = b""
    return blobs


def main():
    random.seed(7)  # the reference samples vantage points with the global RNG (db/vptree.py:431-441)
    blobs = build_library()
    file_hashes = [f"{v:064x}" for v in range(len(blobs))]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        db = DedupeDB.DedupeDb(Path(tmp), "videohashes.sqlite")
        db.init_connection()
        db.create_tables()
        # the reference's own ingestion (dedup.py:396-432 without the progress bar)
        for fh, blob in zip(file_hashes, blobs):
            db.add_to_phashed_files_queue(fh, blob)
        for fh, blob in db.execute("SELECT file_hash, phash FROM phashed_file_queue").fetchall():
            db.add_file(fh)
            db.add_perceptual_hash(blob)
            db.associate_file_with_perceptual_hash(fh, blob)
            db.execute("DELETE FROM phashed_file_queue WHERE file_hash = :file_hash AND phash = :phash",
                       {"file_hash": fh, "phash": blob})
        db.commit()
        tree = vptree.VpTreeManager(db)
        if tree.maintenance_due(51):
            tree.maintain_tree()
        tables = {}
        for name, sql in (("files", "SELECT hash_id, file_hash FROM files ORDER BY hash_id"),
                          ("map", "SELECT phash_id, hash_id FROM shape_perceptual_hash_map ORDER BY hash_id"),
                          ("cache", "SELECT hash_id, searched_distance FROM shape_search_cache ORDER BY hash_id")):
            tables[name] = db.execute(sql).fetchall()
        assert all(sd is None for _, sd in tables["cache"])
        n_tree = db.execute("SELECT COUNT(*) FROM shape_vptree").fetchone()[0]
        hash_ids = [r[0] for r in tables["files"]]
        for thr in (50.0, 75.0):
            d = vptree.fix_vpdq_similarity(thr)
            rows = []
            for hid in hash_ids:
                res = tree.search_file(hid, max_hamming_distance=d)
                assert res[0] == (hid, 0)
                rows.extend((hid, int(o), int(dist)) for o, dist in res)
            out[f"search_thr{int(thr)}"] = np.array(rows, dtype=np.int64)
        # identical-phash shortcut
        rows = []
        for hid in hash_ids:
            rows.extend((hid, int(o), int(dist)) for o, dist in tree.search_file(hid, 0))
        out["search_d0"] = np.array(rows, dtype=np.int64)
        out["files_hash_id"] = np.array([r[0] for r in tables["files"]], dtype=np.int64)
        out["files_file_hash"] = np.array([r[1] for r in tables["files"]])
        out["map"] = np.array(tables["map"], dtype=np.int64)
        out["tree_nodes"] = np.int64(n_tree)
        db.close()

    # pre-0.10.0 format: JSON of "hex,quality,frame", bytes reversed; converted by the reference's upgrade_db
    rng = np.random.default_rng(5)
    old_rows, old_queue = [], []
    for k in range(6):
        nf = int(rng.integers(0, 7))
        fr = rng.integers(0, 256, (nf, 32), dtype=np.uint8)
        q = rng.choice([0, 30, 31, 50, 100], size=nf)
        js = json.dumps([f"{bytes(f[::-1]).hex()},{int(qq)},{i}" for i, (f, qq) in enumerate(zip(fr, q))])
        (old_rows if k < 4 else old_queue).append(js)
    with tempfile.TemporaryDirectory() as tmp:
        db = DedupeDB.DedupeDb(Path(tmp), "old.sqlite")
        db.init_connection()
        db.create_tables()
        db.set_version("0.9.0")
        for k, js in enumerate(old_rows):
            db.execute("INSERT INTO shape_perceptual_hashes ( phash_id, phash ) VALUES ( ?, ? )", (k + 1, js))
        for k, js in enumerate(old_queue):
            db.execute("INSERT INTO phashed_file_queue ( file_hash, phash ) VALUES ( ?, ? )", (f"q{k}", js))
        db.commit()
        assert db.upgrade_db() is True
        conv_rows = [bytes(r[0]) for r in db.execute("SELECT phash FROM shape_perceptual_hashes ORDER BY phash_id").fetchall()]
        conv_queue = [bytes(r[0]) for r in db.execute("SELECT phash FROM phashed_file_queue ORDER BY file_hash").fetchall()]
        out["upgraded_version"] = np.array(db.get_version())
        db.close()

    def pack(blist):
        lens = np.array([len(b) for b in blist], dtype=np.int64)
        return np.frombuffer(b"".join(blist), dtype=np.uint8), lens

    out["blobs_data"], out["blobs_len"] = pack(blobs)
    out["old_json"] = np.array(old_rows + old_queue)
    out["old_converted_data"], out["old_converted_len"] = pack(conv_rows + conv_queue)
    out["meta"] = np.array(json.dumps({
        "reference_version": DedupeDB.__version__, "matcher": "oracle.match_two, comparator le, reduction min, tolerance 31",
        "generator": "tests/golden/gen_reference_vptree.py", "random_seed": 7}))
    path = os.path.join(HERE, "reference_vptree.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()

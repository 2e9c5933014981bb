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
vp.matchHash = lambda q, t, tol: match_hash_bytes(q.bytes, t.bytes, tol)
vp.VpdqHash = hvd_amd.VpdqHash          # the value type is host-side Python in this repo: no GPU involved
vp.VideoHasher = object                 # never constructed here (decoding needs PyAV)
stub.vpdq = vp
sys.modules["hvdaccelerators"] = stub
sys.modules["hvdaccelerators.vpdq"] = vp
sys.modules.setdefault("av", types.ModuleType("av"))  # imported at module level by vpdqpy.py; never used here

from hydrusvideodeduplicator import dedup  # noqa: E402
from hydrusvideodeduplicator.db import DedupeDB, vptree  # noqa: E402


class _FakeHydrus:
    """Stands in for hydrus_api.Client: records what dedup.mark_videos_as_duplicates sends (dedup.py:385-394)."""

    def __init__(self):
        self.relationships = []

    def set_file_relationships(self, rels):
        self.relationships.extend((r["hash_a"], r["hash_b"]) for r in rels)


class _FakeClient:
    def __init__(self):
        self.client = _FakeHydrus()


def run_reference_pipeline(blobs, file_hashes, threshold, tree_factory=None):
    """The reference's OWN process_phashed_file_queue + find_potential_duplicates (dedup.py:396-502) on a fresh database.
    tree_factory: None = the reference's VpTreeManager; else a callable db -> manager that replaces it in both modules
    that construct one (dedup.py and db/DedupeDB.py) -- the import swap of INTEGRATION.md 3c."""
    saved = (dedup.vptree.VpTreeManager, DedupeDB.VpTreeManager)
    try:
        if tree_factory is not None:
            dedup.vptree.VpTreeManager = tree_factory
            DedupeDB.VpTreeManager = tree_factory
        # the reference keeps a process-wide cache of temp-table names that belongs to ONE connection ("do this once per
        # program run", db/vptree.py:136-149): a fresh database needs a fresh cache
        vptree.TemporaryIntegerTableNameCache()
        with tempfile.TemporaryDirectory() as tmp:
            db = DedupeDB.DedupeDb(Path(tmp), "videohashes.sqlite")
            db.init_connection()
            db.create_tables()
            for fh, blob in zip(file_hashes, blobs):
                db.add_to_phashed_files_queue(fh, blob)
            db.commit()
            client = _FakeClient()
            dd = dedup.HydrusVideoDeduplicator(db, client)
            dd.threshold = threshold
            dd.process_phashed_file_queue()
            returned = dd.find_potential_duplicates()
            db.commit()
            cache = db.execute("SELECT hash_id, searched_distance FROM shape_search_cache ORDER BY hash_id").fetchall()
            db.close()
        return client.client.relationships, returned, cache
    finally:
        dedup.vptree.VpTreeManager, DedupeDB.VpTreeManager = saved


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

    # ---- the reference's real pipeline loop: its own tree, then THIS repo's facade swapped in -------------------------
    import hvd_amd.vptree as our_vptree

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_sqlite_adapter import OracleMatcher

    rel_ref, ret_ref, cache_ref = run_reference_pipeline(blobs, file_hashes, 50.0)
    rel_fac, ret_fac, cache_fac = run_reference_pipeline(
        blobs, file_hashes, 50.0, tree_factory=lambda db: our_vptree.VpTreeManager(db, matcher=OracleMatcher(O)))
    assert cache_ref == cache_fac and all(sd == 51 for _, sd in cache_ref)
    assert set(rel_ref) <= set(rel_fac), "the facade must report every pair the reference's tree reports"
    # the facade's answer is the brute-force truth, found from both sides
    truth = set()
    for a in range(len(blobs)):
        for b in range(len(blobs)):
            if a != b and int(match_hash_bytes(blobs[a], blobs[b], 31)) >= 50:
                truth.add((file_hashes[a], file_hashes[b]))
    assert set(rel_fac) == truth and len(rel_fac) == len(truth) and ret_fac == len(truth) // 2
    out["pipeline_pairs_reference_tree"] = np.array(sorted(set(rel_ref)))
    out["pipeline_pairs_facade"] = np.array(sorted(set(rel_fac)))
    out["pipeline_return_reference_tree"] = np.int64(ret_ref)
    out["pipeline_return_facade"] = np.int64(ret_fac)
    print(f"reference pipeline: its own tree reports {len(set(rel_ref))} directed pairs (returns {ret_ref}); with the facade "
          f"swapped in {len(set(rel_fac))} = brute force (returns {ret_fac})")

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

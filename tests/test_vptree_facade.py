"""The unchanged pipeline's search loop (reference dedup.py:445-502) driven against the VpTreeManager-compatible
facade (hvd_amd.vptree) on the f2 SQLite fixture: it must produce the brute-force pair set, incrementally too, at a
per-file cost of a lookup. Also f2's queue ingestion and the pre-0.10 perceptual-hash format.
CPU: the oracle stands in for the GPU matcher (tests may use the oracle); gpu-marked: the real kernels."""
import json
import time

import numpy as np
import pytest

from test_sqlite_adapter import SCHEMA, OracleMatcher, brute_force, build_db


def reference_search_loop(conn, tree, threshold, hvd):
    """dedup.py:452-502, minus Hydrus I/O and progress bars: -> (directed pairs, num_similar_pairs // 2)."""
    search_threshold = hvd.vptree.fix_vpdq_similarity(threshold)
    assert search_threshold > 0 and isinstance(search_threshold, int)
    files = conn.execute(
        "SELECT hash_id FROM shape_search_cache WHERE searched_distance is NULL or searched_distance < :threshold",
        {"threshold": search_threshold}).fetchall()
    directed = []
    for (hash_id,) in files:
        result = tree.search_file(hash_id, max_hamming_distance=search_threshold)
        assert result[0] == (hash_id, 0)
        for similar_hash_id, distance in result:
            if hash_id != similar_hash_id:
                assert 1 <= distance <= search_threshold
                directed.append((hash_id, similar_hash_id))
        conn.execute("UPDATE shape_search_cache SET searched_distance = ? WHERE hash_id = ?;", (search_threshold, hash_id))
    conn.commit()
    return directed, len(directed) // 2


def file_hash_of(conn, hash_id):
    return conn.execute("SELECT file_hash FROM files WHERE hash_id = ?", (hash_id,)).fetchone()[0]


def unordered(conn, directed):
    return {tuple(sorted((file_hash_of(conn, a), file_hash_of(conn, b)))) for a, b in directed}


def check_facade(hvd, oracle, matcher):
    conn, blobs = build_db(hvd)
    n = len(blobs)
    tree = hvd.vptree.VpTreeManager(conn, matcher=matcher)
    assert tree.maintenance_due(51) is False
    tree.maintain_tree()
    # first run: every file pending; each pair is found from both sides
    directed, count = reference_search_loop(conn, tree, 50.0, hvd)
    want = brute_force(oracle, blobs, 50.0, set(range(n)))
    assert unordered(conn, directed) == set(want) and count == len(want) and len(want) >= 5
    assert len(directed) == 2 * len(want)
    # nothing pending any more
    assert reference_search_loop(conn, tree, 50.0, hvd) == ([], 0)
    # identical-phash shortcut at distance 0 (db/vptree.py:875-885): files 3 and 7 share one perceptual hash
    r0 = tree.search_file(4, 0)
    assert r0[0] == (4, 0) and set(r0) == {(4, 0), (8, 0)}
    assert tree.search_file(12, 0)[0] == (12, 0)  # empty hash: shares it with file 13 only through the shortcut
    assert set(tree.search_file(12, 51)) == {(12, 0)}  # ... and is similar to nothing through the search
    # new files arrive through the queue (dedup.py:396-432): two copies of existing videos, one unrelated, one empty
    rng = np.random.default_rng(5)
    new_blobs = [blobs[20], blobs[41][: len(blobs[41]) // 2 // 32 * 32] + rng.integers(0, 256, 32 * 4, dtype=np.uint8).tobytes(),
                 rng.integers(0, 256, 32 * 9, dtype=np.uint8).tobytes(), b""]
    for k, b in enumerate(new_blobs):
        conn.execute("INSERT INTO phashed_file_queue VALUES (?, ?)", (f"{n + k:064x}", b))
    assert hvd.sqlite_adapter.ingest_phashed_file_queue(conn, tree=tree) == len(new_blobs)
    assert conn.execute("SELECT COUNT(*) FROM phashed_file_queue").fetchone()[0] == 0
    all_blobs = blobs + new_blobs
    directed, _ = reference_search_loop(conn, tree, 50.0, hvd)  # only the new files are pending
    want_inc = brute_force(oracle, all_blobs, 50.0, set(range(n, n + len(new_blobs))))
    assert unordered(conn, directed) == set(want_inc) and len(want_inc) >= 2
    assert (f"{20:064x}", f"{n:064x}") in want_inc
    # a rebuilt facade (new process) gives the same answers from the database alone
    tree2 = hvd.vptree.VpTreeManager(conn, matcher=matcher)
    for hash_id in (1, 4, 21, n + 1, n + 2):
        assert sorted(tree2.search_file(hash_id, 51)) == sorted(tree.search_file(hash_id, 51))
    # search_perceptual_hashes with a hash that is NOT in the library (db/vptree.py:664 takes arbitrary hashes): a noisy
    # copy of video 20's hash must find file 21 (and the queued copy of it), a random hash nothing
    probe = np.frombuffer(all_blobs[20], dtype=np.uint8).reshape(-1, 32).copy()
    probe[:, 0] ^= 1
    got = dict(tree.search_perceptual_hashes([probe.tobytes()], 51))
    assert got.get(21) is not None and got.get(n + 1) is not None and all(1 <= d <= 51 for d in got.values())
    assert tree.search_perceptual_hashes([rng.integers(0, 256, 32 * 5, dtype=np.uint8).tobytes()], 51) == []
    assert tree.search_perceptual_hashes([], 51) == [] and tree.search_perceptual_hashes([b""], 51) == []
    tree.reset_search([1, 2])
    assert conn.execute("SELECT COUNT(*) FROM shape_search_cache WHERE searched_distance IS NULL").fetchone()[0] == 2
    return conn, tree


def test_vptree_facade_with_oracle_matcher(hvd, oracle):
    check_facade(hvd, oracle, OracleMatcher(oracle))


@pytest.mark.gpu
def test_vptree_facade_on_gpu(gpu, hvd, oracle):
    conn, tree = check_facade(hvd, oracle, None)
    # per-file cost once the pass is cached: a lookup + the SQL fan-out, not 18 us x tree nodes
    ids = [r[0] for r in conn.execute("SELECT hash_id FROM shape_search_cache").fetchall()]
    t = time.perf_counter()
    for _ in range(20):
        for h in ids:
            tree.search_file(h, 51)
    per_file = (time.perf_counter() - t) / (20 * len(ids))
    assert per_file < 200e-6, per_file


def test_old_format_phashes_are_converted_not_refused(hvd):
    """db/DedupeDB.py:528-584: JSON of "hex,quality,frame", bytes reversed, low quality dropped."""
    import sqlite3

    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    quality = [80, 30, 31, 0, 100]
    old = json.dumps([f"{bytes(f[::-1]).hex()},{q},{k}" for k, (f, q) in enumerate(zip(frames, quality))])
    want = b"".join(bytes(f) for f, q in zip(frames, quality) if q >= 31)
    A = hvd.sqlite_adapter
    assert A.convert_old_vpdq_to_new(old) == want and A.convert_old_vpdq_to_new(old.encode()) == want
    assert A.convert_old_vpdq_to_new("[]") == b""
    assert A.is_old_format(old) and not A.is_old_format(want) and not A.is_old_format(b"")
    conn = sqlite3.connect(":memory:")
    for stmt in SCHEMA:
        conn.execute(stmt)
    conn.execute("INSERT INTO files VALUES (1, 'aa')")
    conn.execute("INSERT INTO shape_perceptual_hashes VALUES (1, ?)", (old,))
    conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (1, 1)")
    conn.execute("INSERT INTO shape_search_cache VALUES (1, NULL)")
    conn.execute("INSERT INTO phashed_file_queue VALUES ('bb', ?)", (old,))
    lib = A.load_library(conn)  # read through the conversion
    assert lib.frames.tobytes() == want and lib.offsets.tolist() == [0, 3]
    assert A.upgrade_old_phashes(conn) == 2
    assert bytes(conn.execute("SELECT phash FROM shape_perceptual_hashes").fetchone()[0]) == want
    assert bytes(conn.execute("SELECT phash FROM phashed_file_queue").fetchone()[0]) == want
    assert A.upgrade_old_phashes(conn) == 0
    # the queued file shares the library file's perceptual hash once ingested
    assert A.ingest_phashed_file_queue(conn) == 1
    assert conn.execute("SELECT COUNT(*) FROM shape_perceptual_hashes").fetchone()[0] == 1
    assert conn.execute("SELECT COUNT(*) FROM shape_perceptual_hash_map").fetchone()[0] == 2


def test_old_format_json_whose_length_is_a_multiple_of_32_is_still_json(hvd):
    """a JSON blob stored as BYTES with len % 32 == 0 was classified as raw hashes and searched as garbage.
    The decision must not depend on the length."""
    import sqlite3

    A = hvd.sqlite_adapter
    rng = np.random.default_rng(10)
    hit = 0
    for n_frames in range(1, 40):
        for width in range(1, 8):  # the frame-number field varies the length
            frames = rng.integers(0, 256, (n_frames, 32), dtype=np.uint8)
            old = json.dumps([f"{bytes(f[::-1]).hex()},{99},{10 ** width + k}" for k, f in enumerate(frames)]).encode()
            assert A.is_old_format(old) and A.is_old_format(old.decode())
            if len(old) % 32 == 0:
                hit += 1
                conn = sqlite3.connect(":memory:")
                for stmt in SCHEMA:
                    conn.execute(stmt)
                conn.execute("INSERT INTO files VALUES (1, 'aa')")
                conn.execute("INSERT INTO shape_perceptual_hashes VALUES (1, ?)", (old,))
                conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (1, 1)")
                conn.execute("INSERT INTO shape_search_cache VALUES (1, NULL)")
                assert A.load_library(conn).frames.tobytes() == frames.tobytes()
                assert A.upgrade_old_phashes(conn) == 1
                assert bytes(conn.execute("SELECT phash FROM shape_perceptual_hashes").fetchone()[0]) == frames.tobytes()
    assert hit >= 3  # the case is really exercised
    # raw hashes that merely start with '[' and end with ']' are not JSON
    raw = bytes([0x5B]) + bytes(rng.integers(0, 256, 62, dtype=np.uint8)) + bytes([0x5D])
    assert not A.is_old_format(raw)
    assert not A.is_old_format(b"[" + b"a" * 30 + b"]")  # ASCII, bracketed, 32 bytes, but not JSON
    assert not A.is_old_format(b"[1, 2, 3, 4, 5, 6, 7, 8, 9, 10 ]")  # JSON, 32 bytes, but not a list of strings


def test_compute_hash_rejects_encoded_video_input(hvd):
    """The reference's caller passes the encoded file (dedup.py:76); decoding is out of scope and must say so."""
    for bad in (b"\x00" * 100, "video.mp4", bytearray(12)):
        with pytest.raises(ValueError, match="encoded video"):
            hvd.Vpdq.computeHash(bad)
    with pytest.raises(ValueError):
        hvd.Vpdq.computeHash(None)


def test_comparator_policy_host_side(hvd, monkeypatch):
    assert hvd.vpdq.frame_max_dist(31) == 31 and hvd.vpdq.frame_max_dist(31.0, "lt") == 30
    assert hvd.vpdq.frame_max_dist(0, "lt") == -1
    with pytest.raises(ValueError):
        hvd.vpdq.frame_max_dist(31, "gt")
    monkeypatch.setattr(hvd.vpdq, "MATCH_COMPARATOR", "lt")
    assert hvd.matchHashBytes(b"\0" * 32, b"\0" * 32, 0) == 0.0  # decided on the host: nothing is < 0
    with pytest.raises(ValueError):
        hvd.matchHashBytes(b"\0" * 31, b"\0" * 32, 0)


def test_pin_script_reports_unpinned_without_the_wheel():
    """tests/golden/import_reference.py is the one-command flip to "parity pinned"; without hvdaccelerators it must
    say so (exit code 3) instead of pretending."""
    import importlib.util
    import os
    import subprocess
    import sys

    try:
        installed = importlib.util.find_spec("hvdaccelerators") is not None
    except ValueError:  # a stub module without a spec left in sys.modules
        installed = False
    if installed:
        pytest.skip("the wheel is installed: run the script itself")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "golden", "import_reference.py")], capture_output=True, text=True)
    assert r.returncode == 3 and "UNPINNED" in r.stdout


# ---------------------------------------------------------------- round 3: scale + coexistence ------------------------
def _tree_rows(conn):
    return {r[0]: r[1:] for r in conn.execute(
        "SELECT phash_id, parent_id, radius, inner_id, inner_population, outer_id, outer_population FROM shape_vptree")}


def _check_tree_is_valid(conn, dist, complete=True):
    """Every perceptual hash is a node (complete=False: every node is a perceptual hash -- a tree that add_leaf stopped
    extending at MAX_TREE_WALK); exactly one root; every node hangs on the side of each ancestor that its distance
    to that ancestor says (inner iff <= radius); populations count the descendants. This is what the reference's search
    (db/vptree.py:707-777) relies on."""
    rows = _tree_rows(conn)
    blobs = {pid: bytes(b) for pid, b in conn.execute("SELECT phash_id, phash FROM shape_perceptual_hashes")}
    assert set(rows) == set(blobs) if complete else set(rows) <= set(blobs)
    roots = [p for p, r in rows.items() if r[0] is None]
    assert len(roots) == 1

    def subtree(p):
        if p is None:
            return []
        _, _, inner, _, outer, _ = rows[p]
        return [p] + subtree(inner) + subtree(outer)

    seen = subtree(roots[0])
    assert sorted(seen) == sorted(rows)  # connected, no node twice
    for p, (parent, radius, inner, ipop, outer, opop) in rows.items():
        ins, outs = subtree(inner), subtree(outer)
        assert ipop == len(ins) and opop == len(outs), p
        for c in (inner, outer):
            if c is not None:
                assert rows[c][0] == p
        if inner is not None:
            assert radius is not None
        for x in ins:
            assert dist(blobs[x], blobs[p]) <= radius, (x, p)
        for x in outs:
            assert dist(blobs[x], blobs[p]) > radius, (x, p)


def test_add_leaf_keeps_the_reference_tree_valid(hvd, oracle):
    """add_leaf used to leave shape_vptree untouched, so a user who went back to the reference's tree
    lost every hash added meanwhile. Now each leaf is inserted by the reference's rule; the table must be a valid tree."""
    import sqlite3

    m = OracleMatcher(oracle)
    frames, offsets, _ = hvd.synth.video_hashes(70, seed=31, frames_per_video=10, copy_fraction=0.4)
    blobs = dedupe_keep_order([frames[offsets[v]:offsets[v + 1]].tobytes() for v in range(70)] + [b""])
    conn = sqlite3.connect(":memory:")
    for stmt in SCHEMA:
        conn.execute(stmt)
    for k, b in enumerate(blobs):
        conn.execute("INSERT INTO phashed_file_queue VALUES (?, ?)", (f"{k:064x}", b))
    # the reference creates one manager per inserted file (db/DedupeDB.py:303-304): do the same
    class PerFile:
        def add_leaf(self, pid, blob):
            hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=True).add_leaf(pid, blob)

    assert hvd.sqlite_adapter.ingest_phashed_file_queue(conn, tree=PerFile()) == len(blobs)
    _check_tree_is_valid(conn, m.calculate_distance)
    assert conn.execute("SELECT COUNT(*) FROM shape_vptree").fetchone()[0] == len(blobs)
    # a second file with a known hash does not touch the tree
    before = _tree_rows(conn)
    conn.execute("INSERT INTO phashed_file_queue VALUES (?, ?)", ("ff" * 32, blobs[5]))
    hvd.sqlite_adapter.ingest_phashed_file_queue(conn, tree=PerFile())
    assert _tree_rows(conn) == before
    # opting out leaves the table alone
    conn.execute("INSERT INTO shape_perceptual_hashes VALUES (9999, ?)", (bytes(64),))
    hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=False).add_leaf(9999, bytes(64))
    assert 9999 not in _tree_rows(conn)
    # ... but not silently: the hash is on record as missing from the reference's tree, a manager that maintains
    # the tree warns about it, and the record empties itself once the reference has rebuilt its tree
    import warnings

    assert [r[0] for r in conn.execute("SELECT phash_id FROM hvd_vptree_skipped")] == [9999]
    hvd.vptree.VpTreeManager(conn, matcher=m).add_leaf(5, blobs[4])  # (a hash that IS in the tree leaves no mark)
    assert conn.execute("SELECT COUNT(*) FROM hvd_vptree_skipped").fetchone()[0] == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=True).tree_incomplete
        assert any("missing from shape_vptree" in str(x.message) for x in w)
    conn.execute("INSERT INTO shape_vptree ( phash_id, parent_id, radius, inner_id, inner_population, outer_id, outer_population ) "
                 "VALUES ( 9999, -1, NULL, NULL, 0, NULL, 0 )")
    assert not hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=True).tree_incomplete
    assert conn.execute("SELECT COUNT(*) FROM hvd_vptree_skipped").fetchone()[0] == 0


def dedupe_keep_order(xs):
    out = []
    for x in xs:
        if x not in out:
            out.append(x)
    return out


def test_reference_tree_and_facade_alternate_on_one_database_without_losing_a_hash():
    """LIVE (build container only): files are added alternately through the reference's own VpTreeManager and through the
    facade on ONE SQLite file; afterwards the reference's own search, on the tree both of them wrote, finds every
    near-duplicate the reference finds on a tree it built alone, for files inserted by either side."""
    import importlib.util
    import os
    import sys

    ref = os.environ.get("HVD_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "src", "hydrusvideodeduplicator")):
        pytest.skip("the reference tree is not present (GPU box)")
    here = os.path.dirname(os.path.abspath(__file__))
    before = set(sys.modules)
    spec = importlib.util.spec_from_file_location("gen_reference_vptree", os.path.join(here, "golden", "gen_reference_vptree.py"))
    gen = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(gen)
        _alternation_checks(gen)
    finally:
        for name in set(sys.modules) - before:
            if name.split(".")[0] in ("hvdaccelerators", "av", "hydrusvideodeduplicator"):
                del sys.modules[name]


def _alternation_checks(gen):
    import random
    import tempfile
    from pathlib import Path

    import hvd_amd.vptree as ours
    from oracle import oracle as O

    blobs = dedupe_keep_order(gen.build_library(n_videos=50, seed=123))
    m = OracleMatcher(O)

    def build(tree_for_file):
        d = Path(tempfile.mkdtemp())
        db = gen.DedupeDB.DedupeDb(d, "a.sqlite")
        db.init_connection()
        db.create_tables()
        for k, b in enumerate(blobs):
            name = f"{k:064x}"
            db.add_file(name)
            db.add_perceptual_hash(b)
            # DedupeDb.associate_file_with_perceptual_hash (db/DedupeDB.py:287-324) with the tree class chosen per file
            gen.DedupeDB.VpTreeManager = tree_for_file(k)
            db.associate_file_with_perceptual_hash(name, b)
        db.commit()
        gen.DedupeDB.VpTreeManager = gen.vptree.VpTreeManager
        return db

    random.seed(4)  # the reference's tree uses unseeded random sampling when it rebalances; none happens at this size
    ref_only = build(lambda k: gen.vptree.VpTreeManager)
    mixed = build(lambda k: gen.vptree.VpTreeManager if k % 2 == 0 else (lambda db: ours.VpTreeManager(db, matcher=m, maintain_reference_tree=True)))
    # same rule, same distances, same insertion order -> the very same tree
    q = "SELECT phash_id, parent_id, radius, inner_id, inner_population, outer_id, outer_population FROM shape_vptree ORDER BY phash_id"
    assert mixed.execute(q).fetchall() == ref_only.execute(q).fetchall()
    _check_tree_is_valid(mixed.conn, m.calculate_distance)
    # and the reference's own search on the mixed tree sees the files the facade inserted
    def search_all(db):
        # the reference keeps a PROCESS-wide cache of temporary table names (db/vptree.py:34-70) that assumes one
        # connection per process; this test holds two databases, so start each from a clean cache
        gen.vptree.TemporaryIntegerTableNameCache()
        tree = gen.vptree.VpTreeManager(db)
        return [sorted(tree.search_file(k + 1, 51)) for k in range(len(blobs))]

    want, got = search_all(ref_only), search_all(mixed)
    assert got == want
    assert sum(len(r) - 1 for r in got) >= 10


class _RecordedMatcher:
    """A matcher that answers the all-pairs pass from precomputed records (no GPU, no oracle): the facade's host-side
    cost at library scale is what is measured."""

    def __init__(self, recs):
        self.recs = recs

    def match_videos(self, frames, offsets, max_dist):
        return self.recs


def test_facade_host_side_scales_to_100k_files(hvd):
    import sqlite3

    from hvd_amd._lib import VMATCH_DTYPE

    n = 100_000
    rng = np.random.default_rng(3)
    conn = sqlite3.connect(":memory:")
    for stmt in SCHEMA:
        conn.execute(stmt)
    blob_rows = [(v + 1, rng.integers(0, 256, 64, dtype=np.uint8).tobytes()) for v in range(n)]
    conn.executemany("INSERT INTO shape_perceptual_hashes VALUES (?, ?)", blob_rows)
    conn.executemany("INSERT INTO files VALUES (?, ?)", ((v + 1, f"{v:064x}") for v in range(n)))
    conn.executemany("INSERT INTO shape_perceptual_hash_map VALUES (?, ?)", ((v + 1, v + 1) for v in range(n)))
    conn.executemany("INSERT INTO shape_search_cache VALUES (?, NULL)", ((v + 1,) for v in range(n)))
    # 150k video-level records: every file ~3 neighbours, two frames per hash
    a = rng.integers(0, n - 1, 150_000)
    b = np.minimum(a + 1 + rng.integers(0, 50, a.size), n - 1)
    keep = a < b
    recs = np.zeros(int(keep.sum()), dtype=VMATCH_DTYPE)
    recs["a"], recs["b"] = a[keep], b[keep]
    recs["q_hits"] = rng.integers(1, 3, recs.size)
    recs["t_hits"] = rng.integers(1, 3, recs.size)
    _, first = np.unique(recs[["a", "b"]], return_index=True)
    recs = recs[np.sort(first)]
    tree = hvd.vptree.VpTreeManager(conn, matcher=_RecordedMatcher(recs))
    t = time.perf_counter()
    tree.search_file(1, 51)  # load + fold
    t_first = time.perf_counter() - t
    assert t_first < 20.0
    # spot-check the folded lists against the scalar functions
    lens = np.full(n, 2)
    for r in recs[:200]:
        d = hvd.fix_vpdq_similarity(hvd.vpdq.percent_from_hits(int(r["q_hits"]), int(r["t_hits"]), 2, 2))
        if d <= 51:
            assert (int(r["b"]) + 1, d) in tree.search_file(int(r["a"]) + 1, 51)
            assert (int(r["a"]) + 1, d) in tree.search_file(int(r["b"]) + 1, 51)
    # the reference's loop: search, then update the cache row (which moves the connection's change counter every time)
    t = time.perf_counter()
    found = 0
    for h in range(1, 20001):
        found += len(tree.search_file(h, 51)) - 1
        conn.execute("UPDATE shape_search_cache SET searched_distance = 51 WHERE hash_id = ?", (h,))
    per_file = (time.perf_counter() - t) / 20000
    assert found > 20000
    assert per_file < 150e-6, per_file  # loose on purpose (a busy CI container); bench.py reports the real figure: 3 us
    # a new file joins behind the facade's back, through this connection: the next search sees it
    conn.execute("INSERT INTO files VALUES (?, ?)", (n + 1, "ee" * 32))
    conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (?, ?)", (1, n + 1))  # shares file 1's perceptual hash
    assert (n + 1, 0) in tree.search_file(1, 0)
    assert (n + 1, 1) in tree.search_file(1, 51)
    conn.execute("DELETE FROM shape_perceptual_hash_map WHERE hash_id = ?", (n + 1,))
    assert (n + 1, 1) not in tree.search_file(1, 51)


def test_radius_101_returns_every_file_like_a_tree_that_prunes_nothing(hvd, oracle):
    """threshold 0 -> search distance 101 = "similarity below 1 %": every hash is within it, empty ones included; the
    facade stores records only for pairs with a frame hit, so this degenerate radius is completed from the library."""
    m = OracleMatcher(oracle)
    conn, blobs = build_db(hvd, n_videos=20)
    tree = hvd.vptree.VpTreeManager(conn, matcher=m)
    for h in (1, 4, 12):
        got = dict(tree.search_file(h, 101)[1:])
        want = {}
        for o in range(1, 21):
            d = m.calculate_distance(blobs[h - 1], blobs[o - 1])
            want[o] = min(d, want.get(o, 999))
        assert got == want
    assert len(tree.search_file(1, 100)) < 21


def test_add_leaf_cost_is_bounded_on_a_degenerate_tree(hvd, oracle):
    """unrelated videos all sit at distance 101, which the reference's insertion rule turns into a chain
    (its own maintain_tree rebalances; the facade's is a no-op), so add_leaf's walk grew by one SELECT + one distance per
    file already ingested. The walk is cut off at MAX_TREE_WALK: the leaf is left out of shape_vptree, the instance says
    so, the facade's own search is unaffected (it never reads the tree)."""
    import sqlite3
    import warnings

    class Counting(OracleMatcher):
        calls = 0

        def calculate_distance(self, a, b):
            Counting.calls += 1
            return super().calculate_distance(a, b)

    m = Counting(oracle)
    rng = np.random.default_rng(7)
    n = 400
    blobs = [rng.integers(0, 256, 32 * 4, dtype=np.uint8).tobytes() for _ in range(n)]  # unrelated: every distance is 101
    conn = sqlite3.connect(":memory:")
    for stmt in SCHEMA:
        conn.execute(stmt)
    for k, b in enumerate(blobs):
        conn.execute("INSERT INTO phashed_file_queue VALUES (?, ?)", (f"{k:064x}", b))
    incomplete = []

    class PerFile:  # the reference's pattern: one manager per inserted file (db/DedupeDB.py:303-304)
        def add_leaf(self, pid, blob):
            t = hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=True)
            t.add_leaf(pid, blob)
            incomplete.append(t.tree_incomplete)

    cap = hvd.vptree.VpTreeManager.MAX_TREE_WALK
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert hvd.sqlite_adapter.ingest_phashed_file_queue(conn, tree=PerFile()) == n
    assert any("shape_vptree" in str(x.message) for x in w)
    assert Counting.calls <= n * cap  # bounded per file (it was n * (n - 1) / 2 = 79 800 without the cap)
    in_tree = conn.execute("SELECT COUNT(*) FROM shape_vptree").fetchone()[0]
    assert cap <= in_tree <= cap + 1 and incomplete.count(True) == n - in_tree
    _check_tree_is_valid(conn, m.calculate_distance, complete=False)  # what IS in the tree is still a valid tree
    # the condition is persisted in the database: the marker table lists exactly the hashes that were left out,
    # a later manager that maintains the tree warns again, one that does not (the default) stays silent
    skipped = {r[0] for r in conn.execute("SELECT phash_id FROM hvd_vptree_skipped")}
    in_tree_ids = {r[0] for r in conn.execute("SELECT phash_id FROM shape_vptree")}
    all_ids = {r[0] for r in conn.execute("SELECT phash_id FROM shape_perceptual_hashes")}
    assert skipped == all_ids - in_tree_ids and len(skipped) == n - in_tree
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        later = hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=True)
        assert later.tree_incomplete and any("missing from shape_vptree" in str(x.message) for x in w2)
        n_w = len(w2)
        assert not hvd.vptree.VpTreeManager(conn, matcher=m).tree_incomplete and len(w2) == n_w
    # once the reference has rebuilt its tree (here: the rows appear in shape_vptree) the marker empties itself
    for pid in skipped:
        conn.execute("INSERT INTO shape_vptree ( phash_id, parent_id, radius, inner_id, inner_population, outer_id, outer_population ) "
                     "VALUES ( ?, -1, NULL, NULL, 0, NULL, 0 )", (pid,))
    assert not hvd.vptree.VpTreeManager(conn, matcher=m, maintain_reference_tree=True).tree_incomplete
    assert conn.execute("SELECT COUNT(*) FROM hvd_vptree_skipped").fetchone()[0] == 0
    conn.execute("DELETE FROM shape_vptree WHERE parent_id = -1")
    # the facade's search sees every file regardless
    tree = hvd.vptree.VpTreeManager(conn, matcher=m)
    res = tree.search_file(4, 101)  # radius 101 = "similarity below 1 %": every file, the capped-out ones included
    assert res[0] == (4, 0) and {h for h, _ in res} == set(range(1, n + 1))


def test_phash_map_cache_survives_rollbacks_and_foreign_commits(hvd, oracle, tmp_path):
    """the cached copy of shape_perceptual_hash_map was keyed on a CHANGE counter that a ROLLBACK takes
    back together with the change -- insert, look, roll back, insert a different row: same key, stale map -- and commits by
    another connection were only noticed when this connection had changed something too."""
    import sqlite3

    path = str(tmp_path / "videohashes.sqlite")
    conn = sqlite3.connect(path)
    for stmt in SCHEMA:
        conn.execute(stmt)
    blob = np.arange(64, dtype=np.uint8).tobytes()
    conn.execute("INSERT INTO shape_perceptual_hashes VALUES (1, ?)", (blob,))
    for h in (1, 2, 3):
        conn.execute("INSERT INTO files VALUES (?, ?)", (h, f"{h:064x}"))
    conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (1, 1)")
    conn.commit()
    tree = hvd.vptree.VpTreeManager(conn, matcher=OracleMatcher(oracle))
    assert tree.search_file(1, 0) == [(1, 0)]
    # same connection: insert, look, ROLLBACK, insert a different row
    conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (1, 2)")
    assert set(tree.search_file(1, 0)) == {(1, 0), (2, 0)}
    conn.rollback()
    conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (1, 3)")
    assert set(tree.search_file(1, 0)) == {(1, 0), (3, 0)}
    conn.commit()
    # another connection commits while this one changes nothing
    other = sqlite3.connect(path)
    other.execute("INSERT INTO shape_perceptual_hash_map VALUES (1, 2)")
    other.commit()
    assert set(tree.search_file(1, 0)) == {(1, 0), (2, 0), (3, 0)}
    conn.commit()  # (the look re-synchronised the TEMP state inside this connection's implicit transaction, which holds a
    #                read lock on the file until the caller's next commit -- dedup.py:491 commits after every file)
    other.execute("DELETE FROM shape_perceptual_hash_map WHERE hash_id = 3")
    other.commit()
    other.close()
    assert set(tree.search_file(1, 0)) == {(1, 0), (2, 0)}
    assert set(tree.search_file(1, 0)) == {(1, 0), (2, 0)}


def test_phash_map_token_sees_a_swap_and_survives_large_ids(hvd, oracle):
    """the cached copy of shape_perceptual_hash_map was keyed on a LINEAR row checksum -- two files that swap
    their perceptual hashes (the reference's DELETE + INSERT re-hash path) left count and sum unchanged -- and its SUM()
    overflowed SQLite's integers around 4 M phash ids, after which every search on the database failed."""
    import sqlite3

    rng = np.random.default_rng(12)
    conn = sqlite3.connect(":memory:")
    for stmt in SCHEMA:
        conn.execute(stmt)
    b1 = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
    b2 = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
    big = 4_000_000_000  # ids this large made SUM(phash_id * 1000003 + hash_id) overflow with a few rows
    conn.execute("INSERT INTO shape_perceptual_hashes VALUES (?, ?)", (big + 1, b1))
    conn.execute("INSERT INTO shape_perceptual_hashes VALUES (?, ?)", (big + 2, b2))
    for h in range(1, 5):
        conn.execute("INSERT INTO files VALUES (?, ?)", (big + 10 + h, f"{h:064x}"))
    conn.executemany("INSERT INTO shape_perceptual_hash_map VALUES (?, ?)",
                     [(big + 1, big + 11), (big + 2, big + 12), (big + 1, big + 13), (big + 2, big + 14)])
    conn.commit()
    tree = hvd.vptree.VpTreeManager(conn, matcher=OracleMatcher(oracle))
    assert set(tree.search_file(big + 11, 0)) == {(big + 11, 0), (big + 13, 0)}
    # files 11 and 12 swap their phashes: (p1,11),(p2,12) -> (p2,11),(p1,12): same row count, same LINEAR sum
    conn.execute("DELETE FROM shape_perceptual_hash_map WHERE hash_id IN (?, ?)", (big + 11, big + 12))
    conn.executemany("INSERT INTO shape_perceptual_hash_map VALUES (?, ?)", [(big + 2, big + 11), (big + 1, big + 12)])
    assert set(tree.search_file(big + 11, 0)) == {(big + 11, 0), (big + 14, 0)}
    assert set(tree.search_file(big + 12, 0)) == {(big + 12, 0), (big + 13, 0)}
    # the trigger-kept state and a resync from the table agree (same modulus on both sides)
    v, c, s = conn.execute("SELECT v, c, s FROM temp.hvd_amd_map_state").fetchone()
    for stmt in hvd.vptree.VpTreeManager._RESYNC_SQL:
        conn.execute(stmt)
    assert conn.execute("SELECT c, s FROM temp.hvd_amd_map_state").fetchone() == (c, s) and c == 4 and 0 <= s < 2 ** 61

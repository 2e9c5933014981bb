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

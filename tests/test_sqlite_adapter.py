"""f2: the adapter for the reference's on-disk database. The SQL logic, the incremental
`shape_search_cache` semantics and the file/phash fan-out are checked on CPU with the oracle
standing in for the GPU matcher (tests may use the oracle); the gpu-marked test runs the same
scenario through the real kernels."""
import sqlite3

import numpy as np
import pytest

SCHEMA = [  # reference db/DedupeDB.py:153-189
    "CREATE TABLE version (version TEXT)",
    "CREATE TABLE files ( hash_id INTEGER PRIMARY KEY, file_hash BLOB_BYTES UNIQUE )",
    "CREATE TABLE shape_perceptual_hashes ( phash_id INTEGER PRIMARY KEY, phash BLOB_BYTES UNIQUE )",
    "CREATE TABLE shape_perceptual_hash_map ( phash_id INTEGER, hash_id INTEGER, PRIMARY KEY ( phash_id, hash_id ) )",
    "CREATE TABLE shape_vptree ( phash_id INTEGER PRIMARY KEY, parent_id INTEGER, radius INTEGER, inner_id INTEGER, "
    "inner_population INTEGER, outer_id INTEGER, outer_population INTEGER )",
    "CREATE TABLE shape_maintenance_branch_regen ( phash_id INTEGER PRIMARY KEY )",
    "CREATE TABLE shape_search_cache ( hash_id INTEGER PRIMARY KEY, searched_distance INTEGER )",
    "CREATE TABLE phashed_file_queue ( file_hash BLOB_BYTES NOT NULL UNIQUE, phash BLOB_BYTES NOT NULL, "
    "PRIMARY KEY ( file_hash, phash ) )",
]


class OracleMatcher:
    """Stand-in for hvd_amd.search on CPU (test only)."""

    def __init__(self, oracle):
        self.o = oracle

    def match_videos(self, frames, offsets, max_dist):
        return self.o.match_videos(frames, offsets, max_dist)

    def calculate_distance(self, a, b):
        """db/vptree.py:29-31 with the oracle as matchHashBytes (comparator le, policy min: the repo's defaults)."""
        na, nb = len(a) // 32, len(b) // 32
        if na == 0 or nb == 0:
            return 101
        q, t = self.o.match_two(bytes(a), bytes(b), 31)
        return (100 - int(min(q * 100.0 / na, t * 100.0 / nb))) + 1

    def match_videos_cross(self, fq, oq, ft, ot, ids_q=None, ids_t=None, max_dist=31):
        from hvd_amd._lib import VMATCH_DTYPE

        out = []
        for a in range(len(oq) - 1):
            for b in range(len(ot) - 1):
                if ids_q is not None and ids_q[a] == ids_t[b]:
                    continue
                q, t = self.o.match_two(fq[oq[a]:oq[a + 1]].tobytes(), ft[ot[b]:ot[b + 1]].tobytes(), max_dist)
                if q or t:
                    out.append((a, b, q, t))
        return np.array(out, dtype=VMATCH_DTYPE)


def build_db(hvd, n_videos=60, seed=81):
    frames, offsets, planted = hvd.synth.video_hashes(n_videos, seed=seed, frames_per_video=12, copy_fraction=0.3)
    blobs = [frames[offsets[v]:offsets[v + 1]].tobytes() for v in range(n_videos)]
    blobs[7] = blobs[3]        # two files with the identical (non-empty if len>0) perceptual hash
    blobs[11] = b""            # empty hashes: never similar to anything, not even each other
    blobs[12] = b""
    conn = sqlite3.connect(":memory:")
    for stmt in SCHEMA:
        conn.execute(stmt)
    phash_id = {}
    for v, b in enumerate(blobs):
        hash_id = v + 1
        conn.execute("INSERT INTO files VALUES (?, ?)", (hash_id, f"{v:064x}"))
        if b not in phash_id:
            phash_id[b] = len(phash_id) + 1
            conn.execute("INSERT INTO shape_perceptual_hashes VALUES (?, ?)", (phash_id[b], b))
        conn.execute("INSERT INTO shape_perceptual_hash_map VALUES (?, ?)", (phash_id[b], hash_id))
        conn.execute("INSERT INTO shape_search_cache VALUES (?, NULL)", (hash_id,))
    conn.commit()
    return conn, blobs


def brute_force(oracle, blobs, threshold, pending):
    """The reference's predicate applied to every file pair (dedup.py:445-502, db/vptree.py:22-31)."""
    out = {}
    for a in range(len(blobs)):
        for b in range(a + 1, len(blobs)):
            if not (a in pending or b in pending):
                continue
            na, nb = len(blobs[a]) // 32, len(blobs[b]) // 32
            if na == 0 or nb == 0:
                continue
            q, t = oracle.match_two(blobs[a], blobs[b], 31)
            sim = min(q * 100.0 / na, t * 100.0 / nb)
            if int(sim) >= int(threshold):
                out[(f"{a:064x}", f"{b:064x}")] = sim
    return out


def check_scenario(hvd, oracle, matcher):
    from hvd_amd import sqlite_adapter as A

    conn, blobs = build_db(hvd)
    n = len(blobs)
    lib = A.load_library(conn)
    assert lib.hash_ids.size == n and lib.phash_ids.size == len(set(blobs))
    # first run: everything pending -> the full pair set
    pairs, ref_count = A.find_potential_duplicates(conn, 50.0, matcher=matcher)
    want = brute_force(oracle, blobs, 50.0, set(range(n)))
    assert {(a, b): s for a, b, s in pairs} == pytest.approx(want)
    assert ref_count == len(want) and len(want) >= 5
    if len(blobs[3]):
        assert (f"{3:064x}", f"{7:064x}") in want           # identical perceptual hash
    assert not any(f"{11:064x}" in k or f"{12:064x}" in k for k in want)  # empty hashes never match
    thr = hvd.fix_vpdq_similarity(50.0)
    assert conn.execute("SELECT COUNT(*) FROM shape_search_cache WHERE searched_distance = ?", (thr,)).fetchone()[0] == n
    # second run: nothing pending
    assert A.find_potential_duplicates(conn, 50.0, matcher=matcher) == ([], 0)
    # new files arrive (reference: REPLACE INTO shape_search_cache ... NULL, db/DedupeDB.py:318-324)
    new = {3, 20, 41}
    for v in new:
        conn.execute("UPDATE shape_search_cache SET searched_distance = NULL WHERE hash_id = ?", (v + 1,))
    pairs, ref_count = A.find_potential_duplicates(conn, 50.0, matcher=matcher)
    want_inc = brute_force(oracle, blobs, 50.0, new)
    assert {(a, b): s for a, b, s in pairs} == pytest.approx(want_inc)
    assert set(want_inc) <= set(want)
    # a stricter run later (higher threshold => smaller search distance) needs no new search;
    # a looser one (lower threshold) does, for every file
    assert A.find_potential_duplicates(conn, 75.0, matcher=matcher) == ([], 0)
    pairs, _ = A.find_potential_duplicates(conn, 30.0, matcher=matcher, update_cache=False)
    assert {(a, b): s for a, b, s in pairs} == pytest.approx(brute_force(oracle, blobs, 30.0, set(range(n))))


def test_sqlite_adapter_with_oracle_matcher(hvd, oracle):
    check_scenario(hvd, oracle, OracleMatcher(oracle))


@pytest.mark.gpu
def test_sqlite_adapter_on_gpu(gpu, hvd, oracle):
    check_scenario(hvd, oracle, None)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 300, 7), (40, 400, 8), (1100, 1300, 9)])
def test_match_videos_cross_vs_oracle(gpu, hvd, oracle, shape):
    """f3: query set x target set (new videos against the library) equals the per-pair matcher."""
    vq, vt, seed = shape
    ft, ot, _ = hvd.synth.video_hashes(vt, seed=seed, frames_per_video=(0, 12), copy_fraction=0.1)
    rng = np.random.default_rng(seed)
    pick = rng.choice(vt, size=min(vq, vt), replace=False)
    # queries: copies of some targets (exact, so they match themselves unless excluded) + fresh ones
    blobs = [ft[ot[p]:ot[p + 1]] for p in pick]
    fq = np.concatenate(blobs) if sum(len(b) for b in blobs) else np.zeros((0, 32), np.uint8)
    oq = np.zeros(len(blobs) + 1, np.int64)
    np.cumsum([len(b) for b in blobs], out=oq[1:])
    m = OracleMatcher(oracle)
    got = hvd.search.match_videos_cross(fq, oq, ft, ot)
    want = m.match_videos_cross(fq, oq, ft, ot)
    assert np.array_equal(got, want)
    ids_q, ids_t = pick.astype(np.int32), np.arange(vt, dtype=np.int32)
    got = hvd.search.match_videos_cross(fq, oq, ft, ot, ids_q, ids_t)
    want = m.match_videos_cross(fq, oq, ft, ot, ids_q, ids_t)
    assert np.array_equal(got, want)
    assert not any(ids_q[r["a"]] == r["b"] for r in got)

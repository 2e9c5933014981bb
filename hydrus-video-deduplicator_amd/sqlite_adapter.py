"""Run the GPU search directly on the reference's on-disk database (`videohashes.sqlite`).

Schema (reference db/DedupeDB.py:153-189):
    files(hash_id PK, file_hash UNIQUE)
    shape_perceptual_hashes(phash_id PK, phash BLOB UNIQUE)        -- N x 32 bytes, db/DedupeDB.py:535-559
    shape_perceptual_hash_map(phash_id, hash_id)                   -- one phash per file; files may share one
    shape_search_cache(hash_id PK, searched_distance)              -- NULL / < threshold => still to search
    phashed_file_queue(file_hash, phash)                           -- hashed, not yet inserted: ingest_phashed_file_queue()

Databases written before 0.10.0 hold every perceptual hash as JSON text (one "hex,quality,frame" string per
frame, bytes in reverse order, low-quality frames still present): `convert_old_vpdq_to_new` /
`upgrade_old_phashes` apply the reference's own migration (db/DedupeDB.py:528-584) instead of failing.

This replaces HydrusVideoDeduplicator.find_potential_duplicates (dedup.py:445-502) without the
VP-tree: every file whose `searched_distance` is NULL or below the search threshold is searched
against the whole library (VpTreeManager.search_file semantics, db/vptree.py:865-902), pairs are
reported once, and the cache is updated exactly as the reference does (dedup.py:488-491), so a
later run of either implementation continues incrementally. Hydrus I/O (marking the pairs) stays
with the caller.
"""

from __future__ import annotations

import sqlite3
from dataclasses import dataclass

import numpy as np

from . import search

QUALITY_TOLERANCE = 31  # db/DedupeDB.py:550-553


def convert_old_vpdq_to_new(old_vpdq_phash_json) -> bytes:
    """A pre-0.10.0 perceptual hash (JSON list of "<64 hex>,<quality>,<frame number>") in today's format: the
    frames with quality >= 31 in order, each hash with its 32 bytes reversed (db/DedupeDB.py:535-559)."""
    import json

    if isinstance(old_vpdq_phash_json, (bytes, bytearray, memoryview)):
        old_vpdq_phash_json = bytes(old_vpdq_phash_json).decode("ascii")
    out = bytearray()
    for feature in json.loads(old_vpdq_phash_json):
        hex_hash, quality, _frame_number = feature.split(",")
        if int(quality) >= QUALITY_TOLERANCE:
            raw = bytes.fromhex(hex_hash)
            if len(raw) != 32:
                raise ValueError("old-format frame hash is not 256 bits")
            out += raw[::-1]
    return bytes(out)


def is_old_format(phash) -> bool:
    """Pre-0.10.0 JSON text instead of raw 32-byte frame hashes. The reference keys its migration on the DB version
    table (db/DedupeDB.py:528-584); rows are judged here by CONTENT so that a half-migrated or foreign file still
    reads correctly -- and independently of the length: a JSON blob whose byte count happens to be a multiple of 32
    (about 1 in 32 of them) is still JSON. The test: a str, or bytes that are '[' ... ']', pure ASCII, and parse
    as a JSON list of strings. Raw hash bytes pass that with probability < 2^-64 (every byte would have to be ASCII)."""
    if isinstance(phash, str):
        return True
    b = bytes(phash)
    if len(b) < 2 or b[:1] != b"[" or b[-1:] != b"]" or not b.isascii():
        return False
    import json

    try:
        parsed = json.loads(b.decode("ascii"))
    except ValueError:
        return False
    return isinstance(parsed, list) and all(isinstance(x, str) for x in parsed)


def upgrade_old_phashes(conn: sqlite3.Connection) -> int:
    """In-place migration of both tables that hold perceptual hashes (db/DedupeDB.py:561-582). -> rows converted."""
    n = 0
    for phash_id, phash in conn.execute("SELECT phash_id, phash FROM shape_perceptual_hashes").fetchall():
        if is_old_format(phash):
            conn.execute("REPLACE INTO shape_perceptual_hashes ( phash_id, phash ) VALUES ( ?, ? )",
                         (phash_id, convert_old_vpdq_to_new(phash)))
            n += 1
    for file_hash, phash in conn.execute("SELECT file_hash, phash FROM phashed_file_queue").fetchall():
        if is_old_format(phash):
            conn.execute("DELETE FROM phashed_file_queue WHERE file_hash = ?", (file_hash,))
            conn.execute("REPLACE INTO phashed_file_queue ( file_hash, phash ) VALUES ( ?, ? )",
                         (file_hash, convert_old_vpdq_to_new(phash)))
            n += 1
    conn.commit()
    return n


def ingest_phashed_file_queue(conn: sqlite3.Connection, tree=None) -> int:
    """Move the hashed-but-not-inserted files into the library tables, as the reference's
    process_phashed_file_queue does (dedup.py:396-432 with db/DedupeDB.py:241-324): file row, perceptual-hash row
    (shared by files with an identical hash), the file's single map row, a NULL search-cache row (= still to be
    searched), and the queue row is deleted. `tree`: a VpTreeManager facade to notify (add_leaf). -> files ingested."""
    rows = conn.execute("SELECT file_hash, phash FROM phashed_file_queue").fetchall()
    for file_hash, phash in rows:
        blob = convert_old_vpdq_to_new(phash) if is_old_format(phash) else bytes(phash)
        if len(blob) % 32:
            raise ValueError("queued phash length is not a multiple of 32")
        conn.execute("INSERT OR IGNORE INTO files ( file_hash ) VALUES ( ? )", (file_hash,))
        row = conn.execute("SELECT phash_id FROM shape_perceptual_hashes WHERE phash = ?", (blob,)).fetchone()
        if row is None:
            conn.execute("INSERT INTO shape_perceptual_hashes ( phash ) VALUES ( ? )", (blob,))
            row = conn.execute("SELECT phash_id FROM shape_perceptual_hashes WHERE phash = ?", (blob,)).fetchone()
        phash_id = int(row[0])
        hash_id = int(conn.execute("SELECT hash_id FROM files WHERE file_hash = ?", (file_hash,)).fetchone()[0])
        if tree is not None:
            tree.add_leaf(phash_id, blob)
        conn.execute("DELETE FROM shape_perceptual_hash_map WHERE hash_id = ?", (hash_id,))  # one phash per file
        conn.execute("INSERT INTO shape_perceptual_hash_map ( phash_id, hash_id ) VALUES ( ?, ? )", (phash_id, hash_id))
        conn.execute("REPLACE INTO shape_search_cache ( hash_id, searched_distance ) VALUES ( ?, NULL )", (hash_id,))
        conn.execute("DELETE FROM phashed_file_queue WHERE file_hash = ? AND phash = ?", (file_hash, phash))
    conn.commit()
    return len(rows)


@dataclass
class Library:
    hash_ids: np.ndarray        # int64[F]  files that have a perceptual hash, ascending
    file_hashes: list           # [F]       files.file_hash
    phash_of_file: np.ndarray   # int64[F]  index into the unique perceptual hashes
    phash_ids: np.ndarray       # int64[P]
    frames: np.ndarray          # uint8[sum,32] unique perceptual hashes, concatenated
    offsets: np.ndarray         # int64[P+1]

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets)


def load_library(conn: sqlite3.Connection) -> Library:
    rows = conn.execute(
        "SELECT m.hash_id, f.file_hash, m.phash_id FROM shape_perceptual_hash_map m "
        "JOIN files f ON f.hash_id = m.hash_id ORDER BY m.hash_id"
    ).fetchall()
    phash_rows = conn.execute(
        "SELECT phash_id, phash FROM shape_perceptual_hashes WHERE phash_id IN "
        "(SELECT DISTINCT phash_id FROM shape_perceptual_hash_map) ORDER BY phash_id"
    ).fetchall()
    phash_ids = np.array([r[0] for r in phash_rows], dtype=np.int64)
    index_of = {int(pid): k for k, pid in enumerate(phash_ids)}
    # a pre-0.10 database is read through the reference's own conversion (upgrade_old_phashes rewrites it in place)
    blobs = [convert_old_vpdq_to_new(r[1]) if is_old_format(r[1]) else bytes(r[1]) for r in phash_rows]
    for b in blobs:
        if len(b) % 32:
            raise ValueError("phash BLOB length is not a multiple of 32")
    offsets = np.zeros(len(blobs) + 1, dtype=np.int64)
    np.cumsum([len(b) // 32 for b in blobs], out=offsets[1:])
    frames = np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(-1, 32).copy()
    return Library(
        hash_ids=np.array([r[0] for r in rows], dtype=np.int64),
        file_hashes=[r[1] for r in rows],
        phash_of_file=np.array([index_of[int(r[2])] for r in rows], dtype=np.int64),
        phash_ids=phash_ids, frames=frames, offsets=offsets)


def pending_hash_ids(conn: sqlite3.Connection, search_threshold: int) -> set:
    """dedup.py:458-461."""
    rows = conn.execute(
        "SELECT hash_id FROM shape_search_cache WHERE searched_distance is NULL or searched_distance < :threshold",
        {"threshold": search_threshold}).fetchall()
    return {int(r[0]) for r in rows}


def find_potential_duplicates(conn: sqlite3.Connection, threshold: float = 50.0, policy: str | None = None,
                              update_cache: bool = True, matcher=None, ingest_queue: bool = True):
    """-> (pairs, reference_count). pairs: sorted list of (file_hash_a, file_hash_b, similarity) with
    hash_id_a < hash_id_b, every unordered pair once, restricted (like the reference) to pairs with at
    least one side still to be searched. reference_count mimics the reference's return value
    (`num_similar_pairs // 2`, dedup.py:502), which counts a pair found from both sides once and a
    pair found from one side only as a half. matcher: object with match_videos / match_videos_cross
    (default: the GPU entry points of hvd_amd.search)."""
    matcher = search if matcher is None else matcher
    search_threshold = search.fix_vpdq_similarity(threshold)
    assert search_threshold > 0
    if ingest_queue:  # the reference builds its tree from the queue right before searching (dedup.py:339-343)
        ingest_phashed_file_queue(conn)
    lib = load_library(conn)
    pending = pending_hash_ids(conn, search_threshold)
    F, P = lib.hash_ids.size, lib.phash_ids.size
    file_pending = np.array([int(h) in pending for h in lib.hash_ids], dtype=bool)
    if F == 0 or not file_pending.any():
        return [], 0
    files_of = [[] for _ in range(P)]
    for f, p in enumerate(lib.phash_of_file):
        files_of[int(p)].append(f)
    phash_pending = np.zeros(P, dtype=bool)
    phash_pending[lib.phash_of_file[file_pending]] = True
    lengths = lib.lengths

    # --- GPU: pairs of distinct perceptual hashes with at least one frame hit -----------------
    if phash_pending.all() or phash_pending.sum() * 2 > P:
        recs = matcher.match_videos(lib.frames, lib.offsets, search.vpdq.frame_max_dist(search.DISTANCE_TOLERANCE))
        a_idx, b_idx = recs["a"].astype(np.int64), recs["b"].astype(np.int64)
    else:
        q_sel = np.flatnonzero(phash_pending)
        q_off = np.zeros(q_sel.size + 1, dtype=np.int64)
        np.cumsum(lengths[q_sel], out=q_off[1:])
        q_frames = np.concatenate([lib.frames[lib.offsets[p]:lib.offsets[p + 1]] for p in q_sel]) if q_off[-1] else \
            np.zeros((0, 32), np.uint8)
        recs = matcher.match_videos_cross(q_frames, q_off, lib.frames, lib.offsets,
                                          ids_q=q_sel.astype(np.int32), ids_t=np.arange(P, dtype=np.int32),
                                          max_dist=search.vpdq.frame_max_dist(search.DISTANCE_TOLERANCE))
        a_idx, b_idx = q_sel[recs["a"].astype(np.int64)], recs["b"].astype(np.int64)
    na, nb = lengths[a_idx].astype(np.float64), lengths[b_idx].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        qp = np.where(na > 0, recs["q_hits"] * 100.0 / na, 0.0)
        tp = np.where(nb > 0, recs["t_hits"] * 100.0 / nb, 0.0)
    pol = search.vpdq.MATCH_POLICY if policy is None else policy
    sim = np.minimum(qp, tp) if pol == "min" else np.maximum(qp, tp)
    keep = sim.astype(np.int64) >= int(threshold)  # <=> fix_vpdq_similarity(sim) <= search_threshold

    phash_pairs = {}
    for a, b, s_ in zip(a_idx[keep], b_idx[keep], sim[keep]):
        key = (int(min(a, b)), int(max(a, b)))
        phash_pairs[key] = max(phash_pairs.get(key, 0.0), float(s_))  # the cross form can see a pair from both sides

    # --- expand to files ---------------------------------------------------------------------
    found = {}
    for (pa, pb), s_ in phash_pairs.items():
        for fa in files_of[pa]:
            for fb in files_of[pb]:
                if file_pending[fa] or file_pending[fb]:
                    found[(min(fa, fb), max(fa, fb))] = s_
    for p in range(P):  # files sharing one non-empty perceptual hash are 100 % similar
        if lengths[p] > 0 and len(files_of[p]) > 1:
            fs = files_of[p]
            for x in range(len(fs)):
                for y in range(x + 1, len(fs)):
                    if file_pending[fs[x]] or file_pending[fs[y]]:
                        found[(fs[x], fs[y])] = 100.0
    directed = sum(int(file_pending[a]) + int(file_pending[b]) for a, b in found)
    pairs = [(lib.file_hashes[a], lib.file_hashes[b], s_) for (a, b), s_ in sorted(found.items())]

    if update_cache:
        conn.executemany("UPDATE shape_search_cache SET searched_distance = ? WHERE hash_id = ?;",
                         [(search_threshold, int(h)) for h in lib.hash_ids[file_pending]])
        conn.commit()
    return pairs, directed // 2

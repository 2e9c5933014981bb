"""Drop-in for the reference's `db/vptree.py` -- same module-level names (`fix_vpdq_similarity`,
`calculate_distance`, `VpTreeManager`) and the same `VpTreeManager` methods the pipeline calls
(`dedup.py:440-491`, `db/DedupeDB.py:287-324`): `add_leaf`, `maintain_tree`, `maintenance_due`,
`regenerate_tree`, `search_perceptual_hashes`, `search_file`, `reset_search` -- with no tree behind them.

The reference answers `search_file(hash_id, d)` by walking a vantage-point tree stored in SQLite,
one `vpdq.matchHashBytes` call per visited node (`db/vptree.py:707-777`); with this package's
per-pair entry that is ~18 us x visited nodes per file. Here the answer comes from ONE brute-force
pass on the GPU, cached: the first search uploads every perceptual hash of the database and runs
the video-level all-pairs search (`hvd_vpdq_match_videos`, counters reduced on the device); hashes
added later through `add_leaf` are compared against the library by a rectangular pass
(`hvd_vpdq_match_videos_cross`) at the next search. A search is then a dictionary lookup plus the
phash -> files fan-out in SQL. The result is what the tree returns when it prunes nothing (the
vPDQ "distance" is not a metric, so the tree itself may miss pairs): every perceptual hash whose
`calculate_distance` to the query is <= the radius.

`db` is the reference's `DedupeDb` (anything with `.execute(sql, params)`), or a `sqlite3.Connection`.
"""

from __future__ import annotations

import numpy as np

from . import search, vpdq
from .search import calculate_distance, fix_vpdq_similarity  # noqa: F401  (same names as db/vptree.py:22-31)


def dedupe_list(xs):
    """Order-preserving de-duplication (db/vptree.py:107-123)."""
    seen, out = set(), []
    for x in xs:
        if x not in seen:
            seen.add(x)
            out.append(x)
    return out


class VpTreeManager:
    def __init__(self, db, matcher=None):
        self.db = db
        self._matcher = search if matcher is None else matcher  # tests inject a CPU stand-in
        self._index = {}          # phash_id -> position
        self._phash_ids = []      # position -> phash_id
        self._blobs = []          # position -> bytes
        self._neigh = []          # position -> {position: (dist as query, dist as target)}
        self._searched = 0        # positions < _searched have been compared with everything before them
        self._loaded = False

    # ---- the parts of the reference API that maintained the tree: cheap or no-ops here ------------------------
    def add_leaf(self, perceptual_hash_id, perceptual_hash):
        """A new perceptual hash joins the library (db/vptree.py:155-283): appended; compared at the next search.
        The reference creates a fresh manager per inserted file (db/DedupeDB.py:303-304) and inserts the hash row before
        calling this, so an instance that has not read the library yet has nothing to do: its first search reads the row."""
        if not self._loaded:
            return
        if perceptual_hash_id in self._index:
            return
        self._append(int(perceptual_hash_id), bytes(perceptual_hash))

    def maintain_tree(self):
        """Nothing to rebalance (db/vptree.py:624-662)."""

    def regenerate_tree(self):
        """Forget the cached pass; the next search rebuilds it from the database (db/vptree.py:285-313)."""
        self.__init__(self.db, self._matcher)

    def maintenance_due(self, search_distance: int) -> bool:
        return False

    def reset_search(self, hash_ids):
        """Clear the search cache for the given hash ids (db/vptree.py:916-923)."""
        for hash_id in hash_ids:
            self.db.execute("UPDATE shape_search_cache SET searched_distance = NULL WHERE hash_id = :hash_id;",
                            {"hash_id": hash_id})

    # ---- library + cached GPU pass ------------------------------------------------------------------------------
    def _append(self, phash_id: int, blob: bytes) -> None:
        if len(blob) % vpdq.BYTES_PER_PDQ_HASH:
            raise ValueError("phash BLOB length is not a multiple of 32")
        self._index[phash_id] = len(self._phash_ids)
        self._phash_ids.append(phash_id)
        self._blobs.append(blob)
        self._neigh.append({})

    def _load(self) -> None:
        if self._loaded:
            return
        self._loaded = True
        for phash_id, blob in self.db.execute("SELECT phash_id, phash FROM shape_perceptual_hashes ORDER BY phash_id").fetchall():
            self._append(int(phash_id), bytes(blob))

    def _csr(self, lo: int, hi: int):
        lens = np.array([len(b) // 32 for b in self._blobs[lo:hi]], dtype=np.int64)
        off = np.zeros(lens.size + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        data = b"".join(self._blobs[lo:hi])
        return np.frombuffer(data, dtype=np.uint8).reshape(-1, 32), off, lens

    def _record(self, a: int, b: int, q_hits: int, t_hits: int) -> None:
        """One video-level record -> the two directed distances (query a / target b and the converse)."""
        na, nb = len(self._blobs[a]) // 32, len(self._blobs[b]) // 32
        d_ab = fix_vpdq_similarity(vpdq.percent_from_hits(q_hits, t_hits, na, nb))
        d_ba = fix_vpdq_similarity(vpdq.percent_from_hits(t_hits, q_hits, nb, na))
        self._neigh[a][b] = (d_ab, d_ba)
        self._neigh[b][a] = (d_ba, d_ab)

    def _refresh(self) -> None:
        self._load()
        n = len(self._blobs)
        if self._searched == n:
            return
        max_dist = vpdq.frame_max_dist(search.DISTANCE_TOLERANCE)
        if self._searched == 0 or (n - self._searched) * 2 > n:
            for d in self._neigh:
                d.clear()
            frames, off, _ = self._csr(0, n)
            for r in self._matcher.match_videos(frames, off, max_dist):
                self._record(int(r["a"]), int(r["b"]), int(r["q_hits"]), int(r["t_hits"]))
        else:  # the hashes added since the last pass against the whole library (themselves included)
            s = self._searched
            fq, oq, _ = self._csr(s, n)
            ft, ot, _ = self._csr(0, n)
            recs = self._matcher.match_videos_cross(fq, oq, ft, ot, ids_q=np.arange(s, n, dtype=np.int32),
                                                    ids_t=np.arange(n, dtype=np.int32), max_dist=max_dist)
            for r in recs:
                self._record(s + int(r["a"]), int(r["b"]), int(r["q_hits"]), int(r["t_hits"]))
        self._searched = n

    def _similar_positions(self, pos: int, radius: int):
        """[(position, distance)] of every library hash within `radius` of library hash `pos` as the query --
        itself included (a non-empty hash matches itself 100 % => distance 1, like the tree's own node)."""
        out = []
        if len(self._blobs[pos]) and 1 <= radius:
            out.append((pos, 1))
        for other, (d_q, _) in self._neigh[pos].items():
            if d_q <= radius:
                out.append((other, d_q))
        return out

    # ---- searches ---------------------------------------------------------------------------------------------------
    def _files_of(self, positions_and_distances):
        """phash -> files fan-out with the smallest distance per file (db/vptree.py:779-811)."""
        if not positions_and_distances:
            return []
        dist_of = {}
        for pos, dist in positions_and_distances:
            pid = self._phash_ids[pos]
            dist_of[pid] = min(dist, dist_of.get(pid, dist))
        ids = sorted(dist_of)
        best = {}
        for c0 in range(0, len(ids), 500):  # SQLite's default limit on bound parameters is 999
            chunk = ids[c0:c0 + 500]
            marks = ",".join("?" * len(chunk))
            for phash_id, hash_id in self.db.execute(
                    f"SELECT phash_id, hash_id FROM shape_perceptual_hash_map WHERE phash_id IN ({marks})", tuple(chunk)).fetchall():
                d = dist_of[int(phash_id)]
                if hash_id not in best or d < best[hash_id]:
                    best[hash_id] = d
        return list(best.items())

    def search_perceptual_hashes(self, search_perceptual_hashes, max_hamming_distance: int) -> list:
        """db/vptree.py:664-815: library hashes (what search_file passes) are answered from the cached pass, any other
        hash by one rectangular pass against the library."""
        out = []
        if len(search_perceptual_hashes) == 0:
            return out
        self._load()
        found, foreign = [], []
        for blob in search_perceptual_hashes:
            blob = bytes(blob)
            row = self.db.execute("SELECT phash_id FROM shape_perceptual_hashes WHERE phash = :phash;",
                                  {"phash": blob}).fetchone()
            if row is None:
                foreign.append(blob)  # not a library hash: compared against the library on the fly below
                continue
            pid = int(row[0])
            if pid not in self._index:
                self._append(pid, blob)
            found.append(self._index[pid])
        if max_hamming_distance == 0:  # identical perceptual hashes only
            return dedupe_list(self._files_of([(p, 0) for p in found]))
        self._refresh()
        hits = []
        for p in found:
            hits.extend(self._similar_positions(p, max_hamming_distance))
        foreign = [b for b in foreign if len(b)]
        if foreign and self._blobs:  # one rectangular pass: the foreign hashes as queries against the whole library
            if any(len(b) % vpdq.BYTES_PER_PDQ_HASH for b in foreign):
                raise ValueError("phash BLOB length is not a multiple of 32")
            lens = np.array([len(b) // 32 for b in foreign], dtype=np.int64)
            oq = np.zeros(lens.size + 1, dtype=np.int64)
            np.cumsum(lens, out=oq[1:])
            fq = np.frombuffer(b"".join(foreign), dtype=np.uint8).reshape(-1, 32)
            ft, ot, lt = self._csr(0, len(self._blobs))
            recs = self._matcher.match_videos_cross(fq, oq, ft, ot, max_dist=vpdq.frame_max_dist(search.DISTANCE_TOLERANCE))
            for r in recs:
                d = fix_vpdq_similarity(vpdq.percent_from_hits(int(r["q_hits"]), int(r["t_hits"]), int(lens[int(r["a"])]),
                                                               int(lt[int(r["b"])])))
                if d <= max_hamming_distance:
                    hits.append((int(r["b"]), d))
        return dedupe_list(self._files_of(hits))

    def search_file(self, hash_id: int, max_hamming_distance: int) -> list:
        """[(hash_id, distance)] of the files similar to `hash_id` within `max_hamming_distance` (a
        `fix_vpdq_similarity` distance, 1..101); the file itself leads the list at distance 0
        (db/vptree.py:865-902)."""
        similar = [(hash_id, 0)]
        if max_hamming_distance == 0:
            rows = self.db.execute(
                "SELECT hash_id FROM shape_perceptual_hash_map WHERE phash_id IN "
                "( SELECT phash_id FROM shape_perceptual_hash_map WHERE hash_id = ? );", (hash_id,)).fetchall()
            similar.extend((r[0], 0) for r in rows)
            return dedupe_list(similar)
        row = self.db.execute("SELECT phash_id FROM shape_perceptual_hash_map WHERE hash_id = :hash_id;",
                              {"hash_id": hash_id}).fetchone()
        assert row is not None
        phash_id = int(row[0])
        self._load()
        if phash_id not in self._index:  # inserted behind the facade's back: pick it up
            blob = self.db.execute("SELECT phash FROM shape_perceptual_hashes WHERE phash_id = ?", (phash_id,)).fetchone()
            assert blob is not None
            self._append(phash_id, bytes(blob[0]))
        self._refresh()
        similar.extend(self._files_of(self._similar_positions(self._index[phash_id], max_hamming_distance)))
        return dedupe_list(similar)

"""Drop-in for the reference's `db/vptree.py` -- same module-level names (`fix_vpdq_similarity`,
`calculate_distance`, `VpTreeManager`) and the same `VpTreeManager` methods the pipeline calls
(`dedup.py:440-491`, `db/DedupeDB.py:287-324`): `add_leaf`, `maintain_tree`, `maintenance_due`,
`regenerate_tree`, `search_perceptual_hashes`, `search_file`, `reset_search` -- answered without walking a tree.

The reference answers `search_file(hash_id, d)` by walking a vantage-point tree stored in SQLite,
one `vpdq.matchHashBytes` call per visited node (`db/vptree.py:707-777`); with this package's
per-pair entry that is ~18 us x visited nodes per file. Here the answer comes from ONE brute-force
pass on the GPU, cached: the first search uploads every perceptual hash of the database and runs
the video-level all-pairs search (`hvd_vpdq_match_videos`, counters reduced on the device); hashes
added later through `add_leaf` are compared against the library by a rectangular pass
(`hvd_vpdq_match_videos_cross`) at the next search. The result is what the tree returns when it prunes
nothing (the vPDQ "distance" is not a metric, so the tree itself may miss pairs): every perceptual hash
whose `calculate_distance` to the query is <= the radius.

Library scale (round 3): the pass's records are folded into directed neighbour lists with numpy (a CSR over the
library positions, no per-record Python), and the two SQL look-ups of a search -- file -> perceptual hash and
perceptual hash -> files (`db/vptree.py:779-811,887`) -- are answered from one in-memory copy of
`shape_perceptual_hash_map`, re-read only when that table changed (the connection's change counter first, then a
counter kept by connection-local TEMP triggers on that table plus SQLite's data_version: two O(1) statements).

Coexistence with the reference's tree (round 3): `add_leaf` keeps the on-disk `shape_vptree` VALID -- the new hash is
inserted as a leaf by the reference's own rule (walk from the root, inner iff distance <= radius, populations and the
rebalancing queue updated; `db/vptree.py:155-283`) -- so a user who goes back to the reference's `VpTreeManager` finds
every hash that was added while the facade was in charge. The facade itself never reads those rows.

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


def directed_distances(q_hits, t_hits, n_a, n_b, policy: str | None = None):
    """Vectorised `fix_vpdq_similarity(percent_from_hits(...))` for record arrays: (distance with a as the query and b
    as the target, distance with b as the query). Same float64 arithmetic as the scalar functions, element by element."""
    policy = vpdq.MATCH_POLICY if policy is None else policy
    q_hits = np.asarray(q_hits, dtype=np.float64)
    t_hits = np.asarray(t_hits, dtype=np.float64)
    n_a = np.asarray(n_a, dtype=np.float64)
    n_b = np.asarray(n_b, dtype=np.float64)
    ok = (n_a > 0) & (n_b > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        pa = np.where(ok, (q_hits * 100.0) / n_a, 0.0)  # share of a's frames with a partner in b
        pb = np.where(ok, (t_hits * 100.0) / n_b, 0.0)
    if policy == "min":
        s_ab = s_ba = np.minimum(pa, pb)
    elif policy == "max":
        s_ab = s_ba = np.maximum(pa, pb)
    elif policy == "query":
        s_ab, s_ba = pa, pb
    elif policy == "target":
        s_ab, s_ba = pb, pa
    else:
        raise ValueError(f"unknown match policy {policy!r}")
    fix = lambda s: (100 - s.astype(np.int64)) + 1  # noqa: E731  int() truncates, and so does astype on values >= 0
    return fix(s_ab), fix(s_ba)


class VpTreeManager:
    # add_leaf's upkeep of the reference's on-disk tree walks from the root to a free slot: one SELECT and one distance per
    # ancestor. Unrelated videos all sit at vPDQ distance 101, which the reference's rule ("inner iff distance <= radius",
    # first radius = the first child's distance) turns into a CHAIN until its own maintain_tree rebalances -- which the
    # facade never runs (its own search needs no tree). A walk is therefore cut off here (O(N) per ingested
    # file, 5e9 distance calls for 100 k files): a balanced tree of 10^6 hashes is ~20 levels deep, so a longer path means
    # the tree has degenerated; the leaf is then NOT inserted, `tree_incomplete` is set and a warning says what to do
    # before the reference's tree is used again (its --clear-search-tree / regenerate_tree rebuilds it from
    # shape_perceptual_hashes, where every hash still is).
    # The condition is PERSISTED: the skipped phash ids go into a marker table of the database, so that every
    # later manager on that file warns again -- and knows -- until the reference has rebuilt its tree (rows whose hash has
    # reached shape_vptree since are dropped at the next look).
    # Round 5: the upkeep of the reference's tree is OPT-IN. The tree is out of this build's scope
    #  and the facade's own search never reads it; a database that goes back to the reference's
    # tree runs the reference's --clear-search-tree once, which rebuilds it from shape_perceptual_hashes. Round 6 (ADVICE
    # r5): with the upkeep off, every hash that is not in an existing shape_vptree is recorded in hvd_vptree_skipped, so
    # that the incomplete tree is a detectable state of the database, not a silent one.
    MAX_TREE_WALK = 128
    SKIPPED_TABLE = "hvd_vptree_skipped"

    def __init__(self, db, matcher=None, maintain_reference_tree: bool = False):
        self.db = db
        self._matcher = search if matcher is None else matcher  # tests inject a CPU stand-in
        self._maintain_tree = maintain_reference_tree
        self.tree_incomplete = False  # a leaf was left out of shape_vptree because the walk exceeded MAX_TREE_WALK
        if maintain_reference_tree:
            self._check_skipped_marker()
        self._index = {}          # phash_id -> position
        self._phash_ids = []      # position -> phash_id
        self._blobs = []          # position -> bytes
        self._searched = 0        # positions < _searched have been compared with everything before them
        self._loaded = False
        # directed neighbour lists over positions (CSR): entries [_n_off[p], _n_off[p+1]) are (target position,
        # distance with p as the query)
        self._n_off = np.zeros(1, dtype=np.int64)
        self._n_dst = np.zeros(0, dtype=np.int64)
        self._n_dist = np.zeros(0, dtype=np.int64)
        # in-memory copy of shape_perceptual_hash_map
        self._map_token = None
        self._files_of_phash = {}   # phash_id -> [hash_id]
        self._phash_of_file = {}    # hash_id -> phash_id
        self._map_checked_at = None
        self._conn = None
        self._map_triggers = None   # None: not tried yet; True: TEMP triggers count the table's changes; False: fallback

    # ---- the parts of the reference API that maintained the tree ------------------------------------------------------
    def add_leaf(self, perceptual_hash_id, perceptual_hash):
        """A new perceptual hash joins the library (db/vptree.py:155-283). For the facade's own search it is appended and
        compared at the next search (the reference creates a fresh manager per inserted file, db/DedupeDB.py:303-304, and
        inserts the hash row before calling this, so an instance that has not read the library yet has nothing to remember:
        its first search reads the row). For the reference's tree it is inserted as a leaf, so that the tree stays whole."""
        perceptual_hash_id = int(perceptual_hash_id)
        perceptual_hash = bytes(perceptual_hash)
        if self._maintain_tree:
            self._insert_into_reference_tree(perceptual_hash_id, perceptual_hash)
        else:
            self._mark_left_out_of_reference_tree(perceptual_hash_id)
        if not self._loaded or perceptual_hash_id in self._index:
            return
        self._append(perceptual_hash_id, perceptual_hash)

    def _mark_left_out_of_reference_tree(self, phash_id: int) -> None:
        """Upkeep of the reference's tree is off (the default): where the database HAS a shape_vptree, record that this hash
        is not in it (the reference's search walks only shape_vptree, db/vptree.py:664-700, so a database written
        by this build and opened by the reference again would silently miss the hash). The marker table is what
        `_check_skipped_marker` reads: a later manager that maintains the tree warns and names the remedy
        (--clear-search-tree), and the marker empties itself once the rows are in the tree. Databases without the tree
        tables (never opened by the reference's tree code) are left alone."""
        try:
            if self.db.execute("SELECT 1 FROM shape_vptree WHERE phash_id = ?;", (phash_id,)).fetchone() is not None:
                return
            self.db.execute(f"CREATE TABLE IF NOT EXISTS {self.SKIPPED_TABLE} ( phash_id INTEGER PRIMARY KEY );")
            self.db.execute(f"INSERT OR IGNORE INTO {self.SKIPPED_TABLE} ( phash_id ) VALUES ( ? );", (phash_id,))
        except Exception as exc:  # noqa: BLE001 - sqlite3.OperationalError: no such table (no reference tree on this file)
            if "no such table" not in str(exc):
                raise

    def _check_skipped_marker(self) -> None:
        """Leaves an earlier manager left out of shape_vptree on this database: still missing -> warn again."""
        try:
            self.db.execute(f"DELETE FROM {self.SKIPPED_TABLE} WHERE phash_id IN ( SELECT phash_id FROM shape_vptree );")
            (n,) = self.db.execute(f"SELECT COUNT(*) FROM {self.SKIPPED_TABLE};").fetchone()
        except Exception as exc:  # noqa: BLE001 - no marker table (nothing was ever skipped) / no tree tables
            if "no such table" in str(exc):
                return
            raise
        if n:
            import warnings

            self.tree_incomplete = True
            warnings.warn(f"{n} perceptual hashes of this database are missing from shape_vptree (left out by an earlier "
                          "run: the tree had degenerated); run the reference's --clear-search-tree before using its tree "
                          "on this database", RuntimeWarning, stacklevel=3)

    def _distance(self, a: bytes, b: bytes) -> int:
        fn = getattr(self._matcher, "calculate_distance", None)
        return int(fn(a, b)) if fn is not None else int(calculate_distance(a, b))

    def _insert_into_reference_tree(self, phash_id: int, phash: bytes) -> None:
        """The reference's leaf insertion (db/vptree.py:155-283), restated: descend from the root -- inside a node's radius
        (or radius still NULL) goes to its inner child, outside to the outer one -- hang the leaf on the first free side
        (an inner leaf fixes the parent's radius to its distance), add one to the population of every ancestor on its side,
        and queue the eldest ancestor that has become lopsided (> 16 descendants, smaller/larger < 0.5) for the reference's
        own `maintain_tree`. A database without the tree tables is left alone."""
        try:
            root = self.db.execute("SELECT phash_id FROM shape_vptree WHERE parent_id IS NULL;").fetchone()
        except Exception as exc:  # noqa: BLE001 - sqlite3.OperationalError: no such table (trimmed test schemas)
            if "no such table" in str(exc):
                return
            raise
        if self.db.execute("SELECT 1 FROM shape_vptree WHERE phash_id = ?;", (phash_id,)).fetchone() is not None:
            # already a node: a second file with a known perceptual hash (db/DedupeDB.py:303-304 calls add_leaf for every
            # file). The reference walks again and REPLACEs the node, which resets its radius and children and orphans
            # its subtree; the hash is in the tree already, so the facade leaves it where it is.
            return
        parent_id = None
        inner_side, outer_side = [], []
        flagged = False
        node = root[0] if root is not None else None
        depth = 0
        while node is not None:
            depth += 1
            if depth > self.MAX_TREE_WALK:
                # nothing has been written for this leaf yet (slots and populations are updated below the loop)
                if not self.tree_incomplete:
                    import warnings

                    warnings.warn(
                        f"shape_vptree is more than {self.MAX_TREE_WALK} levels deep (degenerate until the reference's "
                        "maintain_tree runs): new hashes are no longer inserted into it; run the reference's "
                        "--clear-search-tree (regenerate_tree) before using its tree on this database again",
                        RuntimeWarning, stacklevel=3)
                self.tree_incomplete = True
                self.db.execute(f"CREATE TABLE IF NOT EXISTS {self.SKIPPED_TABLE} ( phash_id INTEGER PRIMARY KEY );")
                self.db.execute(f"INSERT OR IGNORE INTO {self.SKIPPED_TABLE} ( phash_id ) VALUES ( ? );", (phash_id,))
                return
            row = self.db.execute(
                "SELECT phash, radius, inner_id, inner_population, outer_id, outer_population FROM shape_perceptual_hashes "
                "NATURAL JOIN shape_vptree WHERE phash_id = ?;", (node,)).fetchone()
            if row is None:  # a hole in the tree (crash desync): hang the leaf on the ghost, as the reference does
                parent_id = node
                break
            a_phash, radius, inner_id, inner_pop, outer_id, outer_pop = row
            dist = self._distance(phash, bytes(a_phash))
            if radius is None or dist <= radius:
                inner_side.append(node)
                inner_pop += 1
                if inner_id is None:
                    self.db.execute("UPDATE shape_vptree SET inner_id = ?, radius = ? WHERE phash_id = ?;", (phash_id, dist, node))
                    parent_id = node
                nxt = inner_id
            else:
                outer_side.append(node)
                outer_pop += 1
                if outer_id is None:
                    self.db.execute("UPDATE shape_vptree SET outer_id = ? WHERE phash_id = ?;", (phash_id, node))
                    parent_id = node
                nxt = outer_id
            if not flagged and inner_pop + outer_pop > 16 and min(inner_pop, outer_pop) / max(inner_pop, outer_pop) < 0.5:
                self.db.execute("INSERT OR IGNORE INTO shape_maintenance_branch_regen ( phash_id ) VALUES ( ? );", (node,))
                flagged = True
            node = nxt
        for node in inner_side:
            self.db.execute("UPDATE shape_vptree SET inner_population = inner_population + 1 WHERE phash_id = ?;", (node,))
        for node in outer_side:
            self.db.execute("UPDATE shape_vptree SET outer_population = outer_population + 1 WHERE phash_id = ?;", (node,))
        self.db.execute(
            "INSERT OR REPLACE INTO shape_vptree ( phash_id, parent_id, radius, inner_id, inner_population, outer_id, "
            "outer_population ) VALUES ( ?, ?, NULL, NULL, 0, NULL, 0 );", (phash_id, parent_id))

    def maintain_tree(self):
        """Nothing of the facade's needs rebalancing (db/vptree.py:624-662). The rebalancing queue that `add_leaf` feeds
        is left for the reference's own `maintain_tree`, should its tree be used again: a lopsided tree is slow, not wrong."""

    def regenerate_tree(self):
        """Forget the cached pass; the next search rebuilds it from the database (db/vptree.py:285-313)."""
        self.__init__(self.db, self._matcher, self._maintain_tree)

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

    def _load(self) -> None:
        if self._loaded:
            return
        self._loaded = True
        for phash_id, blob in self.db.execute("SELECT phash_id, phash FROM shape_perceptual_hashes ORDER BY phash_id").fetchall():
            self._append(int(phash_id), bytes(blob))

    def _csr(self, lo: int, hi: int):
        lens = np.fromiter((len(b) // 32 for b in self._blobs[lo:hi]), dtype=np.int64, count=hi - lo)
        off = np.zeros(lens.size + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        data = b"".join(self._blobs[lo:hi])
        return np.frombuffer(data, dtype=np.uint8).reshape(-1, 32), off, lens

    def _fold(self, a, b, q_hits, t_hits, lens, keep_old: bool) -> None:
        """Video-level records (positions a, b; the kernel's two counters) -> directed neighbour lists, all in numpy."""
        a = np.asarray(a, dtype=np.int64)
        b = np.asarray(b, dtype=np.int64)
        d_ab, d_ba = directed_distances(q_hits, t_hits, lens[a], lens[b])
        src = np.concatenate([a, b])
        dst = np.concatenate([b, a])
        dist = np.concatenate([d_ab, d_ba])
        if keep_old and self._n_dst.size:
            old_src = np.repeat(np.arange(self._n_off.size - 1, dtype=np.int64), np.diff(self._n_off))
            src = np.concatenate([old_src, src])
            dst = np.concatenate([self._n_dst, dst])
            dist = np.concatenate([self._n_dist, dist])
        order = np.argsort(src, kind="stable")
        n = len(self._blobs)
        self._n_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=n), out=self._n_off[1:])
        self._n_dst = dst[order]
        self._n_dist = dist[order]

    def _refresh(self) -> None:
        self._load()
        n = len(self._blobs)
        if self._searched == n:
            return
        max_dist = vpdq.frame_max_dist(search.DISTANCE_TOLERANCE)
        if self._searched == 0 or (n - self._searched) * 2 > n:
            frames, off, lens = self._csr(0, n)
            r = self._matcher.match_videos(frames, off, max_dist)
            self._fold(r["a"], r["b"], r["q_hits"], r["t_hits"], lens, keep_old=False)
        else:  # the hashes added since the last pass against the whole library (themselves included)
            s = self._searched
            fq, oq, _ = self._csr(s, n)
            ft, ot, lens = self._csr(0, n)
            r = self._matcher.match_videos_cross(fq, oq, ft, ot, ids_q=np.arange(s, n, dtype=np.int32),
                                                 ids_t=np.arange(n, dtype=np.int32), max_dist=max_dist)
            # a pair of two NEW hashes comes back from both sides (a, b) and (b, a) with the counters swapped: keep one
            a = np.asarray(r["a"], dtype=np.int64) + s
            b = np.asarray(r["b"], dtype=np.int64)
            once = (b < s) | (a < b)
            self._fold(a[once], b[once], np.asarray(r["q_hits"])[once], np.asarray(r["t_hits"])[once], lens, keep_old=True)
        self._searched = n

    def _similar_positions(self, pos: int, radius: int):
        """[(position, distance)] of every library hash within `radius` of library hash `pos` as the query --
        itself included (a non-empty hash matches itself 100 % => distance 1, like the tree's own node)."""
        out = []
        if len(self._blobs[pos]) and 1 <= radius:
            out.append((pos, 1))
        lo, hi = int(self._n_off[pos]), int(self._n_off[pos + 1])
        near = {}
        if hi > lo:
            for other, d in zip(self._n_dst[lo:hi].tolist(), self._n_dist[lo:hi].tolist()):
                if d <= radius:
                    out.append((other, d))
                    near[other] = d
        if radius >= 101:  # threshold 0: "similarity below 1 %" = distance 101 is within the radius too, i.e. EVERY hash
            out.extend((other, 101) for other in range(len(self._blobs)) if other != pos and other not in near)
            if not len(self._blobs[pos]):
                out.append((pos, 101))  # an empty hash matches nothing, itself included (db/DedupeDB.py:555-557)
        return out

    # ---- phash <-> files, from one in-memory copy of shape_perceptual_hash_map ------------------------------------------
    def _connection(self):
        if self._conn is None:
            self._conn = getattr(self.db, "conn", None) or self.db
        return self._conn

    # What tells that shape_perceptual_hash_map changed (the first version counted CHANGES, and a counter that
    # is rolled back together with the change it counted reads the same after "insert, look, ROLLBACK, insert another row"):
    # the table's row count and a checksum of its rows, kept current by TEMP triggers -- they live in this connection's temp
    # schema, nothing is written to the user's file, and a ROLLBACK takes them back together with the rows they describe.
    # Commits by OTHER connections do not fire them; SQLite's data_version (O(1), read at every look) reports those, and
    # the two numbers are then taken from the table again.
    # Round 5: the checksum was the plain sum of phash_id * 1000003 + hash_id -- linear, so two files swapping
    # their phashes left it unchanged, and SUM() overflowed SQLite's integers around 4 M phash ids (every search on such a
    # database failed). Now a row's term is a PRODUCT of two bounded residues (< 2^47: a swap changes the sum), the state is
    # kept modulo 2^61 everywhere (triggers and resync agree), and a change counter v rides along: (v, c, s).
    _M = 2305843009213693952  # 2^61
    _ROW_TERM = "((({0}.phash_id * 1000003) % 2147483647) * ({0}.hash_id % 65521 + 1))"
    _VERSION_SQL = (
        "CREATE TEMP TABLE IF NOT EXISTS hvd_amd_map_state ( v INTEGER, c INTEGER, s INTEGER )",
        "CREATE TEMP TRIGGER IF NOT EXISTS hvd_amd_map_ins AFTER INSERT ON main.shape_perceptual_hash_map "
        "BEGIN UPDATE hvd_amd_map_state SET v = v + 1, c = c + 1, s = (s + " + _ROW_TERM.format("NEW") + f") % {_M}; END",
        "CREATE TEMP TRIGGER IF NOT EXISTS hvd_amd_map_del AFTER DELETE ON main.shape_perceptual_hash_map "
        "BEGIN UPDATE hvd_amd_map_state SET v = v + 1, c = c - 1, s = (s + " + f"{_M} - " + _ROW_TERM.format("OLD") + f") % {_M}; END",
        "CREATE TEMP TRIGGER IF NOT EXISTS hvd_amd_map_upd AFTER UPDATE ON main.shape_perceptual_hash_map "
        "BEGIN UPDATE hvd_amd_map_state SET v = v + 1, s = (s + " + f"{_M} - " + _ROW_TERM.format("OLD") + " + " + _ROW_TERM.format("NEW")
        + f") % {_M}; END",
    )
    # (the sum over the table in two halves of a term, 24 low bits and the rest, so that no partial sum can overflow 2^63 for
    # any table SQLite can hold; (hi << 24) mod 2^61 = (hi mod 2^37) << 24)
    _RESYNC_SQL = (
        "DELETE FROM temp.hvd_amd_map_state",
        "INSERT INTO temp.hvd_amd_map_state SELECT 0, COUNT(*), "
        "( COALESCE(SUM(" + _ROW_TERM.format("m") + " & 16777215), 0) % " + f"{_M}"
        " + ( ( COALESCE(SUM(" + _ROW_TERM.format("m") + " >> 24), 0) % 137438953472 ) << 24 ) ) % " + f"{_M} "
        "FROM main.shape_perceptual_hash_map AS m",
    )

    def _map_state(self, resync: bool):
        """(change counter, row count, row checksum) of shape_perceptual_hash_map from the trigger-kept state, or None where the triggers
        cannot be created (a read-only connection: nothing can change through it, data_version alone decides)."""
        if self._map_triggers is None:
            try:
                for stmt in self._VERSION_SQL:
                    self.db.execute(stmt)
                self._map_triggers = True
                resync = True
            except Exception:  # noqa: BLE001 - e.g. a read-only connection
                self._map_triggers = False
        if not self._map_triggers:
            return None
        if resync or self.db.execute("SELECT COUNT(*) FROM temp.hvd_amd_map_state").fetchone()[0] != 1:
            for stmt in self._RESYNC_SQL:
                self.db.execute(stmt)
        return tuple(self.db.execute("SELECT v, c, s FROM temp.hvd_amd_map_state").fetchone())

    def _map_fresh(self) -> None:
        """Make the in-memory copy of shape_perceptual_hash_map current. data_version is read at EVERY look (another
        connection's commit changes nothing on this one); nothing changed on this connection either (its own change
        counter, no SQL) -> done. Otherwise -- the search loop itself updates shape_search_cache after every file,
        dedup.py:488-491 -- one O(1) statement decides."""
        dv = self.db.execute("PRAGMA data_version").fetchone()[0]
        changes = getattr(self._connection(), "total_changes", None)
        if self._map_token is not None and dv == self._map_token[0] and changes is not None and changes == self._map_checked_at:
            return
        foreign_commit = self._map_token is None or dv != self._map_token[0]
        token = (dv, self._map_state(resync=foreign_commit))
        # (creating the triggers / re-synchronising the state counts as a change: read the counter after the statements)
        self._map_checked_at = getattr(self._connection(), "total_changes", None)
        if token == self._map_token and not foreign_commit:
            return
        files_of, phash_of = {}, {}
        for phash_id, hash_id in self.db.execute("SELECT phash_id, hash_id FROM shape_perceptual_hash_map ORDER BY phash_id, hash_id").fetchall():
            files_of.setdefault(phash_id, []).append(hash_id)
            phash_of[hash_id] = phash_id
        self._files_of_phash, self._phash_of_file, self._map_token = files_of, phash_of, token

    def _files_of(self, positions_and_distances, fresh: bool = False):
        """phash -> files fan-out with the smallest distance per file (db/vptree.py:779-811)."""
        if not positions_and_distances:
            return []
        if not fresh:
            self._map_fresh()
        ids = self._phash_ids
        if len(positions_and_distances) > 1:  # ascending perceptual-hash id, like the reference's temp-table join
            positions_and_distances = sorted(positions_and_distances, key=lambda pd: ids[pd[0]])
        best = {}
        files_of = self._files_of_phash
        for pos, dist in positions_and_distances:
            for hash_id in files_of.get(ids[pos], ()):
                if hash_id not in best or dist < best[hash_id]:
                    best[hash_id] = dist
        return list(best.items())

    # ---- searches ---------------------------------------------------------------------------------------------------
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
            if len(recs):
                ra, rb = np.asarray(recs["a"], dtype=np.int64), np.asarray(recs["b"], dtype=np.int64)
                d, _ = directed_distances(recs["q_hits"], recs["t_hits"], lens[ra], lt[rb])
                for p, dist in zip(rb.tolist(), d.tolist()):
                    if dist <= max_hamming_distance:
                        hits.append((p, dist))
        return dedupe_list(self._files_of(hits))

    def search_file(self, hash_id: int, max_hamming_distance: int) -> list:
        """[(hash_id, distance)] of the files similar to `hash_id` within `max_hamming_distance` (a
        `fix_vpdq_similarity` distance, 1..101); the file itself leads the list at distance 0
        (db/vptree.py:865-902)."""
        similar = [(hash_id, 0)]
        self._map_fresh()
        phash_id = self._phash_of_file.get(hash_id)
        if max_hamming_distance == 0:
            if phash_id is not None:
                similar.extend((h, 0) for h in self._files_of_phash.get(phash_id, ()))
            return dedupe_list(similar)
        assert phash_id is not None  # the reference asserts the same (db/vptree.py:887-892)
        phash_id = int(phash_id)
        self._load()
        if phash_id not in self._index:  # inserted behind the facade's back: pick it up
            blob = self.db.execute("SELECT phash FROM shape_perceptual_hashes WHERE phash_id = ?", (phash_id,)).fetchone()
            assert blob is not None
            self._append(phash_id, bytes(blob[0]))
        if self._searched != len(self._blobs):
            self._refresh()
            self._map_fresh()  # (the pass itself does not touch the map, but it may have taken a while)
        similar.extend(self._files_of(self._similar_positions(self._index[phash_id], max_hamming_distance), fresh=True))
        return dedupe_list(similar)

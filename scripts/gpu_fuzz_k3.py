"""Randomised parity sweep of the VIDEO-level search (K3: hvd_vpdq_match_videos through every all-pairs form, and the cross
search) against the CPU oracle (dev tool; tests/ hold the fixed cases). Libraries are ragged on purpose: empty videos, one-frame
videos, videos longer than a workgroup's 1024 rows, copies and partial copies, hashes whose first or second half is drawn from a few
prototypes (first-stage survivors everywhere: the pair queues and the tile route at work).
usage: python scripts/gpu_fuzz_k3.py [seeds=40] [first_seed=0]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, search, synth
from oracle import oracle as O

lib = L.init(0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
FORMS = [0, 9, 12, 18, 8, 13]
bad = 0
t0 = time.time()
for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    V = int(rng.choice([2, 3, 17, 60, 150, 400]))
    lens = rng.choice([0, 1, 2, 5, 33, 64, 64, 64, 130, 300, 1100], V, p=[.06, .06, .06, .1, .1, .15, .15, .1, .1, .08, .04])
    if lens.sum() > 30000:
        lens = np.minimum(lens, 130)
    off = np.zeros(V + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
    n = int(off[-1])
    kind = int(rng.integers(0, 4))
    fr = rng.integers(0, 256, (max(n, 1), 32), dtype=np.uint8)[:n]
    if n and kind == 1:    # lower halves from a few prototypes
        P = int(rng.choice([8, 64, 512])); fr[:, :16] = rng.integers(0, 256, (P, 16), dtype=np.uint8)[rng.integers(0, P, n)]
    elif n and kind == 2:  # upper halves from a few prototypes
        P = int(rng.choice([8, 64, 512])); fr[:, 16:] = rng.integers(0, 256, (P, 16), dtype=np.uint8)[rng.integers(0, P, n)]
    elif n and kind == 3:  # both, with a little noise: survivors in both halves, and real hits between unrelated videos
        P = int(rng.choice([64, 512]))
        fr[:, :16] = rng.integers(0, 256, (P, 16), dtype=np.uint8)[rng.integers(0, P, n)]
        fr[:, 16:] = rng.integers(0, 256, (P, 16), dtype=np.uint8)[rng.integers(0, P, n)]
        fr[:] = synth.flip_bits(fr, rng.integers(0, 30, n), rng)
    for _ in range(int(rng.integers(0, max(1, V // 3) + 1))):  # copies and partial copies around the tolerance
        a, b = rng.integers(0, V, 2)
        m = int(min(lens[a], lens[b]))
        if a == b or m == 0:
            continue
        k = int(rng.integers(1, m + 1))
        md = int(rng.choice([0, 10, 31, 31, 40]))
        fr[off[b]:off[b] + k] = synth.flip_bits(fr[off[a]:off[a] + k], np.clip(rng.integers(md - 3, md + 4, k), 0, 255), rng)
    tol = int(rng.choice([31, 31, 31, 0, 20, 63]))
    want = O.match_videos(fr, off, tol)
    L.check(lib.hvd_debug_set(b"mfma_force_sel", int(rng.choice([-1, 0, 1, 2]))))  # round 5: the first stage's 128 bits
    L.check(lib.hvd_debug_set(b"vmatch_bit_order", int(rng.choice([0, 2, 2]))))    # ... and the data-dependent bit order
    res = {}
    for v in FORMS:
        L.check(lib.hvd_debug_set(b"vmatch_variant", v))
        got = search.match_videos(fr, off, tol)
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH seed {seed}: V={V} n={n} kind={kind} tol={tol} form={v}: got {len(got)} want {len(want)}", flush=True)
    # cross search: a random subset of the videos as queries against all of them, ids exclude the query itself
    if V >= 3 and n:
        qs = np.sort(rng.choice(V, min(V, int(rng.integers(1, 12))), replace=False))
        qoff = np.zeros(qs.size + 1, dtype=np.int64); np.cumsum(lens[qs], out=qoff[1:])
        qfr = np.concatenate([fr[off[v]:off[v + 1]] for v in qs]) if qoff[-1] else np.zeros((0, 32), np.uint8)
        exp = []
        for qi, v in enumerate(qs):
            for t in range(V):
                if t == v or lens[v] == 0 or lens[t] == 0:
                    continue
                q, th = O.match_two(fr[off[v]:off[v + 1]].tobytes(), fr[off[t]:off[t + 1]].tobytes(), tol)
                if q or th:
                    exp.append((qi, t, q, th))
        for v in (0, 18, 12, 9):
            L.check(lib.hvd_debug_set(b"vmatch_variant", v))
            got = search.match_videos_cross(qfr, qoff, fr, off, ids_q=qs.astype(np.int32), ids_t=np.arange(V, dtype=np.int32), max_dist=tol)
            if got.tolist() != exp:
                bad += 1
                print(f"CROSS MISMATCH seed {seed}: V={V} n={n} kind={kind} tol={tol} form={v}: got {len(got)} want {len(exp)}", flush=True)
    L.check(lib.hvd_debug_set(b"vmatch_variant", 0))
    L.check(lib.hvd_debug_set(b"mfma_force_sel", -1))
    L.check(lib.hvd_debug_set(b"vmatch_bit_order", 1))
print(f"{nseeds} seeds from {first}: {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

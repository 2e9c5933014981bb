"""Randomised parity sweep of the all-pairs kernels against the CPU oracle (dev tool; tests/ hold the fixed cases).
usage: python scripts/gpu_fuzz_k2.py [seeds=120] [first_seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth, search
from oracle import oracle as O

lib = L.init(0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
bad = 0
slow = (0.0, None)
t_gpu = t_cpu = 0.0
for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([97, 1000, 1023, 1025, 4097, 9000, 20000, 33000]))
    kind = int(rng.integers(0, 5))
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if kind == 1:      # dense clusters (overflow the per-workgroup pair buffer)
        k = int(rng.integers(2, 6)); size = int(rng.integers(20, 90))
        for c in range(k):
            at = int(rng.integers(0, max(1, n - size)))
            base = rng.integers(0, 256, 32, dtype=np.uint8)
            db[at:at + size] = synth.flip_bits(np.tile(base, (min(size, n - at), 1)), rng.integers(0, 12, min(size, n - at)), rng)
    elif kind == 2:    # degenerate lower half
        db[:, :16] = synth.flip_bits(np.tile(rng.integers(0, 256, 32, dtype=np.uint8), (n, 1)), rng.integers(0, 10, n), rng)[:, :16]
    elif kind == 3:    # degenerate upper half
        db[:, 16:] = synth.flip_bits(np.tile(rng.integers(0, 256, 32, dtype=np.uint8), (n, 1)), rng.integers(0, 10, n), rng)[:, 16:]
    elif kind == 4:    # prototypes in both halves
        db[:, :16] = rng.integers(0, 256, (32, 16), dtype=np.uint8)[rng.integers(0, 32, n)]
        db[:, 16:] = rng.integers(0, 256, (32, 16), dtype=np.uint8)[rng.integers(0, 32, n)]
        db[:, 31] ^= rng.integers(0, 256, n, dtype=np.uint8)
    m = min(n // 3, 300)   # planted near-duplicates around the tolerance
    src = rng.choice(n, m, replace=False); dst = rng.choice(n, m, replace=False)
    md = int(rng.choice([0, 5, 31, 31, 31, 40, 63, 64, 100]))
    db[dst] = synth.flip_bits(db[src], np.clip(rng.integers(md - 3, md + 4, m), 0, 255), rng)
    group = (rng.integers(0, max(2, n // 7), n).astype(np.int32) if rng.random() < 0.4 else None)
    if group is not None:
        group.sort()
    variant = int(rng.choice([8, 9, 12, 13, 13, 13, 18, 18, 18]))
    tc = time.time()
    want = O.allpairs(db, md, group=group, cap=1 << 22, num_threads=8)
    t_cpu += time.time() - tc
    d_db = L.DeviceBuffer.from_array(db)
    d_img = M.expand_fp4(d_db.ptr, n)
    d_grp = L.DeviceBuffer.from_array(group) if group is not None else None
    cap = max(len(want) + 10, 16)
    d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8); d_cnt.zero()
    tg = time.time()
    sel = int(rng.choice([-1, -1, 0, 1, 2, 2]))  # round 5: which 128 bits the first stage sees -- the probe's choice or forced
    L.check(lib.hvd_debug_set(b"mfma_force_sel", sel))
    M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, d_grp.ptr if d_grp else None, md, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, variant)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    tg = time.time() - tg
    t_gpu += tg
    if tg > slow[0]:
        slow = (tg, f"seed {seed} n={n} kind={kind} md={md} variant={variant} sel={sel} pairs={cnt}")
    got = d_pairs.to_array(L.PAIR_DTYPE, min(cnt, cap))
    got = got[np.lexsort((got["j"], got["i"]))]
    ok = cnt == len(want) and np.array_equal(got, want)
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}: n={n} kind={kind} md={md} variant={variant} sel={sel} group={group is not None} got {cnt} want {len(want)}", flush=True)
    for b in (d_db, d_img, d_pairs, d_cnt, d_grp):
        if b is not None:
            b.free()
print(f"{nseeds} seeds from {first}: {bad} mismatches, {time.time() - t0:.0f} s (oracle {t_cpu:.0f} s, GPU launches {t_gpu:.1f} s; slowest launch {slow[0] * 1e3:.0f} ms: {slow[1]})")
sys.exit(1 if bad else 0)

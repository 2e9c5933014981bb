"""CPU tests: the C-ABI library loads and exports every declared symbol, fails loudly
without a GPU (no CPU fallback), and the host-side logic mirrors the reference's
semantics (VpdqHash value type, similarity policies, pair predicate)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols(headers=("hvd_mi355x.h", "hvd_mi355x_bench.h")):
    """Every function include/*.h declares: the drop-in boundary (hvd_mi355x.h) and the tests' / bench's own entry points
    (hvd_mi355x_bench.h: workload generator, test hook -- deliberately a separate header)."""
    syms = []
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        syms += re.findall(r"^\s*int\s+(hvd_\w+)\s*\(", text, flags=re.M)
    return sorted(set(syms))


def test_bench_symbols_are_not_in_the_product_header():
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["hvd_mi355x.h", "hvd_mi355x_bench.h"]
    product = declared_symbols(("hvd_mi355x.h",))
    assert "hvd_dev_synth_video_frames" not in product and "hvd_debug_parallel_copy" not in product
    assert declared_symbols(("hvd_mi355x_bench.h",)) == ["hvd_debug_parallel_copy", "hvd_dev_synth_video_frames"]


def test_header_symbols_all_exported_and_bound(hvd):
    from hvd_amd import _lib

    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/*.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    assert lib.hvd_abi_version() == 6


def test_no_gpu_means_loud_failure_not_fallback(hvd):
    from hvd_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path cannot be exercised here")
    with pytest.raises(_lib.HvdError) as e:
        _lib.init(0)
    assert e.value.code == _lib.HVD_ERR_NO_DEVICE
    with pytest.raises(_lib.HvdError):
        hvd.vpdq.hash_frames(np.zeros((1, 64, 64), np.uint8))
    with pytest.raises(_lib.HvdError):
        hvd.matchHashBytes(b"\0" * 32, b"\0" * 32, 31)
    with pytest.raises(_lib.HvdError):
        hvd.allpairs_hamming(np.zeros((4, 32), np.uint8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hydrus-video-deduplicator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                src = open(path).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "libhvd_oracle" not in src, f
            elif f.endswith((".cpp", ".hip", ".h", "Makefile")):
                src = open(path).read()
                assert "hvd_cpu_" not in src, f"{f} references an oracle symbol"
                assert not re.search(r"#include\s+[\"<][^\">]*oracle", src), f


def test_dct_matrix_host_copy_matches_oracle(hvd, oracle):
    import ctypes as C

    from hvd_amd import _lib

    d = np.zeros((16, 64), np.float32)
    _lib.check(_lib.load().hvd_dct_matrix(d.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(d.view(np.uint32), oracle.dct_matrix().view(np.uint32))
    # the compiled table is authoritative (hvd_init no longer consults libm); this host's libm must still agree with it
    m = np.zeros((16, 64), np.float32)
    _lib.check(_lib.load().hvd_dct_matrix_libm(m.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(d.view(np.uint32), m.view(np.uint32))


def test_vpdqhash_value_semantics(hvd):
    H = hvd.VpdqHash
    assert H.bytesPerPdqHash == 32
    a = H(bytes(range(32)) + bytes(range(32, 64)))
    assert len(a) == 2 and a.bytes == bytes(range(64))
    assert H.from_string(str(a)) == a and not (H.from_string(str(a)) != a)
    assert str(a) == bytes(range(64)).hex() and "\n" not in str(a)
    assert H.from_string(str(a) + "\n") == a  # tests read the text with file.read()/readline()
    assert len(H(b"")) == 0 and str(H(b"")) == ""
    assert a != H(bytes(64))
    with pytest.raises(ValueError):
        H(b"\0" * 31)
    with pytest.raises(ValueError):
        H.from_string("abc")
    with pytest.raises(ValueError):
        H.from_string("zz" * 32)
    assert hvd.hashing.decode_phash_from_str(hvd.hashing.encode_phash_to_str(a)) == a


def test_percent_policies(hvd):
    p = hvd.vpdq.percent_from_hits
    assert p(3, 1, 4, 2, "min") == 50.0
    assert p(3, 1, 4, 2, "max") == 75.0
    assert p(3, 1, 4, 2, "query") == 75.0
    assert p(3, 1, 4, 2, "target") == 50.0
    assert p(0, 0, 0, 5, "min") == 0.0 and p(0, 0, 5, 0, "max") == 0.0  # empty side => 0 (DedupeDB.py:555-557)
    assert p(64, 64, 64, 64) == 100.0
    with pytest.raises(ValueError):
        p(1, 1, 1, 1, "median")


def test_fix_vpdq_similarity_matches_reference_formula(hvd):
    f = hvd.fix_vpdq_similarity
    assert f(100.0) == 1 and f(0.0) == 101 and f(75.0) == 26 and f(74.99) == 27 and f(50) == 51


def test_pair_predicate_int_sim_ge_int_threshold(hvd):
    from hvd_amd._lib import VMATCH_DTYPE

    recs = np.array([(0, 1, 32, 32), (0, 2, 31, 64), (1, 2, 64, 31), (2, 3, 1, 1)], dtype=VMATCH_DTYPE)
    lengths = np.array([64, 64, 64, 64])
    # 32/64 = 50.0 -> kept at threshold 50; 31/64 = 48.4 -> int 48 < 50
    got = hvd.search.similar_video_pairs(recs, lengths, 50.0, "min")
    assert got.tolist() == [[0, 1]]
    got = hvd.search.similar_video_pairs(recs, lengths, 50.0, "max")
    assert got.tolist() == [[0, 1], [0, 2], [1, 2]]
    # truncation: threshold 48.9 -> int 48, and sim 48.4 -> int 48 => kept (reference: fix_vpdq_similarity)
    got = hvd.search.similar_video_pairs(recs, lengths, 48.9, "min")
    assert got.tolist() == [[0, 1], [0, 2], [1, 2]]
    with pytest.raises(ValueError):
        hvd.search.similar_video_pairs(recs, lengths, 0.5, "min")


@pytest.mark.parametrize("n,world", [(5000, 2), (20000, 3), (70000, 8), (1025, 4), (2, 2)])
def test_tile_ownership_partitions_the_upper_triangle(hvd, n, world):
    from hvd_amd import multigpu as M

    area = 0
    per_rank = []
    for r in range(world):
        a = 0
        for row0, row1, col0, col1 in M.tiles_of_rank(n, r, world):
            # pairs (i,j) of this tile with i<j: count analytically
            for_rows = np.arange(row0, row1)
            lo = np.maximum(col0, for_rows + 1)
            a += int(np.clip(col1 - lo, 0, None).sum())
        per_rank.append(a)
        area += a
    assert area == n * (n - 1) // 2, "tiles of all ranks must cover every i<j pair exactly once"
    if n >= 20000:
        assert max(per_rank) <= 1.25 * (area / world), f"imbalanced: {per_rank}"


def test_synth_is_deterministic(hvd):
    a, pa = hvd.synth.hash_db(2000, seed=9, plant_fraction=0.01)
    b, pb = hvd.synth.hash_db(2000, seed=9, plant_fraction=0.01)
    assert np.array_equal(a, b) and np.array_equal(pa, pb)
    f1 = hvd.synth.frames_gray(5, seed=4)
    f2 = hvd.synth.frames_gray(5, seed=4)
    assert np.array_equal(f1, f2) and f1.shape == (5, 64, 64)


def test_vpdqhash_text_forms_and_hash_semantics(hvd):
    H = hvd.VpdqHash
    raw = bytes(range(32)) * 3
    a = H(raw)
    assert H.from_string(raw.hex().upper()) == a          # hex case does not matter
    assert H.from_string("  " + raw.hex() + "\r\n") == a   # surrounding whitespace from file.read()
    assert a.frames().shape == (3, 32) and a.frames().tobytes() == raw
    assert len({a, H(raw), H(raw[:32])}) == 2             # usable as a dict/set key, value semantics
    assert H(bytearray(raw)) == a and H(memoryview(raw)) == a
    assert repr(a) == "VpdqHash(frames=3)"
    assert (a == raw) is False                            # only equal to another VpdqHash
    with pytest.raises(ValueError):
        H.from_string(raw.hex()[:-2])                     # not a whole number of frames


def test_similarity_of_records_direction_rules(hvd):
    """The reference discovers {A,B} from A's search or from B's (dedup.py:468-482): asymmetric
    policies therefore take the better direction, the symmetric one the minimum."""
    from hvd_amd._lib import VMATCH_DTYPE

    recs = np.array([(0, 1, 10, 2)], dtype=VMATCH_DTYPE)
    lengths = np.array([10, 20])
    f = hvd.search.similarity_of_records
    assert f(recs, lengths, "min")[0] == 10.0      # min(100 %, 10 %)
    assert f(recs, lengths, "query")[0] == 100.0   # A as query: 10/10; B as query: 2/20 -> best 100
    assert f(recs, lengths, "target")[0] == 100.0
    assert f(recs, lengths, "max")[0] == 100.0
    with pytest.raises(ValueError):
        f(recs, lengths, "mean")


def test_find_potential_duplicates_rejects_bad_blobs_before_touching_the_gpu(hvd):
    with pytest.raises(ValueError):
        hvd.find_potential_duplicates([b"\0" * 32, b"\0" * 31])


def test_compiled_dct_table_is_the_host_matrix(hvd, oracle):
    """The 64x64 hash kernel carries the DCT matrix as instruction literals (csrc/dct_table.inc, generated by
    scripts/gen_dct_table.py); it must be bit-identical to what the library's host code and the oracle compute."""
    import ctypes as C

    from hvd_amd import _lib

    txt = open(os.path.join(ROOT, "hydrus-video-deduplicator_amd", "csrc", "dct_table.inc")).read()
    baked = np.array([int(x, 16) for x in re.findall(r"0x([0-9A-F]{8})u", txt)], dtype=np.uint32)
    assert baked.size == 16 * 64
    host = np.zeros(16 * 64, dtype=np.float32)
    _lib.check(_lib.load().hvd_dct_matrix(host.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(host.view(np.uint32), baked)
    assert np.array_equal(oracle.dct_matrix().reshape(-1).view(np.uint32), baked)


# ---------------------------------------------------------------- round 4: the in-process device group -------------------
@pytest.mark.parametrize("variant", [9, 12, 13, 18])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_in_process_split_is_the_same_partition_for_every_form_the_probe_may_pick(hvd, world, variant):
    """The contexts of a device group take tile (rb, cb) by (rb + cb) % world like the ranks of the process-per-GPU mode
    (same kernel argument). The probe picks the form on the device, identically on every context (replicated DB); whatever
    it picks, the shares must tile the upper triangle exactly once -- including the round-4 pair-queue forms (15, and 18: the
    one the probe picks on frame hashes), whose workgroups cover the fetch form's 1024 rows."""
    from hvd_amd import multigpu as M

    n = 70_000
    assert M.tile_geometry(n, 13) == M.tile_geometry(n, 9) == M.tile_geometry(n, 18)
    area = 0
    for r in range(world):
        for row0, row1, col0, col1 in M.tiles_of_rank(n, r, world, variant=variant):
            rows = np.arange(row0, row1)
            area += int(np.clip(col1 - np.maximum(col0, rows + 1), 0, None).sum())
    assert area == n * (n - 1) // 2


def test_group_api_without_a_gpu_fails_loudly(hvd):
    import ctypes as C

    from hvd_amd import _lib

    lib = _lib.load()
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path cannot be exercised here")
    assert _lib.context_count() == 0 and _lib.group_exchange() == "none"
    devs = (C.c_int * 2)(0, 0)
    assert lib.hvd_init_devices(devs, 2) == _lib.HVD_ERR_NO_DEVICE
    assert lib.hvd_init_devices(devs, 0) == _lib.HVD_ERR_ARG and lib.hvd_init_devices(None, 2) == _lib.HVD_ERR_ARG
    assert lib.hvd_init_devices(devs, 17) == _lib.HVD_ERR_ARG
    assert lib.hvd_set_context(1) == _lib.HVD_ERR_ARG and lib.hvd_set_context(0) == _lib.HVD_OK
    assert _lib.context_count() == 0
    with pytest.raises(_lib.HvdError):
        _lib.init_devices([0, 0])


def test_hvd_devices_environment_is_parsed_before_any_device_is_touched(hvd, monkeypatch):
    from hvd_amd import _lib

    lib = _lib.load()
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible: hvd_init would bind it")
    monkeypatch.setenv("HVD_DEVICES", "0,x")
    assert lib.hvd_init(0) == _lib.HVD_ERR_ARG
    monkeypatch.setenv("HVD_DEVICES", "1,0")  # the list must start with the device the caller asks for
    assert lib.hvd_init(0) == _lib.HVD_ERR_ARG
    monkeypatch.setenv("HVD_DEVICES", "0,1")
    assert lib.hvd_init(0) == _lib.HVD_ERR_NO_DEVICE
    monkeypatch.delenv("HVD_DEVICES")
    assert lib.hvd_init(0) == _lib.HVD_ERR_NO_DEVICE


def test_thread_rendezvous_is_a_rendezvous():
    """The control plane of `bench.py --single-process` / the in-process ranks: same interface as the TCP rendezvous."""
    import threading

    from hvd_amd.rendezvous import ThreadRendezvous

    world = 3
    members = ThreadRendezvous.group(world)
    out = [None] * world

    def body(r):
        m = members[r]
        res = []
        for k in range(50):
            res.append(m.allgather(bytes([r, k])))
            res.append(m.allreduce_max([r, -r, k * r]))
            res.append(m.allreduce_min([r + k]))
            res.append(m.broadcast(b"x%d" % k if r == 1 else None, src=1))
            m.barrier()
        out[r] = res

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert out[0] is not None and out[0] == out[1] == out[2]
    assert out[0][0] == [b"\x00\x00", b"\x01\x00", b"\x02\x00"] and out[0][1] == [2.0, 0.0, 0.0] and out[0][3] == b"x0"


def test_library_load_defaults_the_ipc_mode_without_overriding_the_user():
    """Hosts whose driver only supports dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 for RCCL between processes; the library
    sets it when it is loaded (before the HSA runtime starts) unless the user chose a value."""
    import subprocess
    import sys

    code = ("import ctypes as C, os, sys; sys.path.insert(0, %r); import hvd_amd._lib as L; L.load(); "
            "g = C.CDLL(None).getenv; g.restype = C.c_char_p; print((g(b'HSA_ENABLE_IPC_MODE_LEGACY') or b'unset').decode())" % ROOT)
    for preset, want in ((None, "0"), ("1", "1")):
        env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
        if preset is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = preset
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        assert r.stdout.decode().strip().splitlines()[-1] == want


def test_unverified_policies_are_labelled_and_warned_about_once(hvd, monkeypatch):
    """the comparator / reduction defaults are guesses until hvdaccelerators can be consulted. Every
    artefact carries the labels (`policy_labels`), a default is marked `(unverified)`, a value chosen through the environment
    is not; the first matchHash* call with the comparator left at its default raises ONE RuntimeWarning per process."""
    import warnings

    from hvd_amd import vpdq

    monkeypatch.delenv("HVD_MATCH_COMPARATOR", raising=False)
    monkeypatch.delenv("HVD_MATCH_POLICY", raising=False)
    lab = vpdq.policy_labels()
    assert lab["comparator"] == f"{vpdq.MATCH_COMPARATOR} (unverified)" and lab["reduction"] == f"{vpdq.MATCH_POLICY} (unverified)"
    assert lab["dct"] in ("strict", "fma") and "unverified" in lab["hash_text"]
    monkeypatch.setattr(vpdq, "_warned_unverified", False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        vpdq.warn_unverified_policies()
        vpdq.warn_unverified_policies()
    assert len(rec) == 1 and issubclass(rec[0].category, RuntimeWarning) and "HVD_MATCH_COMPARATOR" in str(rec[0].message)
    # an explicit choice: no mark, no warning
    monkeypatch.setenv("HVD_MATCH_COMPARATOR", "lt")
    monkeypatch.setattr(vpdq, "_warned_unverified", False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        vpdq.warn_unverified_policies()
    assert not rec
    assert "(unverified)" not in vpdq.policy_labels()["comparator"]


@pytest.mark.parametrize("n,world", [(300_000, 2), (300_000, 8), (1_000_000, 8), (77_777, 3)])
def test_scaling_model_counts_every_step_of_the_pass_exactly_once(hvd, n, world):
    """hvd_amd.multigpu.rank_work_shares / predict_step: the per-rank work the prediction rests on is the
    kernel's own walk -- summed over the ranks it is the single-GPU walk, whatever the world size; the tile-cyclic deal is
    balanced to well under a per cent; the predicted curve is monotone and its N = 1 point is the measured constant."""
    M = hvd.multigpu
    one = M.rank_work_shares(n, 1)
    sh = M.rank_work_shares(n, world)
    assert sh.sum() == one.sum() and one.sum() > 0
    assert sh.max() / sh.mean() < 1.01
    assert M.rank_work_shares(n, world, tiles=True).sum() == M.rank_work_shares(n, 1, tiles=True).sum()
    p1, pw = M.predict_step(n, 1), M.predict_step(n, world)
    assert pw["kernel_ms"] < p1["kernel_ms"] and pw["kernel_ms"] * world > p1["kernel_ms"] * 0.99
    assert n < 300_000 or pw["ms_per_step"] < p1["ms_per_step"]  # (a tiny DB is all fixed cost: more GPUs add the exchange)
    weak = M.predict_scaling(mode="weak")
    strong = M.predict_scaling(mode="strong")
    assert [p["n_gpus"] for p in weak] == [1, 2, 4, 8] and weak[0]["efficiency"] == 1.0
    assert all(a["efficiency"] >= b["efficiency"] for a, b in zip(strong, strong[1:]))
    assert abs(weak[0]["kernel_ms"] - M.SCALING_MODEL["kernel_ms_per_1e11_cmp"] * 4.999995) < 0.01

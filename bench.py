#!/usr/bin/env python
"""bench.py -- headline benchmark of the VPDQ hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one brute-force all-pairs pass (256-bit Hamming, tolerance 31) over a synthetic
hash DB resident in HBM, including the candidate-pair exchange: BASELINE.json configs[2]
(1M hashes, ~5e11 comparisons) at N=1. For N>1 the DB grows so that the comparisons per GPU
stay fixed (weak scaling: n = 1M*sqrt(N)); the DB is replicated, tiles of the pair matrix
are dealt round-robin to the ranks, the only collective is the RCCL all-gather of each
rank's candidate pairs. The frame-hashing half of the metric (configs[1]: 10k pre-decoded
64x64 frames) is measured in the same run and reported under "frames_hashed".

Rank 0 prints ONE JSON line. The CPU baseline (the oracle, a port -- the reference's real
arithmetic is the absent hvdaccelerators wheel) is timed on rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP4_PEAK_TFLOPS = 10000.0     # dense FP4 MFMA peak (MI355X_MICROARCH.md; measured ceiling 9099)
BYTES_PER_COMPARISON = 64      # two 32-byte operands, no reuse credited
BYTES_PER_FRAME_64 = 4096 + 32 + 4


def host_threads() -> int:
    """Threads for the CPU baseline: os.cpu_count() bounded by the affinity mask and the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hashes", type=int, default=1_000_000, help="DB size at N=1")
    ap.add_argument("--frames", type=int, default=10_000)
    ap.add_argument("--variant", type=int, default=-1, help="all-pairs kernel variant (-1: product default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def main():
    # Native libraries (gloo, RCCL) print banners on fd 1; the contract is ONE JSON line on stdout.
    # Keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import hvd_amd
    from hvd_amd import _lib as L
    from hvd_amd import multigpu as M
    from hvd_amd import search, synth

    ndev = L.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one process per GPU: LOCAL_RANK picks the device; if the launcher masks devices per rank
    # (HIP_VISIBLE_DEVICES), every rank sees a single device 0. HVD_FORCE_DEVICE: dev testing only.
    dev = int(os.environ.get("HVD_FORCE_DEVICE", local_rank if local_rank < ndev else local_rank % ndev))
    lib = L.init(dev)

    dist = None
    exchange = None
    exchange_kind = "none"
    hard_exit = False  # a bootstrap thread stuck inside RCCL cannot be joined: leave with os._exit
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        boot = M.TorchDistExchange()
        uid = boot.broadcast_bytes(M.RcclExchange.create_unique_id() if rank == 0 else None, 128, src=0)
        # ncclCommInitRank is collective; bound it so that a hung bootstrap degrades to the gloo exchange
        # (reported in the JSON) instead of producing no measurement at all.
        import threading

        box = {}

        def _init():
            try:
                box["ex"] = M.RcclExchange(rank, world, uid)
            except Exception as exc:
                box["err"] = exc

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("HVD_RCCL_INIT_TIMEOUT", "120")))
        ok = 1 if ("ex" in box and not th.is_alive()) else 0
        import torch

        agree = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)  # all ranks use RCCL, or none does
        if int(agree.item()) == 1:
            exchange = box["ex"]
            exchange_kind = "rccl"
        else:
            why = "timed out" if th.is_alive() else repr(box.get("err", "failed on another rank"))
            print(f"[bench] rank {rank}: RCCL init {why}; exchanging candidates over gloo", file=sys.stderr)
            exchange = None
            exchange_kind = "gloo-fallback"
            hard_exit = th.is_alive()

    def barrier():
        # the kernels run on the library's own HIP stream, which hvd_dev_sync() drains; torch's device-wide
        # synchronize is added when torch is loaded anyway (N > 1) so that nothing of any stream is in flight
        L.check(lib.hvd_dev_sync())
        if dist is not None:
            import torch

            if torch.cuda.is_available():
                torch.cuda.synchronize(dev)  # this rank's GPU only
            dist.barrier()

    variant = search.DEFAULT_VARIANT if args.variant < 0 else args.variant

    # ---------------- workload: replicated synthetic hash DB ---------------------------
    n = int(round(args.hashes * math.sqrt(world) / 1024.0)) * 1024 if world > 1 else args.hashes
    db, planted = synth.hash_db(n, seed=3)
    d_db = L.DeviceBuffer.from_array(db)
    img_bytes = C.c_size_t(0)
    L.check(lib.hvd_fp4_image_bytes(n, C.byref(img_bytes)))
    d_img = L.DeviceBuffer(img_bytes.value)
    cap = 1 << 20
    d_pairs = L.DeviceBuffer(16 * cap)
    d_cnt = L.DeviceBuffer(8)
    total_cmp = n * (n - 1) // 2

    kernel_ms = []

    def step(v=variant, timed=True):
        d_cnt.zero()
        if v >= 8:  # the FP4 image is rebuilt inside every step: it is part of the pass, not a cached index
            L.check(lib.hvd_dev_expand_fp4(d_db.ptr, n, d_img.ptr))
        L.check(lib.hvd_timer_start())
        M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, rank, world, d_pairs.ptr, cap, d_cnt.ptr, v)
        ms = C.c_float(0)
        L.check(lib.hvd_timer_stop(C.byref(ms)))  # hipEvents on the library stream; also syncs
        if timed:
            kernel_ms.append(ms.value)
        cnt = int(d_cnt.to_array(np.uint64, 1)[0])
        if cnt > cap:
            raise RuntimeError("pair buffer overflow in bench")
        if world == 1:
            return d_pairs.to_array(L.PAIR_DTYPE, cnt)
        if exchange is not None:
            return exchange.allgather_pairs_dev(d_pairs.ptr, cnt)
        return boot.allgather_pairs(d_pairs.to_array(L.PAIR_DTYPE, cnt))

    for _ in range(args.warmup):
        step(timed=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_avg_ms = float(km.item())
    else:
        kernel_avg_ms = float(np.mean(kernel_ms))

    # parity gate that runs with every measurement: every planted pair within tolerance is
    # reported with its exact distance, and every reported pair verifies on the host
    merged = M.merge_pairs([recs])
    dist_host = np.unpackbits(db[merged["i"]] ^ db[merged["j"]], axis=1).sum(1)
    assert np.array_equal(dist_host, merged["dist"]) and (merged["dist"] <= 31).all()
    d_pl = np.unpackbits(db[planted[:, 0]] ^ db[planted[:, 1]], axis=1).sum(1)
    want = {(int(min(s, d)), int(max(s, d))) for (s, d, _), dd in zip(planted, d_pl) if dd <= 31}
    assert want <= set(zip(merged["i"].tolist(), merged["j"].tolist())), "planted duplicate pair missed"

    ms_per_step = elapsed / args.steps * 1e3
    value = total_cmp / (elapsed / args.steps)

    # ---------------- frames hashed / s (BASELINE configs[1]), every rank hashes its own batch ----
    fr = synth.frames_gray(args.frames, seed=2)
    d_f = L.DeviceBuffer.from_array(fr)
    d_h = L.DeviceBuffer(32 * args.frames)
    d_q = L.DeviceBuffer(4 * args.frames)
    reps = 20
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, args.frames, 64, 64, 1, None, d_h.ptr, d_q.ptr))
    barrier()
    t0 = time.perf_counter()
    L.check(lib.hvd_timer_start())
    for _ in range(reps):
        L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, args.frames, 64, 64, 1, None, d_h.ptr, d_q.ptr))
    ms = C.c_float(0)
    L.check(lib.hvd_timer_stop(C.byref(ms)))
    barrier()
    k1_wall = time.perf_counter() - t0
    k1_ms = ms.value / reps
    if dist is not None:
        import torch

        tw = torch.tensor([k1_wall, k1_ms], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        k1_wall, k1_ms = float(tw[0].item()), float(tw[1].item())

    if rank != 0:
        if exchange is not None:
            exchange.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if hard_exit:
            os._exit(0)
        return

    # kernel-level roofline (rank 0's share of the comparisons per launch)
    cmp_per_launch = total_cmp / world
    achieved = cmp_per_launch * BYTES_PER_COMPARISON / (kernel_avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"allpairs_n{n}_v{variant}_w{world}"
            traffic = tj.get(key)
        except Exception:
            traffic = None
    hbm_equiv = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(achieved / HBM_PEAK_GBS, 3),
                 "note": "SURVEY.md 8d accounting: 64 B per comparison with no operand reuse credited; frac > 1 "
                         "because tiles re-use operands from registers/LDS"}
    if variant >= 8:
        # executed matrix work: 2 (prefilter) or 4 MFMAs of 2*32*32*64 flop per 1024 comparisons
        flop_per_cmp = 256.0 if variant in (9, 11) else 512.0
        tfl = cmp_per_launch * flop_per_cmp / (kernel_avg_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": f"k_allpairs_mfma(variant={variant})", "achieved": round(tfl, 1),
                    "peak": FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / FP4_PEAK_TFLOPS, 3),
                    "traffic": traffic, "kernel_ms": round(kernel_avg_ms, 3),
                    "instr": "v_mfma_f32_32x32x64_f8f6f4 cbsz:4 blgp:4 on the +-1 FP4 image of the hashes",
                    "flop_per_comparison_executed": flop_per_cmp, "hbm_equivalent": hbm_equiv}
    else:
        roofline = {"bound": "hbm", "kernel": f"k_allpairs(variant={variant})", "traffic": traffic,
                    "kernel_ms": round(kernel_avg_ms, 3), **hbm_equiv,
                    "note": hbm_equiv["note"] + "; binding unit: integer VALU (v_bcnt_u32_b32 issues at half rate, "
                                                "profiles/r01_ubench_valu.txt)"}

    extra = {}
    # full-popcount variant next to the default, for transparency (same DB, same launch shape)
    if world == 1:
        for v, name in ((0, "popcount_full_16op"), (1, "popcount_prefilter128"), (8, "mfma_fp4_full"),
                        (9, "mfma_fp4_prefilter128")):
            ks = []
            for r in range(3):
                d_cnt.zero()
                L.check(lib.hvd_timer_start())
                M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
                ms = C.c_float(0)
                L.check(lib.hvd_timer_stop(C.byref(ms)))
                if r:
                    ks.append(ms.value)
            extra[name] = {"kernel_ms": round(float(np.mean(ks)), 3),
                           "comparisons_per_s": float(f"{total_cmp / (np.mean(ks) * 1e-3):.4g}")}

    fps = world * args.frames / (k1_ms * 1e-3)
    frames_out = {
        "workload": f"{args.frames} pre-decoded synthetic 64x64 gray frames per GPU -> PDQ hash + quality "
                    "(BASELINE configs[1]; frames are independent, ranks hash disjoint batches, no collective)",
        "value": float(f"{fps:.4g}"), "unit": "frames/s", "kernel_ms": round(k1_ms, 4), "dtype": "f32",
        "n_gpus": world, "wall_value": float(f"{world * args.frames * reps / k1_wall:.4g}"),
        "roofline": {"bound": "hbm", "kernel": "k_pdq_hash64", "achieved": round(fps / world * BYTES_PER_FRAME_64 / 1e9, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(fps / world * BYTES_PER_FRAME_64 / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "note": "fp32-VALU-bound at 64x64 (bit-exact non-FMA DCT: 2 VALU ops per MAC; PMC: SIMDs "
                             "issue-saturated, profiles/r01_pmc_k1.txt), not HBM-bound"},
    }
    cpu = None
    video_match = None
    if world == 1:
        # the reference's real frame geometry: 512x512 packed RGB24 (vpdqpy/vpdqpy.py:90-95)
        n_rgb = 6144  # two full rounds of the 3072 resident waves of k_down512w (4.8 GB of frames)
        rgb = synth.frames_rgb(16, seed=6)
        sb = C.c_size_t(0)
        L.check(lib.hvd_pdq_scratch_bytes(n_rgb, 512, 512, 3, C.byref(sb)))
        d_rf = L.DeviceBuffer(n_rgb * 786432)
        for rep in range(n_rgb // 16):  # the 16 distinct frames, replicated on the device
            L.check(lib.hvd_memcpy_h2d(C.c_void_p(d_rf.ptr + rep * rgb.nbytes), rgb.ctypes.data, rgb.nbytes))
        d_rs = L.DeviceBuffer(sb.value)
        d_rh = L.DeviceBuffer(32 * n_rgb)
        d_rq = L.DeviceBuffer(4 * n_rgb)
        rgb_ms = 1e9
        for r in range(9):  # best of 8 after a warm-up: the chip comes straight from the matrix-core workload
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_rf.ptr, n_rgb, 512, 512, 3, d_rs.ptr, d_rh.ptr, d_rq.ptr))
            ms = C.c_float(0)
            L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r:
                rgb_ms = min(rgb_ms, ms.value)
        rgb_fps = n_rgb / (rgb_ms * 1e-3)
        frames_out["rgb24_512x512"] = {
            "workload": f"{n_rgb} pre-decoded synthetic 512x512 RGB24 frames (the reference's hash_frame input): luma + "
                        "2x Jarosz + decimate (k_down512w, one wave per frame) + k_pdq_hash64",
            "value": float(f"{rgb_fps:.4g}"), "unit": "frames/s", "ms": round(rgb_ms, 3),
            "roofline": {"bound": "hbm", "achieved": round(rgb_fps * 786468 / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(rgb_fps * 786468 / 1e9 / HBM_PEAK_GBS, 3), "traffic": None,
                         "note": "algorithmic bytes = 786432 in + 36 out per frame. Memory-side traffic is 1.26 MB per frame "
                                 "(PMC, profiles/r01_pmc_down512w.txt: 192-byte runs re-fetch a shared 128-byte line, plus the "
                                 "pass-B state scratch), against a streaming ceiling of 6.2 TB/s for this access shape "
                                 "(profiles/r01_ubench_hbm_runs.txt) and an instruction floor of ~5e6 frames/s"}}
        if not args.no_cpu_baseline:
            from oracle import oracle as O  # cpu_baseline leg only

            cores = host_threads()
            # bounded sample of the same workload: calibrate, then ~cpu-seconds of work
            n0 = min(n, 150_000)  # big enough that thread start-up does not dominate on many-core hosts
            O.allpairs_count(db[:n0], 31, num_threads=cores)  # warm: page in, spawn once
            t = time.perf_counter()
            O.allpairs_count(db[:n0], 31, num_threads=cores)
            rate = (n0 * (n0 - 1) / 2) / (time.perf_counter() - t)
            ns = int(min(n, max(n0, math.sqrt(2 * rate * args.cpu_seconds))))
            t = time.perf_counter()
            O.allpairs_count(db[:ns], 31, num_threads=cores)
            dt = time.perf_counter() - t
            cpu_cmp = ns * (ns - 1) / 2 / dt
            n1 = min(n, 60_000)  # single-thread figure on a small prefix (~1-2 s)
            t = time.perf_counter()
            O.allpairs_count(db[:n1], 31, num_threads=1)
            cpu_cmp_1t = n1 * (n1 - 1) / 2 / (time.perf_counter() - t)
            t = time.perf_counter()
            ho, qo = O.hash_frames(fr, num_threads=cores)
            dtf = time.perf_counter() - t
            hg = d_h.to_array(np.uint8, 32 * args.frames).reshape(-1, 32)
            qg = d_q.to_array(np.int32, args.frames)
            assert np.array_equal(hg, ho) and np.array_equal(qg, qo), "GPU frame hashes differ from the oracle"
            t = time.perf_counter()
            hro, qro = O.hash_frames(np.concatenate([rgb] * 16), num_threads=cores)  # 256 frames
            dtr = time.perf_counter() - t
            hr = d_rh.to_array(np.uint8, 32 * n_rgb).reshape(-1, 32)
            qr = d_rq.to_array(np.int32, n_rgb)
            assert (np.array_equal(hr, np.tile(hro[:16], (n_rgb // 16, 1))) and
                    np.array_equal(qr, np.tile(qro[:16], n_rgb // 16))), "GPU rgb512 hashes differ from the oracle"
            frames_out["rgb24_512x512"]["cpu_frames_per_s"] = float(f"{256 / dtr:.4g}")
            # K3: 2000 videos x 64 frame hashes, every video pair; host buffers in, records out
            vfr, voff, _ = synth.video_hashes(2000, seed=7, frames_per_video=64, copy_fraction=0.02)
            search.match_videos(vfr[:6400], voff[:101])  # warm
            t = time.perf_counter()
            rec_g = search.match_videos(vfr, voff)
            dt_g = time.perf_counter() - t
            t = time.perf_counter()
            rec_c = O.match_videos(vfr, voff)
            dt_c = time.perf_counter() - t
            assert np.array_equal(rec_g, rec_c), "GPU video-match records differ from the oracle"
            k3_cmp = 2000 * 1999 // 2 * 4096
            video_match = {"workload": "2000 synthetic videos x 64 frame hashes, all video pairs (hvd_vpdq_match_videos, "
                                       "host buffers in, match records out)",
                           "value": float(f"{k3_cmp / dt_g:.4g}"), "unit": "frame comparisons/s", "ms": round(dt_g * 1e3, 2),
                           "records": int(len(rec_g)), "cpu_value": float(f"{k3_cmp / dt_c:.4g}"), "cpu_threads": 1,
                           "note": "small problem: transfer + launch overheads dominate the GPU figure (config 5, 50k "
                                   "videos, runs at the all-pairs kernel's rate: scripts/e2e_config5.py)"}
            cpu = {"value": float(f"{cpu_cmp:.4g}"), "unit": "comparisons/s", "cores": cores, "kind": "port",
                   "sample": f"oracle (C, popcnt, pthreads) all-pairs over the first {ns} of the {n} hashes "
                             f"({ns * (ns - 1) // 2:.3g} comparisons, {dt:.1f} s)",
                   "value_1thread": float(f"{cpu_cmp_1t:.4g}"),
                   "speedup_over_1thread": round(cpu_cmp / cpu_cmp_1t, 1), "os_cpu_count": os.cpu_count(),
                   "frames_per_s": float(f"{args.frames / dtf:.4g}"),
                   "frames_sample": f"oracle PDQ over the same {args.frames} frames, {cores} threads, {dtf:.2f} s",
                   "note": "the reference's real CPU path (hvdaccelerators 0.4.0) is not installable offline; this "
                           "is the oracle port"}

    out = {
        "metric": "hash-pair comparisons/sec", "value": float(f"{value:.5g}"), "unit": "comparisons/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp4(e2m1, +-1 image of the hash bits) x fp4 -> f32, exact" if variant >= 8 else "u32", "data": "synthetic",
        "config": {"workload": f"all-pairs 256-bit Hamming (tolerance 31) over {n} synthetic hashes with planted "
                               f"near-duplicates, {total_cmp:.6g} comparisons per step"
                               + (" (BASELINE configs[2])" if world == 1 and n == 1_000_000 else
                                  f" (configs[2] scaled weakly: n = 1M*sqrt({world}))"),
                   "n_hashes": n, "max_dist": 31, "kernel_variant": variant,
                   "parallelism": f"tile-cyclic x{world}, DB replicated, exchange={exchange_kind}",
                   "pairs_found": int(len(merged))},
        "roofline": roofline,
    }
    if extra:
        out["kernel_variants"] = extra
    if world == 1:
        # the reference-shaped loop -- one Python call per video pair, as db/vptree.py:29-31,737 issues them
        # (tests/benchmarks/test_benchmark_vpdqpy.py:62-73 has the same shape). 64-frame hashes, 1024 calls.
        vf, voff, _ = synth.video_hashes(33, seed=1, frames_per_video=64, copy_fraction=0.1)
        blobs = [vf[voff[v]:voff[v + 1]].tobytes() for v in range(33)]
        hvd_amd.calculate_distance(blobs[0], blobs[1])
        t = time.perf_counter()
        ncall = 0
        for a_ in range(32):
            for b_ in range(32):
                hvd_amd.calculate_distance(blobs[a_], blobs[b_ + 1])
                ncall += 1
        per_call = (time.perf_counter() - t) / ncall
        out["reference_shaped_loop"] = {
            "what": "calculate_distance(a, b) = fix_vpdq_similarity(matchHashBytes(a, b, 31)) on 64-frame video hashes, "
                    "one call per pair from Python (the reference's VP-tree call pattern)",
            "us_per_call": round(per_call * 1e6, 1), "calls": ncall,
            "frame_comparisons_per_s": float(f"{4096 / per_call:.3g}"),
            "note": "launch-bound by construction; the batch entry points above replace the loop, not the callee"}
    if frames_out:
        out["frames_hashed"] = frames_out
    if video_match:
        out["video_match"] = video_match
    if cpu:
        out["cpu_baseline"] = cpu
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()
    if exchange is not None:
        exchange.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if hard_exit:
        os._exit(0)


if __name__ == "__main__":
    main()

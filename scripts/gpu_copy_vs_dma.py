"""Dev probe: what bounds hash_frame(bytes) at 512x512 RGB24? (1) the parallel NT copy of 786 432-byte frames into page-locked
memory, alone; (2) the pinned H2D DMA alone; (3) both at once (copy into one pinned buffer while another is uploaded)."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L
lib = L.init(0)
FB, NF = 786432, 85
nb = FB * NF
hp = [C.c_void_p() for _ in range(2)]
for p in hp:
    L.check(lib.hvd_host_malloc(C.byref(p), nb)); C.memset(p, 1, nb)
d = L.DeviceBuffer(nb)
src = np.random.default_rng(1).integers(0, 256, (300, FB), dtype=np.uint8)  # 236 MB of source frames: DRAM-resident


def copy_pass(dst, threads, reps):
    t = time.perf_counter()
    for r in range(reps):
        for k in range(NF):
            lib.hvd_debug_parallel_copy(C.c_void_p(dst.value + k * FB), src[(r * NF + k) % 300].ctypes.data_as(C.c_void_p), FB, threads)
    return reps * nb / (time.perf_counter() - t) / 1e9


def dma_pass(srcp, reps):
    t = time.perf_counter()
    for _ in range(reps):
        L.check(lib.hvd_memcpy_h2d(d.ptr, srcp, nb))
    return reps * nb / (time.perf_counter() - t) / 1e9


for mode in (1, 0):
    L.check(lib.hvd_debug_set(b"copy_nt", mode))
    print("copy_nt", mode)
    for th in (1, 2, 4, 8):
        copy_pass(hp[0], th, 2)
        alone = copy_pass(hp[0], th, 10)
        box = {}
        stop = threading.Event()

        def bg():
            n, t = 0, time.perf_counter()
            while not stop.is_set():
                L.check(lib.hvd_memcpy_h2d(d.ptr, hp[1], nb)); n += 1
            box["dma"] = n * nb / (time.perf_counter() - t) / 1e9

        tb = threading.Thread(target=bg); tb.start()
        both = copy_pass(hp[0], th, 10)
        stop.set(); tb.join()
        print(f"  threads {th}: copy alone {alone:6.1f} GB/s | with DMA running: copy {both:6.1f} GB/s, DMA {box['dma']:5.1f} GB/s", flush=True)
print("DMA alone", round(dma_pass(hp[1], 20), 1), "GB/s (64 MiB copies)")

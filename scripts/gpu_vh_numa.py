"""Dev probe: does CPU placement bound hash_frame(bytes)? 512x512 RGB24 frames through VideoHasher with the process pinned to
(a) nothing, (b) the CPUs of the GPU's NUMA node, (c) the CPUs of the other node. Each case runs in a child process (first-touch
placement of the pinned ring and of the source frames follows the affinity in force when they are allocated)."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, ctypes as C
sys.path.insert(0, %r)
cpus = os.environ.get("HVD_TEST_CPUS")
if cpus:
    os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
import numpy as np
from hvd_amd import _lib as L, synth, vpdq
lib = L.init(0)
nb = 256 << 20
hp = C.c_void_p(); L.check(lib.hvd_host_malloc(C.byref(hp), nb)); C.memset(hp, 1, nb)
d = L.DeviceBuffer(nb); rates = []
for _ in range(6):
    t = time.perf_counter(); L.check(lib.hvd_memcpy_h2d(d.ptr, hp, nb)); rates.append(nb / (time.perf_counter() - t) / 1e9)
print("  h2d probe GB/s", round(float(np.median(rates[1:])), 1), flush=True)
rgb = synth.frames_rgb(16, seed=6)
video = np.ascontiguousarray(rgb[np.arange(300) %% 16])
hh, qq = vpdq.hash_frames(video)
want = hh[qq >= 31].tobytes()
frames = [video[k].tobytes() for k in range(300)]
for nt in (1, 2, 4, 8):
    for rep in range(2):
        t = time.perf_counter()
        for v in range(10):
            hs = vpdq.VideoHasher(1, 512, 512, nt)
            for f in frames:
                hs.hash_frame(f)
            assert hs.finish().bytes == want
        dt = time.perf_counter() - t
    print(f"  bytes num_threads {nt}: {3000 / dt:8.0f} frames/s {3000 * 786432 / dt / 1e9:6.2f} GB/s {dt / 3000 * 1e6:6.2f} us/frame", flush=True)
t = time.perf_counter()
for v in range(10):
    hs = vpdq.VideoHasher(1, 512, 512, 0)
    for k in range(300):
        hs.acquire_frame(3); hs.commit_frame()
    hs.finish()
dt = time.perf_counter() - t
print(f"  acquire_only: {3000 / dt:8.0f} frames/s {3000 * 786432 / dt / 1e9:6.2f} GB/s", flush=True)
''' % ROOT


def cpulist(path):
    out = set()
    for part in open(path).read().strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


sys.path.insert(0, ROOT)
from hvd_amd import _lib as L
info = L.runtime_info()
pci = info["devices"][0]["pci"].lower()
try:
    node = int(open(f"/sys/bus/pci/devices/{pci}/numa_node").read())
except OSError as e:
    node = -1
    print("no numa_node for", pci, e)
print("GPU", pci, "numa node", node, "affinity now:", len(os.sched_getaffinity(0)), "cpus")
nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
cases = [("no affinity", None)]
for nd in nodes:
    cases.append((f"node {nd}" + (" (the GPU's)" if nd == node else ""), cpulist(f"/sys/devices/system/node/node{nd}/cpulist")))
for name, cpus in cases:
    print(name, flush=True)
    env = dict(os.environ)
    if cpus:
        env["HVD_TEST_CPUS"] = ",".join(str(c) for c in sorted(cpus & os.sched_getaffinity(0)))
    subprocess.run([sys.executable, "-c", CHILD], env=env)

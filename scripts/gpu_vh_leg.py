"""Dev: bench.py's videohasher_stream leg on its own (twice)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
for _ in range(2):
    out = bench.videohasher_stream_leg(lib, L, synth, hvd_amd.vpdq)
    for g in ("512x512_rgb24", "64x64_gray"):
        print(g, {k: (v["us_per_frame"], v["h2d_frac"]) for k, v in out[g].items() if isinstance(v, dict)}, flush=True)

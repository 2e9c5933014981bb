"""GPU probe: the videohasher_stream leg of bench.py on its own (drop-in VideoHasher call pattern vs pinned H2D rate)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import hvd_amd  # noqa: E402
from hvd_amd import _lib as L  # noqa: E402
from hvd_amd import synth  # noqa: E402

lib = L.init(0)
print(json.dumps(bench.videohasher_stream_leg(lib, L, synth, hvd_amd.vpdq), indent=1))

"""MI355X-native VPDQ hashing + pairwise similarity: a drop-in for `hvdaccelerators.vpdq`
and the reference's hashing facade (hashing.py, vpdqpy/vpdqpy.py, db/vptree.py:22-31),
backed by hand-written gfx950 HIP kernels through a ctypes C-ABI (include/hvd_mi355x.h).

The directory name has hyphens; import it as ``hvd_amd`` (see /hvd_amd.py).
"""

from . import (_lib, hashing, multigpu, pipeline, rendezvous, search, sqlite_adapter, synth, vpdq,  # noqa: F401
               vpdqpy, vptree)
from .hashing import compute_phash, decode_phash_from_str, encode_phash_to_str, get_phash_similarity  # noqa: F401
from .search import (allpairs_hamming, calculate_distance, find_potential_duplicates,  # noqa: F401
                     fix_vpdq_similarity, match_videos)
from .vpdq import VideoHasher, VpdqHash, matchHash, matchHashBytes  # noqa: F401
from .pipeline import DeviceLibrary, dedupe_frames_on_device, dedupe_videos, hash_videos  # noqa: F401
from .vpdqpy import Vpdq  # noqa: F401

__version__ = "0.1.0"

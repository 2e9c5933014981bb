"""ctypes binding of libhvd_mi355x.so (C-ABI: include/hvd_mi355x.h).

This is the only place the shared library is loaded. There is no CPU fallback: if the
library is missing, or no gfx950 device is visible, the first compute call raises.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HVD_LIB_PATH") or os.path.join(_HERE, "libhvd_mi355x.so")  # override: A/B builds

HVD_OK = 0
HVD_ERR_ARG = -1
HVD_ERR_HIP = -2
HVD_ERR_OVERFLOW = -3
HVD_ERR_NO_DEVICE = -4
HVD_ERR_RCCL = -5
HVD_ERR_STATE = -6

PAIR_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("dist", "<u4"), ("pad", "<u4")])
VMATCH_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("q_hits", "<u4"), ("t_hits", "<u4")])

# name -> (restype, argtypes); every symbol include/hvd_mi355x.h declares.
_vp, _i64, _int, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
SIGNATURES = {
    "hvd_abi_version": (_int, []),
    "hvd_device_count": (_int, [C.POINTER(_int)]),
    "hvd_init": (_int, [_int]),
    "hvd_init_devices": (_int, [C.POINTER(C.c_int), _int]),
    "hvd_context_count": (_int, [C.POINTER(C.c_int)]),
    "hvd_set_context": (_int, [_int]),
    "hvd_get_context": (_int, []),
    "hvd_group_exchange": (_int, []),
    "hvd_group_abort": (_int, []),
    "hvd_group_rearm": (_int, []),
    "hvd_runtime_info": (_int, [C.c_char_p, _sz]),
    "hvd_shutdown": (_int, []),
    "hvd_last_error": (_int, [C.c_char_p, _sz]),
    "hvd_dct_matrix": (_int, [_vp]),
    "hvd_dct_matrix_libm": (_int, [_vp]),
    "hvd_pdq_hash_frames_gray_u8": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "hvd_pdq_hash_frames_rgb24_u8": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "hvd_allpairs_hamming256": (_int, [_vp, _i64, _vp, _int, _vp, _i64, C.POINTER(_i64)]),
    "hvd_match_two": (_int, [_vp, _i64, _vp, _i64, _int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hvd_vpdq_match_videos": (_int, [_vp, _vp, _i64, _int, _vp, _i64, C.POINTER(_i64)]),
    "hvd_vpdq_match_videos_cross": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _int, _vp, _i64,
                                           C.POINTER(_i64)]),
    "hvd_hasher_create": (_int, [_int, _int, _int, _i64, C.POINTER(_vp)]),
    "hvd_hasher_push": (_int, [_vp, _vp]),
    "hvd_hasher_set_threads": (_int, [_vp, _int]),
    "hvd_hasher_acquire": (_int, [_vp, C.POINTER(_vp)]),
    "hvd_hasher_commit": (_int, [_vp]),
    "hvd_hasher_acquire_n": (_int, [_vp, _i64, C.POINTER(_vp), C.POINTER(_i64)]),
    "hvd_hasher_commit_n": (_int, [_vp, _i64]),
    "hvd_hasher_pending": (_int, [_vp, C.POINTER(_i64)]),
    "hvd_hasher_finish": (_int, [_vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "hvd_hasher_destroy": (_int, [_vp]),
    "hvd_dev_malloc": (_int, [C.POINTER(_vp), _sz]),
    "hvd_dev_free": (_int, [_vp]),
    "hvd_host_malloc": (_int, [C.POINTER(_vp), _sz]),
    "hvd_host_free": (_int, [_vp]),
    "hvd_dev_memset": (_int, [_vp, _int, _sz]),
    "hvd_memcpy_h2d": (_int, [_vp, _vp, _sz]),
    "hvd_memcpy_d2h": (_int, [_vp, _vp, _sz]),
    "hvd_memcpy_d2d": (_int, [_vp, _vp, _sz]),
    "hvd_dev_sync": (_int, []),
    "hvd_device_synchronize": (_int, []),
    "hvd_set_pdq_dct_mode": (_int, [_int]),
    "hvd_get_pdq_dct_mode": (_int, []),
    "hvd_debug_set": (_int, [C.c_char_p, _int]),
    "hvd_debug_get": (_int, [C.c_char_p, C.POINTER(_int)]),
    "hvd_pdq_scratch_bytes": (_int, [_i64, _int, _int, _int, C.POINTER(_sz)]),
    "hvd_dev_pdq_hash_frames": (_int, [_vp, _i64, _int, _int, _int, _vp, _vp, _vp]),
    "hvd_dev_allpairs_hamming256": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _i64, _vp, _int]),
    "hvd_fp4_image_bytes": (_int, [_i64, C.POINTER(_sz)]),
    "hvd_dev_expand_fp4": (_int, [_vp, _i64, _vp]),
    "hvd_dev_allpairs_hamming256_mfma": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int, _vp, _i64, _vp, _int]),
    "hvd_dev_cross_hamming256_mfma": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _int, _int, _int, _vp, _i64, _vp]),
    "hvd_dev_video_of_frames": (_int, [_vp, _i64, _i64, _vp]),
    "hvd_dev_compact_kept": (_int, [_vp, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, C.POINTER(_i64)]),
    "hvd_dev_vpdq_match_videos": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _i64, _vp]),
    "hvd_dev_vpdq_emit_again": (_int, [_vp, _i64, _vp]),
    "hvd_dev_vpdq_match_videos_cross": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _int, _int, _int, _vp, _i64,
                                               _vp]),
    # include/hvd_mi355x_bench.h (tests / bench only, not part of the drop-in boundary)
    "hvd_dev_synth_video_frames": (_int, [_vp, _i64, _i64, _int, C.c_uint64, _vp]),
    "hvd_debug_parallel_copy": (_int, [_vp, _vp, C.c_size_t, _int]),
    "hvd_allpairs_tile_geometry": (_int, [_i64, _int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "hvd_timer_start": (_int, []),
    "hvd_timer_stop": (_int, [C.POINTER(C.c_float)]),
    "hvd_timer_mark": (_int, [_int]),
    "hvd_timer_between": (_int, [_int, _int, C.POINTER(C.c_float)]),
    "hvd_comm_unique_id": (_int, [_vp]),
    "hvd_comm_init": (_int, [_vp, _int, _int]),
    "hvd_comm_allgather_pairs": (_int, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "hvd_comm_allgather_bytes": (_int, [_vp, _vp, _sz]),
    "hvd_comm_destroy": (_int, []),
    "hvd_comm_abort": (_int, []),
}


class HvdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"hvd_mi355x error {code}: {message}")
        self.code = code


_lib = None
_lock = threading.Lock()
_inited_device = None


def load() -> C.CDLL:
    """dlopen the library and bind every symbol (no device needed for this)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HvdError(HVD_ERR_STATE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (make -C hydrus-video-deduplicator_amd/csrc). There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().hvd_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != HVD_OK:
        raise HvdError(rc, last_error())


def device_count() -> int:
    n = C.c_int(0)
    check(load().hvd_device_count(C.byref(n)))
    return n.value


def init(device: int | None = None) -> C.CDLL:
    """Bind this process to one GPU (default: $HVD_DEVICE, else $LOCAL_RANK, else the first of $HVD_DEVICES, else 0).
    With HVD_DEVICES=0,1,2,... the library turns this into a device GROUP (hvd_init_devices): the host-buffer entry points
    -- and with them the search, the VpTreeManager facade and the SQLite adapter -- then shard over all listed GPUs inside
    this one process."""
    global _inited_device
    lib = load()
    if device is None:
        first = os.environ.get("HVD_DEVICES", "0").split(",")[0].strip() or "0"
        device = int(os.environ.get("HVD_DEVICE", os.environ.get("LOCAL_RANK", first)))
    if _inited_device != device:
        check(lib.hvd_init(device))
        _inited_device = device
    return lib


def init_devices(devices) -> C.CDLL:
    """Bind this process to a group of GPUs, one context per listed device (include/hvd_mi355x.h: hvd_init_devices). A
    device may be listed twice (two contexts on one GPU; the exchange steps then go through host memory)."""
    global _inited_device
    lib = load()
    devs = [int(d) for d in devices]
    arr = (C.c_int * len(devs))(*devs)
    check(lib.hvd_init_devices(arr, len(devs)))
    _inited_device = devs[0]
    return lib


def context_count() -> int:
    n = C.c_int(0)
    check(load().hvd_context_count(C.byref(n)))
    return n.value


def set_context(index: int) -> None:
    """The calling THREAD's current context (and HIP device) from now on; device buffers belong to the context they were
    allocated on."""
    check(load().hvd_set_context(int(index)))


def group_exchange() -> str:
    return {0: "none", 1: "rccl", 2: "host"}[load().hvd_group_exchange()]


def group_abort() -> None:
    """Release the other contexts' threads from an exchange step this thread will never reach (hvd_group_abort)."""
    load().hvd_group_abort()


def group_rearm() -> None:
    """Put the group back to work after an abandoned exchange (hvd_group_rearm): host barrier re-armed, aborted RCCL
    communicators re-created. Only while no thread is inside a group call."""
    check(load().hvd_group_rearm())


def runtime_info() -> dict:
    """HIP / RCCL versions and library paths, the visible devices, the group's peer matrix (hvd_runtime_info)."""
    import json

    buf = C.create_string_buffer(1 << 16)
    check(load().hvd_runtime_info(buf, len(buf)))
    info = json.loads(buf.value.decode("utf-8", "replace"))
    from . import vpdq  # (the comparator / reduction policies live above the C-ABI, which takes an inclusive distance bound)

    info["policies"] = vpdq.policy_labels()
    return info


def ensure() -> C.CDLL:
    return init() if _inited_device is None else _lib


def shutdown() -> None:
    global _inited_device
    if _lib is not None and _inited_device is not None:
        import sys

        pl = sys.modules.get(__package__ + ".pipeline")
        if pl is not None:
            pl.release_record_buffers()  # (device memory of the old binding: must not be handed out after a re-init)
        check(_lib.hvd_shutdown())
        _inited_device = None


class DeviceBuffer:
    """Owning handle of an HBM allocation made through the C-ABI."""

    def __init__(self, nbytes: int):
        lib = ensure()
        p = C.c_void_p()
        check(lib.hvd_dev_malloc(C.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    @classmethod
    def from_array(cls, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        if arr.nbytes:
            check(_lib.hvd_memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def to_array(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            assert out.nbytes <= self.nbytes
            check(_lib.hvd_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self) -> None:
        check(_lib.hvd_dev_memset(self.ptr, 0, self.nbytes))

    def free(self) -> None:
        if self.ptr is not None and _lib is not None and _inited_device is not None:
            _lib.hvd_dev_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

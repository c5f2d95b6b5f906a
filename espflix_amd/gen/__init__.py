"""Synthetic MPEG-1 I/P stream generator (ctypes binding of espflix_amd/gen/libefx_gen.so).

Workload tooling for bench.py and the tests: deterministic 352x192 streams as specified in
SURVEY.md section 8(d).  Not part of the decode path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

FLAG_I_ONLY = 1
FLAG_CUSTOM_MATRICES = 2
FLAG_WIDE_SLICES = 4
FLAG_LONG_SKIPS = 8
FLAG_FLAT_BRIGHT = 16
FLAG_RATE_1500K = 32   # mean picture ~6.25 kB (1.5 Mbit/s at 30 Hz, the service's profile)
FLAG_HUGE_LEVELS = 64  # AC levels up to +-255 / -256, every escape form of player.cpp:1092-1099, coded zeros, runs > 31
FLAG_SLICE_EXTRA = 256  # 0-3 extra_bit_slice = 1 + information bytes in every slice header (player.cpp:1261-1262)
FLAG_ODD_HEADERS = 128  # B / D / reserved picture types on P-coded pictures, user_data / extension units, varying f_code

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libefx_gen.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make gen`")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.efxgen_batch_create.restype = vp
        L.efxgen_batch_create.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int]
        L.efxgen_batch_destroy.argtypes = [vp]
        L.efxgen_batch_es_size.restype = C.c_uint64
        L.efxgen_batch_es_size.argtypes = [vp, C.c_int]
        L.efxgen_batch_es_copy.restype = C.c_uint64
        L.efxgen_batch_es_copy.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.efxgen_batch_ts.restype = C.c_uint64
        L.efxgen_batch_ts.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.efxgen_fnv1a64.restype = C.c_uint64
        L.efxgen_fnv1a64.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.efxgen_batch_offsets.restype = C.c_int
        L.efxgen_batch_offsets.argtypes = [vp, C.c_int, vp, C.c_int]
        _lib = L
    return _lib


def fnv1a64(buf, h: int = 0xCBF29CE484222325) -> int:
    """FNV-1a-64 of a contiguous array's bytes (the hash of the committed golden vectors)."""
    a = np.ascontiguousarray(buf)
    return int(_load().efxgen_fnv1a64(a.ctypes.data, a.nbytes, h))


class Batch:
    """n_streams generated streams (ids first_id ...), n_pictures pictures each, GOP length gop."""

    def __init__(self, first_id: int, n_streams: int, n_pictures: int, gop: int = 12, flags: int = 0,
                 threads: int | None = None):
        L = _load()
        threads = threads or (os.cpu_count() or 1)
        self._h = L.efxgen_batch_create(first_id, n_streams, n_pictures, gop, flags, threads)
        if not self._h:
            raise ValueError("bad generator arguments")
        self.n_streams, self.n_pictures = n_streams, n_pictures

    def es(self, i: int) -> np.ndarray:
        L = _load()
        n = L.efxgen_batch_es_size(self._h, i)
        out = np.empty(n, dtype=np.uint8)
        L.efxgen_batch_es_copy(self._h, i, out.ctypes.data, n)
        return out

    def ts(self, i: int) -> np.ndarray:
        L = _load()
        n = L.efxgen_batch_ts(self._h, i, None, 0)
        out = np.empty(n, dtype=np.uint8)
        L.efxgen_batch_ts(self._h, i, out.ctypes.data, n)
        return out

    def picture_offsets(self, i: int) -> np.ndarray:
        out = np.zeros(self.n_pictures + 1, dtype=np.uint32)
        _load().efxgen_batch_offsets(self._h, i, out.ctypes.data, out.size)
        return out

    def all_es(self):
        return [self.es(i) for i in range(self.n_streams)]

    def close(self):
        if self._h:
            _load().efxgen_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""espflix_amd -- Python host binding (ctypes) for libefx, the MI355X-native espflix hot path.

This package is a thin mirror of the C-ABI in include/efx.h (which in turn stands in for the
reference's MpegDecoder / Frame / push_video / video_isr / write_pcm_16 surface, reference
src/player.h:34-165, src/video.h:36-50, src/video.cpp:1122, espflix.ino:123).  All work happens
in hand-written HIP kernels inside espflix_amd/libefx.so; there is NO CPU fallback: importing
works anywhere, but creating a Decoder without the built library or without a gfx950 device
raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

FRAME_WIDTH = 352
FRAME_HEIGHT = 192
FRAME_STRIDE = 528
STRIP_ROWS = 16
STRIPS = 12
STRIP_BYTES = 8448
FRAME_BYTES = 101376

OPT_GROUPS, OPT_PARSE_CAP, OPT_RECON_MODE, OPT_RECON_WAVES, OPT_RECON_SPINS, OPT_RECON_ITEMS, OPT_SBC_SERIAL, OPT_DEMUX_FUSED = 1, 2, 3, 4, 5, 6, 7, 8   # efx_option
SBC_PROBE_FIRST = 1
FORMAT_ES = 0
FORMAT_TS = 1

STREAM_BAD_SIZE = 1
STREAM_TRUNCATED = 2
STREAM_TOO_MANY_UNITS = 4
STREAM_BAD_VLC = 8
STREAM_MB_OVERRUN = 16
STREAM_COEF_OVERRUN = 32
STREAM_SERIAL_HUNT = 64   # the reference would misread bits between two start codes (its marker hunt is bit-serial): see efx.h
STREAM_INTERNAL = 256     # never expected: a lost hand-over inside the reconstruction kernel (efx.h)
STREAM_SLICE_ORDER = 128  # slice start codes of a picture not strictly rising in bitstream order: see efx.h

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EFX_LIB") or os.path.join(_HERE, "libefx.so")  # EFX_LIB: development builds


class EfxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libefx status {status}: {message}")
        self.status = status


class _Config(C.Structure):
    _fields_ = [
        ("device", C.c_int),
        ("max_streams", C.c_int),
        ("max_pictures", C.c_int),
        ("ring_depth", C.c_int),
        ("max_stream_bytes", C.c_size_t),
        ("hip_stream", C.c_void_p),
    ]


class _VideoParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("line_width", "line_count", "hsync", "hsync_long", "hsync_short",
                                       "burst_start", "burst_width", "active_start")]


class _FieldOpts(C.Structure):
    _fields_ = [("first_stream", C.c_int), ("n_streams", C.c_int), ("slot", C.c_int), ("other_slot", C.c_int),
                ("ntsc", C.c_int), ("frame_counter", C.c_int), ("hscroll", C.c_int), ("overlay", C.c_void_p),
                ("overlay_stride", C.c_size_t), ("overlay_blend", C.c_int), ("overlay_progress", C.c_int)]


class _IdxRec(C.Structure):
    _fields_ = [("first_pts", C.c_int64), ("last_pts", C.c_int64), ("bin_size", C.c_uint32), ("trick_speed", C.c_uint32),
                ("sample_count", C.c_uint32), ("reserved", C.c_uint32)]


class _Timing(C.Structure):
    _fields_ = [("index_ms", C.c_float), ("parse_ms", C.c_float), ("recon_ms", C.c_float), ("total_ms", C.c_float),
                ("pictures", C.c_uint64), ("slices", C.c_uint64), ("coefficients", C.c_uint64), ("es_bytes", C.c_uint64),
                ("demux_ms", C.c_float), ("timed_calls", C.c_uint32), ("ts_bytes", C.c_uint64), ("groups", C.c_uint32),
                ("parse_halves", C.c_uint16), ("mixed", C.c_uint16), ("recon_launches", C.c_uint32)]


# every symbol include/efx.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SYMBOLS = {
    "efx_create": (C.c_int, [C.POINTER(_Config), C.POINTER(_P)]),
    "efx_destroy": (None, [_P]),
    "efx_last_error": (C.c_char_p, [_P]),
    "efx_status_string": (C.c_char_p, [C.c_int]),
    "efx_upload_streams": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t), C.c_int]),
    "efx_upload_streams_inplace": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t), C.c_int]),
    "efx_download_es": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "efx_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "efx_host_free": (C.c_int, [_P, _P]),
    "efx_host_register": (C.c_int, [_P, _P, C.c_size_t]),
    "efx_host_unregister": (C.c_int, [_P, _P]),
    "efx_stream_layout": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "efx_upload_done": (C.c_int, [_P]),
    "efx_debug_poke": (C.c_int, [_P, _P, C.c_size_t, C.c_longlong]),
    "efx_debug_recon_stats": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "efx_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "efx_get_option": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "efx_reset": (C.c_int, [_P]),
    "efx_erase_frames": (C.c_int, [_P]),
    "efx_play_reset": (C.c_int, [_P]),
    "efx_stream_state": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "efx_decode": (C.c_int, [_P]),
    "efx_decode_from": (C.c_int, [_P, C.c_int]),
    "efx_decode_range": (C.c_int, [_P, C.c_int, C.c_int]),
    "efx_stream_picture_slot": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "efx_sync": (C.c_int, [_P]),
    "efx_picture_count": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "efx_stream_status": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    "efx_picture_pts": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "efx_picture_slot": (C.c_int, [_P, C.c_int]),
    "efx_frame_device_ptr": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "efx_download_frame": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "efx_frame_hashes": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "efx_upload_frame": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "efx_video_get_params": (C.c_int, [C.c_int, C.POINTER(_VideoParams)]),
    "efx_composite_fields": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "efx_composite_fields_ex": (C.c_int, [_P, C.POINTER(_FieldOpts), _P]),
    "efx_demux_audio": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t), _P, C.c_size_t, _P]),
    "efx_index_streams": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t), _P, C.c_uint32, _P, _P, C.c_size_t]),
    "efx_idx_build": (C.c_size_t, [_P, C.POINTER(_P), _P, C.c_size_t]),
    "efx_idx_pts2offset": (C.c_uint32, [_P, C.c_int64, C.c_int]),
    "efx_idx_pts2pts": (C.c_int64, [_P, C.c_int64, C.c_int]),
    "efx_sbc_state_bytes": (C.c_size_t, []),
    "efx_sbc_decode": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, _P, C.c_int]),
    "efx_pdm": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P]),
    "efx_set_timing": (C.c_int, [_P, C.c_int]),
    "efx_get_timing": (C.c_int, [_P, C.POINTER(_Timing)]),
    "efx_partition_first": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "efx_numa_node_of_pci": (C.c_int, [C.c_char_p]),
    "efx_numa_cpus_of_node": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int]),
    "efx_numa_bind_thread": (C.c_int, [C.c_int]),
    "efx_multi_create": (C.c_int, [C.POINTER(_Config), C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "efx_multi_destroy": (None, [_P]),
    "efx_multi_device_count": (C.c_int, [_P]),
    "efx_multi_context": (_P, [_P, C.c_int]),
    "efx_multi_last_error": (C.c_char_p, [_P]),
    "efx_multi_upload_streams": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t), C.c_int]),
    "efx_multi_locate": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "efx_multi_decode": (C.c_int, [_P]),
    "efx_multi_sync": (C.c_int, [_P]),
    "efx_multi_reset": (C.c_int, [_P]),
    "efx_multi_results": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "efx_multi_frame_hashes": (C.c_int, [_P, _P]),
    "efx_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "efx_device_free": (C.c_int, [_P, _P]),
    "efx_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "efx_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
}

_lib = None


def load_library() -> C.CDLL:
    """Load libefx.so and bind every entry point; raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()); "
                          "espflix_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the C-ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def video_params(ntsc: bool) -> dict:
    """Geometry video_init(ntsc) establishes (reference src/video.cpp:572-630)."""
    p = _VideoParams()
    _check(None, load_library().efx_video_get_params(1 if ntsc else 0, C.byref(p)))
    return {n: getattr(p, n) for n, _ in _VideoParams._fields_}


def idx_build(recs3, samples3) -> bytes:
    """merge_index: video.idx bytes from three (rec dict, samples) results of Decoder.index_streams."""
    lib = load_library()
    recs = (_IdxRec * 3)(*[_IdxRec(**r) for r in recs3])
    arrs = [np.ascontiguousarray(s, dtype=np.uint32) for s in samples3]
    ptrs = (_P * 3)(*[a.ctypes.data for a in arrs])
    n = lib.efx_idx_build(recs, ptrs, None, 0)
    out = np.zeros(n, dtype=np.uint8)
    assert lib.efx_idx_build(recs, ptrs, out.ctypes.data, n) == n
    return out.tobytes()


def idx_pts2offset(idx: bytes, pts: int, speed: int) -> int:
    h = np.frombuffer(idx[:104], dtype=np.uint8).copy()
    return int(load_library().efx_idx_pts2offset(h.ctypes.data, pts, speed))


def idx_pts2pts(idx: bytes, pts: int, speed: int) -> int:
    h = np.frombuffer(idx[:104], dtype=np.uint8).copy()
    return int(load_library().efx_idx_pts2pts(h.ctypes.data, pts, speed))


def sbc_state_bytes() -> int:
    return int(load_library().efx_sbc_state_bytes())


def _check(ctx, status: int):
    if status != 0:
        lib = load_library()
        msg = lib.efx_status_string(status).decode()
        if ctx:
            detail = lib.efx_last_error(ctx).decode()
            if detail:
                msg += f" ({detail})"
        raise EfxError(status, msg)


@dataclass
class Timing:
    index_ms: float
    parse_ms: float
    recon_ms: float
    total_ms: float
    pictures: int
    slices: int
    coefficients: int
    es_bytes: int
    demux_ms: float = 0.0
    ts_bytes: int = 0
    timed_calls: int = 0
    groups: int = 1  # reconstruction groups of the newest call: one k_recon launch per group and picture index
    parse_halves: int = 1  # parse halves of the newest call (side by side on the parse streams)
    mixed: int = 0   # 1: the averaged calls did not all run with that structure
    recon_launches: int = 0  # reconstruction kernel launches of the newest call (groups x pictures, or groups with k_recon_all)


class DeviceBuffer:
    """A raw HBM allocation owned by a Decoder context."""

    def __init__(self, dec: "Decoder", nbytes: int):
        self._dec = dec
        self.nbytes = nbytes
        p = _P()
        _check(dec._ctx, dec._lib.efx_device_alloc(dec._ctx, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        _check(self._dec._ctx, self._dec._lib.efx_memcpy_h2d(self._dec._ctx, self.ptr, a.ctypes.data, a.nbytes))

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(self._dec._ctx, self._dec._lib.efx_memcpy_d2h(self._dec._ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self._dec._lib.efx_device_free(self._dec._ctx, self.ptr)
            self.ptr = None


class Decoder:
    """Batched MpegDecoder: `max_streams` independent streams, `max_pictures` pictures per decode.

    ring_depth = 2 reproduces the reference's two frame buffers (_fb[2], src/player.h:37-40);
    ring_depth = max_pictures + 1 keeps every decoded picture resident.
    """

    def __init__(self, max_streams: int, max_pictures: int, ring_depth: int = 2, device: int = 0,
                 max_stream_bytes: int = 0, hip_stream: int = 0):
        self._lib = load_library()
        self._ctx = _P()
        cfg = _Config(device, max_streams, max_pictures, ring_depth, max_stream_bytes, hip_stream or None)
        _check(None, self._lib.efx_create(C.byref(cfg), C.byref(self._ctx)))
        self.max_streams, self.max_pictures, self.ring_depth = max_streams, max_pictures, max(2, ring_depth)
        self.n_streams = 0

    def close(self):
        if self._ctx:
            self._lib.efx_destroy(self._ctx)
            self._ctx = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- bitstream in ---------------------------------------------------------------------
    def upload(self, streams, fmt: int = FORMAT_ES):
        """streams: sequence of bytes / uint8 arrays (one per stream)."""
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray, memoryview))
                else np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        ptrs = (_P * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        _check(self._ctx, self._lib.efx_upload_streams(self._ctx, n, ptrs, lens, fmt))
        self.n_streams = n

    def prepare_upload(self, streams):
        """The ctypes argument arrays of upload() for a batch that is uploaded repeatedly (ingest benchmarks)."""
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray, memoryview))
                else np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        return arrs, (_P * n)(*[a.ctypes.data for a in arrs]), (C.c_size_t * n)(*[a.size for a in arrs]), n

    def upload_prepared(self, prepared, fmt: int = FORMAT_ES, in_place: bool = False):
        """in_place: efx_upload_streams_inplace -- the batch must lie in an arena of this context (place_in_arena); the
        library writes the tails into the layout's gaps and the transfer reads the arena.  Otherwise the staged path."""
        _, ptrs, lens, n = prepared
        fn = self._lib.efx_upload_streams_inplace if in_place else self._lib.efx_upload_streams
        _check(self._ctx, fn(self._ctx, n, ptrs, lens, fmt))
        self.n_streams = n

    # -- in-place ingest (efx_host_alloc / efx_stream_layout / efx_upload_done) ---------------
    def host_arena(self, nbytes: int) -> np.ndarray:
        """Page-locked host memory of this context as a uint8 array (freed with the context)."""
        p = _P()
        _check(self._ctx, self._lib.efx_host_alloc(self._ctx, nbytes, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def place_in_arena(self, arena: np.ndarray, streams):
        """Lay a batch out in `arena` the way efx_stream_layout prescribes and return the prepared argument arrays
        for upload_prepared(..., in_place=True): such a batch is transferred straight from the arena, no staging copy."""
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray, memoryview))
                else np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        off = (C.c_size_t * (n + 1))()
        _check(self._ctx, self._lib.efx_stream_layout(n, lens, off))
        assert off[n] <= arena.size, "arena too small for the batch"
        base = arena.ctypes.data
        for a, o in zip(arrs, off):
            arena[o:o + a.size] = a
        return arena, (_P * n)(*[base + off[i] for i in range(n)]), lens, n

    def upload_done(self) -> bool:
        r = self._lib.efx_upload_done(self._ctx)
        if r < 0:
            _check(self._ctx, r)
        return bool(r)

    def set_option(self, option: int, value: int):
        _check(self._ctx, self._lib.efx_set_option(self._ctx, option, value))

    def recon_stats(self):
        """Header words of k_recon_all's hand-over buffer after the most recent call (development aid)."""
        out = (C.c_uint32 * 64)()
        _check(self._ctx, self._lib.efx_debug_recon_stats(self._ctx, out))
        return list(out)

    def get_option(self, option: int) -> int:
        v = C.c_int()
        _check(self._ctx, self._lib.efx_get_option(self._ctx, option, C.byref(v)))
        return v.value

    def reset(self):
        _check(self._ctx, self._lib.efx_reset(self._ctx))

    def play_reset(self):
        """MpegDecoder::reset() between plays: the PTS latch is cleared, the ring position survives."""
        _check(self._ctx, self._lib.efx_play_reset(self._ctx))

    def stream_state(self, stream: int = 0):
        """(frame index, a PTS has been latched, newest PES PTS) of a stream."""
        fi, seen, pts = C.c_uint32(), C.c_int(), C.c_int64()
        _check(self._ctx, self._lib.efx_stream_state(self._ctx, stream, C.byref(fi), C.byref(seen), C.byref(pts)))
        return fi.value, bool(seen.value), pts.value

    def erase_frames(self):
        _check(self._ctx, self._lib.efx_erase_frames(self._ctx))

    # -- decode ---------------------------------------------------------------------------
    def decode(self, sync: bool = True, first_picture: int = 0, n_pictures: int | None = None):
        """efx_decode (first_picture = 0) / efx_decode_from / efx_decode_range (at most n_pictures pictures per stream):
        the decoder keeps going from call to call."""
        if n_pictures is None:
            _check(self._ctx, self._lib.efx_decode_from(self._ctx, first_picture))
        else:
            _check(self._ctx, self._lib.efx_decode_range(self._ctx, first_picture, n_pictures))
        if sync:
            self.sync()

    def sync(self):
        _check(self._ctx, self._lib.efx_sync(self._ctx))

    def picture_count(self, stream: int) -> int:
        n = C.c_int()
        _check(self._ctx, self._lib.efx_picture_count(self._ctx, stream, C.byref(n)))
        return n.value

    def stream_status(self, stream: int) -> int:
        b = C.c_uint32()
        _check(self._ctx, self._lib.efx_stream_status(self._ctx, stream, C.byref(b)))
        return b.value

    def es(self, stream: int) -> bytes:
        """The elementary stream the decoder sees for `stream` (device-demultiplexed for TS input)."""
        n = C.c_size_t()
        _check(self._ctx, self._lib.efx_download_es(self._ctx, stream, None, 0, C.byref(n)))
        buf = (C.c_uint8 * max(n.value, 1))()
        _check(self._ctx, self._lib.efx_download_es(self._ctx, stream, buf, n.value, C.byref(n)))
        return bytes(buf[: n.value])

    def picture_pts(self, stream: int, picture: int) -> int:
        p = C.c_int64()
        _check(self._ctx, self._lib.efx_picture_pts(self._ctx, stream, picture, C.byref(p)))
        return p.value

    def picture_slot(self, picture: int, stream: int = 0) -> int:
        """Ring slot of a picture of the last decode (synchronises)."""
        slot = C.c_int()
        _check(self._ctx, self._lib.efx_stream_picture_slot(self._ctx, stream, picture, C.byref(slot)))
        return slot.value

    # -- frames out -----------------------------------------------------------------------
    def frame_ptr(self, stream: int, slot: int) -> int:
        p = _P()
        _check(self._ctx, self._lib.efx_frame_device_ptr(self._ctx, stream, slot, C.byref(p)))
        return p.value

    def download_frame(self, stream: int, slot: int) -> np.ndarray:
        out = np.empty(FRAME_BYTES, dtype=np.uint8)
        _check(self._ctx, self._lib.efx_download_frame(self._ctx, stream, slot, out.ctypes.data))
        return out

    def download_picture(self, stream: int, picture: int) -> np.ndarray:
        return self.download_frame(stream, self.picture_slot(picture))

    def upload_frame(self, stream: int, slot: int, frame: np.ndarray):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        assert f.size == FRAME_BYTES
        _check(self._ctx, self._lib.efx_upload_frame(self._ctx, stream, slot, f.ctypes.data))

    def frame_hashes(self, first_stream: int = 0, n: int | None = None) -> np.ndarray:
        """FNV-1a-64 of every ring frame, shape (n, ring_depth), computed on the device."""
        n = self.n_streams - first_stream if n is None else n
        out = np.empty((n, self.ring_depth), dtype=np.uint64)
        _check(self._ctx, self._lib.efx_frame_hashes(self._ctx, first_stream, n, out.ctypes.data))
        return out

    # -- video / audio out ----------------------------------------------------------------
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def composite_fields(self, dst: DeviceBuffer | int, first_stream: int, n_streams: int, slot: int, ntsc: bool,
                         frame_counter: int):
        ptr = dst.ptr if isinstance(dst, DeviceBuffer) else dst
        _check(self._ctx, self._lib.efx_composite_fields(self._ctx, first_stream, n_streams, slot, 1 if ntsc else 0,
                                                         frame_counter, ptr))

    def composite_fields_ex(self, dst: DeviceBuffer | int, first_stream: int, n_streams: int, slot: int, ntsc: bool,
                            frame_counter: int, other_slot: int | None = None, hscroll: int = 0,
                            overlay: DeviceBuffer | int | None = None, overlay_stride: int = 0, overlay_blend: int = 0,
                            overlay_progress: int = 0):
        """video_isr with the two-frame slide (_hscroll) and the overlay / progress bar (composite())."""
        g = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        o = _FieldOpts(first_stream, n_streams, slot, slot if other_slot is None else other_slot, 1 if ntsc else 0,
                       frame_counter, hscroll, g(overlay), overlay_stride, overlay_blend, overlay_progress)
        _check(self._ctx, self._lib.efx_composite_fields_ex(self._ctx, C.byref(o), g(dst)))

    def demux_audio(self, streams, audio: DeviceBuffer | int, stride: int, audio_len: DeviceBuffer | int):
        """Audio elementary streams (push_audio's input) of a batch of transport streams, on the device."""
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray, memoryview))
                else np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        ptrs = (_P * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        g = lambda b: b.ptr if isinstance(b, DeviceBuffer) else b
        _check(self._ctx, self._lib.efx_demux_audio(self._ctx, n, ptrs, lens, g(audio), stride, g(audio_len)))

    def index_streams(self, streams, trick_speed=None, bin_size: int = 7500, samples_cap: int = 0):
        """make_index + pts2seq for a batch of transport streams.  Returns [(rec dict, samples)]."""
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray, memoryview))
                else np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        cap = samples_cap or max(16, max(a.size // 188 for a in arrs) + 2)
        ptrs = (_P * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        recs = (_IdxRec * n)()
        samples = np.zeros((n, cap), dtype=np.uint32)
        sp = None if trick_speed is None else (C.c_uint32 * n)(*trick_speed)
        _check(self._ctx, self._lib.efx_index_streams(self._ctx, n, ptrs, lens, sp, bin_size, recs, samples.ctypes.data, cap))
        return [({f: getattr(recs[i], f) for f, _ in _IdxRec._fields_}, samples[i, :recs[i].sample_count].copy())
                for i in range(n)]

    def sbc_decode(self, n_streams: int, frames: DeviceBuffer | int, stream_stride: int, frame_bytes: int, n_frames: int,
                   state: DeviceBuffer | int, pcm: DeviceBuffer | int, pcm_stride: int, ret: DeviceBuffer | int | None = None,
                   pcm_count: DeviceBuffer | int | None = None, probe_first: bool = False):
        g = lambda b: None if b is None else (b.ptr if isinstance(b, DeviceBuffer) else b)
        _check(self._ctx, self._lib.efx_sbc_decode(self._ctx, n_streams, g(frames), stream_stride, frame_bytes, n_frames,
                                                   g(state), g(pcm), pcm_stride, g(ret), g(pcm_count), 1 if probe_first else 0))

    def pdm(self, n_streams: int, pcm: DeviceBuffer | int, n_samples: int, state: DeviceBuffer | int,
            dst: DeviceBuffer | int):
        g = lambda b: b.ptr if isinstance(b, DeviceBuffer) else b
        _check(self._ctx, self._lib.efx_pdm(self._ctx, n_streams, g(pcm), n_samples, g(state), g(dst)))

    # -- measurement ----------------------------------------------------------------------
    def set_timing(self, enable: bool):
        _check(self._ctx, self._lib.efx_set_timing(self._ctx, 1 if enable else 0))

    def timing(self) -> Timing:
        t = _Timing()
        _check(self._ctx, self._lib.efx_get_timing(self._ctx, C.byref(t)))
        return Timing(t.index_ms, t.parse_ms, t.recon_ms, t.total_ms, t.pictures, t.slices, t.coefficients, t.es_bytes,
                      t.demux_ms, t.ts_bytes, t.timed_calls, max(1, t.groups), max(1, t.parse_halves), t.mixed, t.recon_launches)


def partition_first(total: int, parts: int, part: int) -> int:
    """First stream of part `part` when `total` streams are dealt to `parts` devices (efx_partition_first): stream k lives on
    device floor(k * parts / total).  Host-only."""
    return load_library().efx_partition_first(total, parts, part)


def numa_node_of_pci(bus_id: str) -> int:
    """NUMA node sysfs publishes for a PCI device (efx_numa_node_of_pci; -1: unknown).  Host-only."""
    return load_library().efx_numa_node_of_pci(bus_id.encode())


def numa_cpus_of_node(node: int):
    buf = (C.c_int * 4096)()
    n = load_library().efx_numa_cpus_of_node(node, buf, 4096)
    return [buf[i] for i in range(min(n, 4096))]


def numa_bind_thread(node: int) -> int:
    """Bind the calling thread to the CPUs of `node` that it may already use; returns how many (0: mask left alone)."""
    return load_library().efx_numa_bind_thread(node)


class MultiDecoder:
    """efx_multi: one context and one host thread per device of a node, streams dealt in contiguous blocks (no collective
    on the data path).  max_streams is the per-device capacity."""

    def __init__(self, devices, max_streams: int, max_pictures: int, ring_depth: int = 2, max_stream_bytes: int = 0):
        self._lib = load_library()
        self._m = _P()
        cfg = _Config(0, max_streams, max_pictures, ring_depth, max_stream_bytes, None)
        devs = (C.c_int * len(devices))(*devices)
        st = self._lib.efx_multi_create(C.byref(cfg), devs, len(devices), C.byref(self._m))
        if st != 0:
            raise EfxError(st, self._lib.efx_status_string(st).decode())
        self.ring_depth, self.n_streams = max(2, ring_depth), 0

    def _check(self, st):
        if st != 0:
            raise EfxError(st, (self._lib.efx_multi_last_error(self._m) or b"").decode() or self._lib.efx_status_string(st).decode())

    def upload(self, streams, fmt: int = FORMAT_ES):
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        n = len(arrs)
        ptrs = (_P * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        self._check(self._lib.efx_multi_upload_streams(self._m, n, ptrs, lens, fmt))
        self.n_streams = n

    def decode(self, sync: bool = True):
        self._check(self._lib.efx_multi_decode(self._m))
        if sync:
            self.sync()

    def sync(self):
        self._check(self._lib.efx_multi_sync(self._m))

    def locate(self, stream: int):
        d, l = C.c_int(), C.c_int()
        self._check(self._lib.efx_multi_locate(self._m, stream, C.byref(d), C.byref(l)))
        return d.value, l.value

    def results(self):
        n = (C.c_int * self.n_streams)()
        st = (C.c_uint32 * self.n_streams)()
        self._check(self._lib.efx_multi_results(self._m, n, st))
        return np.array(n[:]), np.array(st[:])

    def frame_hashes(self) -> np.ndarray:
        out = np.empty((self.n_streams, self.ring_depth), dtype=np.uint64)
        self._check(self._lib.efx_multi_frame_hashes(self._m, out.ctypes.data))
        return out

    def close(self):
        if self._m:
            self._lib.efx_multi_destroy(self._m)
            self._m = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

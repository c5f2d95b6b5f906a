"""ctypes binding of the TEST oracle (oracle/_build/libefx_oracle.so) and of the compiled
reference harnesses (oracle/_ref/*).  Only tests, smoke() and bench.py's cpu_baseline leg use it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "_build", "libefx_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
FRAME_BYTES = 101376
FNV_BASIS = 0xCBF29CE484222325

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, capture_output=True)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.efxo_decode.restype = C.c_long
        L.efxo_decode.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp, vp, C.c_long]
        L.efxo_ts_to_es.restype = C.c_size_t
        L.efxo_ts_to_es.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        L.efxo_fnv1a64.restype = C.c_uint64
        L.efxo_fnv1a64.argtypes = [vp, C.c_size_t, C.c_uint64]
        L.efxo_video_field.restype = C.c_long
        L.efxo_video_field.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.efxo_video_params.argtypes = [C.c_int, vp]
        L.efxo_color_tab.argtypes = [C.c_int, vp]
        L.efxo_tables.argtypes = [vp, vp]
        L.efxo_pdm_second_order.argtypes = [vp, vp, vp, C.c_int]
        L.efxo_write_pcm_16.argtypes = [vp, vp, vp, C.c_int, vp]
        _lib = L
    return _lib


def decode(data: np.ndarray, fmt: int, flush_last: bool = True, want_frames: bool = False, max_frames: int = 512):
    """Returns (n, hashes[n], pts[n], frames[n, FRAME_BYTES] or None)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    hashes = np.zeros(max_frames, dtype=np.uint64)
    pts = np.zeros(max_frames, dtype=np.int64)
    frames = np.zeros((max_frames, FRAME_BYTES), dtype=np.uint8) if want_frames else None
    n = lib().efxo_decode(data.ctypes.data, data.size, fmt, 1 if flush_last else 0,
                          frames.ctypes.data if want_frames else None, pts.ctypes.data, hashes.ctypes.data, max_frames)
    n = min(n, max_frames)
    return n, hashes[:n], pts[:n], (frames[:n] if want_frames else None)


def trace_levels(data: np.ndarray, fmt: int) -> dict:
    """Runs the oracle with its parse trace on (efxo_set_trace) and summarises the AC levels it decoded -- the
    values of player.cpp:1087-1103 before dequantisation: min / max, how many lie outside -127..127 (only the 16-bit
    escape forms reach those), coded zeros, -256, the pictures' coding types and the longest zero run."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    st = {"min": 0, "max": 0, "wide": 0, "zero": 0, "m256": 0, "coefs": 0, "max_run": 0, "pic_types": set(), "r_sizes": set(),
          "full_pel": set(), "abandoned": 0, "bad": 0, "esc_forms": [0, 0, 0], "esc_small_long": 0, "esc_max_run": 0,
          "slices": 0, "slice_extra": {}}  # slice_extra: (picture type, extra_information_slice bytes) -> slices
    cur = {"intra": False, "last": -1}

    def cb(_user, kind, a, b, c, e):
        if kind == 0:      # slice: c = type | full_pel << 4 | r_size << 8 | decoded << 16
            st["slices"] += 1
            st["pic_types"].add(c & 15)
            if (c & 15) != 1:
                st["r_sizes"].add((c >> 8) & 0xFF)
                st["full_pel"].add((c >> 4) & 1)
        elif kind == 1:    # macroblock
            cur["intra"] = bool(b & 1)
        elif kind == 3:    # block end
            cur["last"] = -1
            if b == -1:
                st["abandoned"] += 1
            elif b == -2:
                st["bad"] += 1
        elif kind == 4:    # escape: a = form (0 "xx", 1 "00 xx", 2 "80 xx"), b = run, c = level
            st["esc_forms"][a] += 1
            st["esc_small_long"] += a != 0 and -127 <= c <= 127   # a level the short form could carry, sent the long way
            st["esc_max_run"] = max(st["esc_max_run"], b)
        elif kind == 5:    # slice header with extra_information_slice: a = bytes, c = picture type
            st["slice_extra"][(c, a)] = st["slice_extra"].get((c, a), 0) + 1
        elif kind == 2:    # coefficient: b = scan position, c = level
            if cur["intra"] and b == 0 and cur["last"] < 0:
                cur["last"] = 0
                return     # intra DC value
            st["coefs"] += 1
            st["max_run"] = max(st["max_run"], b - cur["last"] - 1)
            cur["last"] = b
            st["min"] = min(st["min"], c)
            st["max"] = max(st["max"], c)
            st["wide"] += c > 127 or c < -127
            st["zero"] += c == 0
            st["m256"] += c == -256

    FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
    fn = FN(cb)
    L = lib()
    L.efxo_set_trace.argtypes = [FN, C.c_void_p]
    L.efxo_set_trace(fn, None)
    try:
        L.efxo_decode(data.ctypes.data, data.size, fmt, 1, None, None, None, 0)
    finally:
        L.efxo_set_trace(FN(), None)
    return st


def ts_to_es(ts: np.ndarray) -> np.ndarray:
    ts = np.ascontiguousarray(ts, dtype=np.uint8)
    out = np.zeros(ts.size, dtype=np.uint8)
    n = lib().efxo_ts_to_es(ts.ctypes.data, ts.size, out.ctypes.data, out.size)
    return out[:n].copy()


def fnv1a64(buf: np.ndarray, h: int = FNV_BASIS) -> int:
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return int(lib().efxo_fnv1a64(buf.ctypes.data, buf.size, h))


def chain_hash(hashes) -> int:
    h = FNV_BASIS
    for x in hashes:
        h = fnv1a64(np.frombuffer(int(x).to_bytes(8, "little"), dtype=np.uint8), h)
    return h


def video_params(ntsc: bool) -> np.ndarray:
    p = np.zeros(8, dtype=np.int32)
    lib().efxo_video_params(1 if ntsc else 0, p.ctypes.data)
    return p


def video_field(frames2: np.ndarray, ntsc: bool, frame_counter0: int, nfields: int) -> np.ndarray:
    frames2 = np.ascontiguousarray(frames2, dtype=np.uint8)
    assert frames2.size == 2 * FRAME_BYTES
    p = video_params(ntsc)
    out = np.zeros((nfields, p[1], p[0]), dtype=np.uint16)
    r = lib().efxo_video_field(frames2.ctypes.data, 1 if ntsc else 0, frame_counter0, nfields, out.ctypes.data)
    assert r == p[0] * p[1]
    return out


def video_field_ex(frames2: np.ndarray, ntsc: bool, frame_counter0: int, nfields: int, front: int = 0, hscroll=None,
                   overlay=None, blend: int = 0, progress: int = 0) -> np.ndarray:
    """hscroll: one value per field or None; overlay: 1280 bytes or None; blend decrements per field."""
    frames2 = np.ascontiguousarray(frames2, dtype=np.uint8)
    assert frames2.size == 2 * FRAME_BYTES
    p = video_params(ntsc)
    out = np.zeros((nfields, p[1], p[0]), dtype=np.uint16)
    hs = None if hscroll is None else np.ascontiguousarray(hscroll, dtype=np.int16)
    ov = None if overlay is None else np.ascontiguousarray(overlay, dtype=np.uint8)
    assert hs is None or hs.size == nfields
    assert ov is None or ov.size == 1280
    L = lib()
    L.efxo_video_field_ex.restype = C.c_long
    L.efxo_video_field_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_void_p]
    r = L.efxo_video_field_ex(frames2.ctypes.data, 1 if ntsc else 0, frame_counter0, nfields, front,
                              hs.ctypes.data if hs is not None else None, ov.ctypes.data if ov is not None else None,
                              blend, progress, out.ctypes.data)
    assert r == p[0] * p[1]
    return out


def color_tab(ntsc: bool) -> np.ndarray:
    t = np.zeros(768, dtype=np.uint32)
    lib().efxo_color_tab(1 if ntsc else 0, t.ctypes.data)
    return t


def tables():
    zz = np.zeros(64, dtype=np.uint8)
    pm = np.zeros(64, dtype=np.uint8)
    lib().efxo_tables(zz.ctypes.data, pm.ctypes.data)
    return zz, pm


def pdm(state: np.ndarray, pcm: np.ndarray) -> np.ndarray:
    """pdm_second_order over the whole buffer; state (3 x int32) is updated in place."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.zeros(2 * pcm.size, dtype=np.uint16)
    lib().efxo_pdm_second_order(state.ctypes.data, out.ctypes.data, pcm.ctypes.data, pcm.size)
    return out


def write_pcm_16(state: np.ndarray, beep: C.c_int, samples) -> np.ndarray:
    out = np.zeros(256, dtype=np.uint16)
    if samples is None:
        lib().efxo_write_pcm_16(state.ctypes.data, C.addressof(beep), None, 128, out.ctypes.data)
    else:
        s = np.ascontiguousarray(samples, dtype=np.int16)
        lib().efxo_write_pcm_16(state.ctypes.data, C.addressof(beep), s.ctypes.data, s.size, out.ctypes.data)
    return out


# ---- the compiled, unmodified reference (only where oracle/_ref has been built) ---------------

def have_ref() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, n)) for n in ("efx_ref_decode", "efx_ref_video", "efx_ref_pdm"))


def ref_decode(ts: np.ndarray | str, flush_last: bool = True, want_frames: bool = False):
    """Run the reference decoder on a TS blob (or '@splash' / '@vmedia').  Returns
    (hashes, pts, frames or None)."""
    with tempfile.TemporaryDirectory() as td:
        if isinstance(ts, str):
            src = ts
        else:
            src = os.path.join(td, "in.ts")
            np.ascontiguousarray(ts, dtype=np.uint8).tofile(src)
        out = os.path.join(td, "out.bin") if want_frames else "-"
        cmd = [os.path.join(REF_DIR, "efx_ref_decode"), "decode", src, out] + (["flush"] if flush_last else [])
        p = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, timeout=120)
        rows = [l.split() for l in p.stderr.splitlines() if l.startswith("F ")]
        hashes = np.array([int(r[3], 16) for r in rows], dtype=np.uint64)
        pts = np.array([int(r[2]) for r in rows], dtype=np.int64)
        frames = np.fromfile(out, dtype=np.uint8).reshape(-1, FRAME_BYTES) if want_frames else None
        return hashes, pts, frames


def ref_audio_es(ts: np.ndarray) -> np.ndarray:
    """Every byte the reference's demux hands to push_audio() while it plays the transport stream."""
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(td, "in.ts"), os.path.join(td, "audio.bin")
        np.ascontiguousarray(ts, dtype=np.uint8).tofile(src)
        env = dict(os.environ, EFX_REF_AUDIO_OUT=out)
        subprocess.run([os.path.join(REF_DIR, "efx_ref_decode"), "decode", src, "-", "flush"], stderr=subprocess.DEVNULL,
                       stdout=subprocess.DEVNULL, timeout=120, env=env)
        return np.fromfile(out, dtype=np.uint8)


def ref_video_field(frames2: np.ndarray, ntsc: bool, nfields: int) -> np.ndarray:
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(td, "f.bin"), os.path.join(td, "o.bin")
        np.ascontiguousarray(frames2, dtype=np.uint8).tofile(src)
        subprocess.run([os.path.join(REF_DIR, "efx_ref_video"), "field", src, "1" if ntsc else "0", str(nfields), out],
                       check=True, timeout=600)
        p = video_params(ntsc)
        return np.fromfile(out, dtype=np.uint16).reshape(nfields, p[1], p[0])


def ref_video_field_ex(frames2: np.ndarray, ntsc: bool, nfields: int, front: int = 0, hscroll=None, overlay=None,
                       blend: int = 0, progress: int = 0) -> np.ndarray:
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(td, "f.bin"), os.path.join(td, "o.bin")
        np.ascontiguousarray(frames2, dtype=np.uint8).tofile(src)
        hs, ov = "-", "-"
        if hscroll is not None:
            hs = os.path.join(td, "h.bin")
            np.ascontiguousarray(hscroll, dtype=np.int16).tofile(hs)
        if overlay is not None:
            ov = os.path.join(td, "v.bin")
            np.ascontiguousarray(overlay, dtype=np.uint8).tofile(ov)
        subprocess.run([os.path.join(REF_DIR, "efx_ref_video"), "fieldx", src, "1" if ntsc else "0", str(nfields), out,
                        str(front), hs, ov, str(blend), str(progress)], check=True, timeout=600)
        p = video_params(ntsc)
        return np.fromfile(out, dtype=np.uint16).reshape(nfields, p[1], p[0])


def ref_video_params(ntsc: bool):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "p.bin")
        subprocess.run([os.path.join(REF_DIR, "efx_ref_video"), "params", "1" if ntsc else "0", out], check=True)
        raw = np.fromfile(out, dtype=np.uint32)
        return raw[:8].view(np.int32), raw[8:8 + 768], raw[8 + 768:8 + 768 + 8]


def ref_pdm(pcm: np.ndarray, silence_every: int = 0, beep_at: int = -1) -> np.ndarray:
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(td, "pcm.bin"), os.path.join(td, "o.bin")
        np.ascontiguousarray(pcm, dtype=np.int16).tofile(src)
        cmd = [os.path.join(REF_DIR, "efx_ref_pdm"), src, out]
        if silence_every:
            cmd += ["silence_every", str(silence_every)]
        if beep_at >= 0:
            cmd += ["beep_at", str(beep_at)]
        subprocess.run(cmd, check=True, timeout=600)
        return np.fromfile(out, dtype=np.uint16)


# ---- SBC audio ---------------------------------------------------------------------------------
class SbcState(C.Structure):
    _fields_ = [("hdr", C.c_uint8 * 8), ("sb_sample", C.c_int32 * 256), ("v", C.c_int32 * 340),
                ("v_offset", C.c_uint8 * 32)]


def sbc_tables():
    syn = np.zeros(128, dtype=np.int32)
    pro = np.zeros(80, dtype=np.int32)
    lib().efxo_sbc_tables.argtypes = [C.c_void_p, C.c_void_p]
    lib().efxo_sbc_tables(syn.ctypes.data, pro.ctypes.data)
    return syn, pro


def sbc_decode(frames: np.ndarray, frame_bytes: int, probe: bool = False):
    """Decode concatenated frames with the restatement.  Returns (pcm int16, [(ret, decoded)])."""
    L = lib()
    L.efxo_sbc_decode.restype = C.c_int
    L.efxo_sbc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.efxo_sbc_init.argtypes = [C.c_void_p]
    st = SbcState()
    L.efxo_sbc_init(C.byref(st))
    data = np.concatenate([np.ascontiguousarray(frames, dtype=np.uint8), np.zeros(1024, np.uint8)])
    n = (data.size - 1024) // frame_bytes
    order = ([0] if probe else []) + list(range(n))
    out, rets = [], []
    pcm = np.zeros(256, dtype=np.int16)
    for fi in order:
        dec = C.c_int(0)
        pcm[:] = 0
        r = L.efxo_sbc_decode(C.byref(st), data[fi * frame_bytes:].ctypes.data, frame_bytes, pcm.ctypes.data, C.byref(dec))
        rets.append((r, dec.value))
        out.append(pcm[:dec.value // 2].copy())
    return (np.concatenate(out) if out else np.zeros(0, np.int16)), rets


def ts_audio_es(ts: np.ndarray) -> np.ndarray:
    ts = np.ascontiguousarray(ts, dtype=np.uint8)
    out = np.zeros(ts.size, dtype=np.uint8)
    L = lib()
    L.efxo_ts_audio_es.restype = C.c_size_t
    L.efxo_ts_audio_es.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    n = L.efxo_ts_audio_es(ts.ctypes.data, ts.size, out.ctypes.data, out.size)
    return out[:n].copy()


def ref_sbc_decode(frames: np.ndarray, frame_bytes: int, probe: bool = False):
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(td, "f.bin"), os.path.join(td, "o.pcm")
        np.ascontiguousarray(frames, dtype=np.uint8).tofile(src)
        p = subprocess.run([os.path.join(REF_DIR, "efx_ref_sbc"), "decode", src, str(frame_bytes), out] +
                           (["probe"] if probe else []), check=True, timeout=120, stderr=subprocess.PIPE, text=True)
        rets = [(int(l.split()[2]), int(l.split()[3])) for l in p.stderr.splitlines() if l.startswith("R ")]
        return np.fromfile(out, dtype=np.int16), rets


def ref_sbc_tables():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.bin")
        subprocess.run([os.path.join(REF_DIR, "efx_ref_sbc"), "tables", out], check=True)
        t = np.fromfile(out, dtype=np.int32)
        return t[:128], t[128:208]


# ---- trick-play index --------------------------------------------------------------------------
IDX_HDR_BYTES = 104
IDX_PAD = [slice(36, 40), slice(68, 72), slice(100, 104)]   # tail padding of the three idx_rec


def idx_masked(b) -> np.ndarray:
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    for s in IDX_PAD:
        a[s] = 0
    return a


def make_idx(streams3) -> bytes:
    """video.idx bytes for (main, fwd, rwd) transport streams via the restatement."""
    L = lib()
    arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams3]
    ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * 3)(*[a.size for a in arrs])
    L.efxo_make_idx.restype = C.c_size_t
    L.efxo_make_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    cap = L.efxo_make_idx(ptrs, lens, None, 0)   # size query
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = L.efxo_make_idx(ptrs, lens, out.ctypes.data, out.size)
    assert n == cap
    return out[:n].tobytes()


def ts_sequences(ts):
    ts = np.ascontiguousarray(ts, dtype=np.uint8)
    cap = ts.size // 188 + 1
    sp, so = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.uint32)
    first, last = C.c_int64(0), C.c_int64(0)
    L = lib()
    L.efxo_ts_sequences.restype = C.c_long
    L.efxo_ts_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    n = L.efxo_ts_sequences(ts.ctypes.data, ts.size, C.byref(first), C.byref(last), sp.ctypes.data, so.ctypes.data, cap)
    return first.value, last.value, sp[:n].copy(), so[:n].copy()


def idx_query(hdr: bytes, pts: int, speed: int):
    L = lib()
    L.efxo_idx_pts2offset.restype = C.c_uint32
    L.efxo_idx_pts2offset.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.efxo_idx_pts2pts.restype = C.c_int64
    L.efxo_idx_pts2pts.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    h = np.frombuffer(hdr[:IDX_HDR_BYTES], dtype=np.uint8).copy()
    return int(L.efxo_idx_pts2offset(h.ctypes.data, pts, speed)), int(L.efxo_idx_pts2pts(h.ctypes.data, pts, speed))


def ref_make_idx(streams3) -> bytes:
    with tempfile.TemporaryDirectory() as td:
        for name, s in zip(("video.ts", "video_fwd.ts", "video_rwd.ts"), streams3):
            np.ascontiguousarray(s, dtype=np.uint8).tofile(os.path.join(td, name))
        subprocess.run([os.path.join(REF_DIR, "efx_ref_index"), td], check=True, timeout=120)
        return open(os.path.join(td, "video.idx"), "rb").read()


def ref_idx_query(idx: bytes, queries):
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "video.idx")
        open(path, "wb").write(idx)
        text = "".join(f"{p} {s}\n" for p, s in queries)
        r = subprocess.run([os.path.join(REF_DIR, "efx_ref_idx"), path], input=text, capture_output=True, text=True,
                           check=True, timeout=120)
        return [tuple(int(x) for x in l.split()) for l in r.stdout.splitlines()]

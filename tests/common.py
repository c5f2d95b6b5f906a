"""Shared deterministic inputs for the tests and for tests/golden/make_golden.py."""
import numpy as np

FRAME_BYTES = 101376
SYN_FLAGS = [0, 1, 2, 4, 8, 16]   # generator flavours (espflix_amd.gen.FLAG_*)
SYN_IDS = [0, 1, 7]               # 7: full_pel_forward = 1, odd ids: forward_f_code = 2


def lcg_frames(seed: int = 12345) -> np.ndarray:
    """Two frames filled by the LCG of SURVEY.md section 8c: s = s*1664525 + 1013904223,
    byte = min(s >> 24, 248), strip by strip (rows of 528 bytes incl. chroma)."""
    out = np.empty(2 * FRAME_BYTES, dtype=np.uint8)
    s = np.uint64(seed)
    vals = np.empty(out.size, dtype=np.uint64)
    x = int(seed)
    for i in range(out.size):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = x >> 24
    return np.minimum(vals, 248).astype(np.uint8)


def random_frames(seed: int) -> np.ndarray:
    """Two frames of full-range random bytes (exercises the blitter's carries)."""
    return np.random.default_rng(seed).integers(0, 256, size=2 * FRAME_BYTES, dtype=np.uint8)


def pdm_pcm(k: int = 0, calls: int = 40) -> np.ndarray:
    """x[n] = 8000 sin(2 pi (220+k) n / 48000), plus two calls of
    full-scale noise."""
    n = np.arange(128 * calls)
    x = np.round(8000 * np.sin(2 * np.pi * (220 + k) * n / 48000)).astype(np.int16)
    rng = np.random.default_rng(100 + k)
    x[128 * 10:128 * 12] = rng.integers(-32768, 32767, 256)
    return x


def fnv_bytes(a: np.ndarray) -> int:
    import oracle
    return oracle.fnv1a64(np.ascontiguousarray(a).view(np.uint8).reshape(-1))

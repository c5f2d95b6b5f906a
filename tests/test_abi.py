"""The C-ABI library loads and exports every symbol include/efx.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

import espflix_amd as efx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "efx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(efx_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(efx._SYMBOLS)


def test_library_exports_every_symbol():
    assert os.path.exists(efx.LIB_PATH), "build libefx.so first (make lib / __graft_entry__.build())"
    lib = ctypes.CDLL(efx.LIB_PATH)
    for name in declared_symbols():
        getattr(lib, name)


def test_host_only_entry_points(golden):
    assert efx.load_library().efx_status_string(0) == b"ok"
    for ntsc in (True, False):
        p = efx.video_params(ntsc)
        want = golden["tables"]["params_" + ("ntsc" if ntsc else "pal")]
        assert [p[k] for k in ("line_width", "line_count", "hsync", "hsync_long", "hsync_short", "burst_start",
                               "burst_width", "active_start")] == want


def test_constants_match_header():
    hdr = open(os.path.join(ROOT, "include", "efx.h")).read()
    for name, val in (("EFX_FRAME_BYTES", efx.FRAME_BYTES), ("EFX_FRAME_STRIDE", efx.FRAME_STRIDE),
                      ("EFX_STRIP_BYTES", efx.STRIP_BYTES), ("EFX_STRIPS", efx.STRIPS)):
        assert int(re.search(name + r"\s+(\d+)", hdr).group(1)) == val


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(efx.EfxError) as e:
        efx.Decoder(1, 1)
    assert e.value.status in (-2, -3)


def test_stream_partition_is_the_one_of_the_scaling_job():
    """efx_partition_first (the multi-device entry points' partition, host-only): stream k of n lives on device
    floor(k * R / n) -- the blocks bench.py's ranks take (espflix_amd.dist.shard_fixed) -- every stream exactly once,
    block sizes differing by at most one."""
    from espflix_amd import dist
    for total, parts in ((8192, 8), (8192, 3), (1000, 7), (5, 8), (1, 1), (0, 4), (1024, 2)):
        firsts = [efx.partition_first(total, parts, r) for r in range(parts + 1)]
        assert firsts[0] == 0 and firsts[-1] == total and firsts == sorted(firsts)
        sizes = [b - a for a, b in zip(firsts, firsts[1:])]
        assert max(sizes) - min(sizes) <= 1
        for r in range(parts):
            if total:
                assert (firsts[r], firsts[r + 1]) == dist.shard_fixed(r, parts, total)
            for k in range(firsts[r], firsts[r + 1]):
                assert k * parts // total == r
    assert efx.partition_first(10, 0, 0) == -1 and efx.partition_first(-1, 2, 0) == -1


def test_stream_layout_is_the_upload_rule():
    """efx_stream_layout (host only): where a caller lays the streams of a batch out in a page-locked arena so that
    efx_upload_streams transfers them in place -- 16-byte aligned starts, room for the 9-byte end-of-data tail behind each."""
    import ctypes as C
    lib = efx.load_library()
    lens = [0, 1, 6, 7, 8, 23, 4096, 45001]
    arr = (C.c_size_t * len(lens))(*lens)
    off = (C.c_size_t * (len(lens) + 1))()
    assert lib.efx_stream_layout(len(lens), arr, off) == 0
    pos = 0
    for i, n in enumerate(lens):
        assert off[i] == pos and pos % 16 == 0
        pos += (n + 9 + 15) // 16 * 16
    assert off[len(lens)] == pos
    assert lib.efx_stream_layout(0, arr, off) != 0

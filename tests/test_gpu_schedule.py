"""Every launch structure on purpose (efx_set_option): results must not depend on how a call is cut into launches -- one
k_recon launch per picture index or one k_recon_all per group (eager / deferred hand-over signal, any number of items per wave),
one reconstruction group or several, a capped or an uncapped parse kernel -- and the in-place ingest path must deliver the
same batch as the staged one.  Goldens: the unmodified reference decoder's per-picture hashes (tests/golden/bench_gop12.u64)."""
import os

import numpy as np
import pytest

import common
import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 12


@pytest.fixture(scope="module")
def efx():
    import espflix_amd
    espflix_amd.load_library()
    return espflix_amd


@pytest.fixture(scope="module")
def batch():
    from espflix_amd import gen
    golden = np.fromfile(os.path.join(ROOT, "tests", "golden", "bench_gop12.u64"), dtype="<u8").reshape(8192, P)
    return gen.Batch(0, 128, P, 12, 0).all_es(), golden


def all_pictures(efx, dec, n):
    h = dec.frame_hashes()
    return np.stack([h[:n, dec.picture_slot(p)] for p in range(P)], axis=1)


@pytest.mark.parametrize("mode,items", [(0, 16), (1, 16), (2, 16), (2, 1), (2, 3), (2, 0), (1, 0)])
@pytest.mark.parametrize("n", [1, 5, 8, 128])
def test_reconstruction_structures_every_picture(efx, batch, mode, items, n):
    streams, golden = batch
    dec = efx.Decoder(n, P, P + 1, max_stream_bytes=sum(s.size for s in streams[:n]) + 4096)
    dec.set_option(efx.OPT_RECON_MODE, mode)
    dec.set_option(efx.OPT_RECON_ITEMS, items)
    dec.upload(streams[:n], efx.FORMAT_ES)
    dec.decode()
    assert np.array_equal(all_pictures(efx, dec, n), golden[:n])
    assert not any(dec.stream_status(i) for i in range(n))
    dec.close()


@pytest.mark.parametrize("groups,cap", [(1, 1), (1, 2), (2, 0), (4, 1), (16, 2), (0, 0)])
def test_groups_and_parse_cap_pinned(efx, batch, groups, cap):
    """The same call sequence under every group count / cap mode, back to back and one call at a time: the reference's double
    buffer must hold pictures 10 and 11 after every call."""
    streams, golden = batch
    n = len(streams)
    dec = efx.Decoder(n, P, 2, max_stream_bytes=sum(s.size for s in streams) + 4096)
    dec.set_option(efx.OPT_GROUPS, groups)
    dec.set_option(efx.OPT_PARSE_CAP, cap)
    assert dec.get_option(efx.OPT_GROUPS) == groups and dec.get_option(efx.OPT_PARSE_CAP) == cap
    dec.upload(streams, efx.FORMAT_ES)
    for sync in (False, False, True, False):
        dec.decode(sync=sync)
    dec.sync()
    h = dec.frame_hashes()
    for p in (P - 2, P - 1):
        assert np.array_equal(h[:, dec.picture_slot(p)], golden[:n, p])
    dec.close()


def test_decode_range_small_budgets_through_the_single_launch(efx, batch):
    """efx_decode_range asks for one or two pictures at a time (the streaming adapter): k_recon_all with n_pictures = 1, 2."""
    streams, golden = batch
    n = 16
    dec = efx.Decoder(n, P, P + 1, max_stream_bytes=sum(s.size for s in streams[:n]) + 4096)
    dec.upload(streams[:n], efx.FORMAT_ES)
    first = 0
    got = np.zeros((n, P), dtype=np.uint64)
    for budget in (1, 2, 1, 3, 5):
        dec.decode(first_picture=first, n_pictures=budget)
        h = dec.frame_hashes()
        for p in range(budget):
            got[:, first + p] = h[:n, dec.picture_slot(p)]
        first += budget
    assert first == P and np.array_equal(got, golden[:n])
    dec.close()


def test_staged_upload_never_writes_caller_memory(efx, batch):
    """Round-5 ADVICE (medium): efx_upload_streams used to choose the in-place path from the pointer pattern -- ONE stream
    lying in an arena always matched, and the library wrote its 9-byte tail and the zero fill behind it, i.e. over the first
    bytes of whatever the caller had packed there.  Streams packed back to back in an arena and uploaded one at a time through
    efx_upload_streams must leave every byte of the arena as it was."""
    streams, golden = batch
    dec = efx.Decoder(1, P, P + 1, max_stream_bytes=int(max(s.size for s in streams[:4])) + 4096)
    total = sum(int(s.size) for s in streams[:4])
    arena = dec.host_arena(total + 64)
    pos, at = 0, []
    for s_ in streams[:4]:
        arena[pos:pos + s_.size] = s_
        at.append(pos)
        pos += s_.size
    arena[pos:] = 0xA5
    before = arena.copy()
    import ctypes as C
    for k, s_ in enumerate(streams[:4]):
        ptrs = (C.c_void_p * 1)(arena.ctypes.data + at[k])
        lens = (C.c_size_t * 1)(int(s_.size))
        dec.upload_prepared((None, ptrs, lens, 1), efx.FORMAT_ES)
        dec.decode()
        assert np.array_equal(all_pictures(efx, dec, 1)[0], golden[k]), k
        assert np.array_equal(arena, before), k
    dec.close()


def test_in_place_ingest_equals_staged(efx, batch):
    """A batch laid out in a page-locked arena of the context (efx_host_alloc + efx_stream_layout) is transferred where it lies;
    the decoder must see exactly the bytes the staged path gives it -- ES and TS input -- and upload_done() must come true."""
    streams, golden = batch
    n = 64
    from espflix_amd import gen
    b = gen.Batch(0, n, P, 12, 0)
    for fmt, data in ((efx.FORMAT_ES, streams[:n]), (efx.FORMAT_TS, [b.ts(k) for k in range(n)])):
        total = sum(s.size for s in data)
        dec = efx.Decoder(n, P, P + 1, max_stream_bytes=total + 4096)
        arena = dec.host_arena(total + 32 * n + 4096)
        prep = dec.place_in_arena(arena, data)
        for rep in range(3):  # both bitstream buffers, and the first one again
            dec.upload_prepared(prep, fmt, in_place=True)
            dec.decode()
            assert dec.upload_done()
            assert np.array_equal(all_pictures(efx, dec, n), golden[:n]), (fmt, rep)
        es_in_place = [dec.es(k) for k in (0, 1, n - 1)]
        dec.upload(data, fmt)  # the staged path
        assert [dec.es(k) for k in (0, 1, n - 1)] == es_in_place
        # a batch that is NOT in layout order (two streams swapped) is refused in place, and decodes through the staged path
        swapped = list(prep[1])
        ptrs = type(prep[1])(*([swapped[1], swapped[0]] + swapped[2:]))
        lens = type(prep[2])(*([prep[2][1], prep[2][0]] + list(prep[2])[2:]))
        with pytest.raises(Exception):
            dec.upload_prepared((prep[0], ptrs, lens, n), fmt, in_place=True)
        dec.upload_prepared((prep[0], ptrs, lens, n), fmt)
        dec.decode()
        got = all_pictures(efx, dec, n)
        assert np.array_equal(got[0], golden[1]) and np.array_equal(got[1], golden[0]) and np.array_equal(got[2:], golden[2:n])
        dec.close()

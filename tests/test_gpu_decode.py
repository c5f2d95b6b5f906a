"""Parity tests proper: the HIP decode path (through the C-ABI, espflix_amd/libefx.so) against
the reference-derived golden vectors and against the CPU oracle on the same inputs."""
import numpy as np
import pytest

import common
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def efx():
    import espflix_amd
    espflix_amd.load_library()
    return espflix_amd


def gpu_hashes(efx, streams, fmt, max_pictures, ring_depth=None):
    dec = efx.Decoder(max_streams=len(streams), max_pictures=max_pictures,
                      ring_depth=ring_depth or max_pictures + 1, max_stream_bytes=sum(len(s) for s in streams) + 4096)
    dec.upload(streams, fmt)
    dec.decode()
    hashes = dec.frame_hashes()
    res = []
    for i in range(len(streams)):
        n = dec.picture_count(i)
        res.append(dict(n=n, status=dec.stream_status(i), hashes=[int(hashes[i, dec.picture_slot(p)]) for p in range(n)],
                        pts=[dec.picture_pts(i, p) for p in range(n)]))
    dec.close()
    return res


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_embedded_clips_vs_reference_golden(efx, clip, clips, golden):
    g = golden["clips"][clip]
    r = gpu_hashes(efx, [clips[clip]], efx.FORMAT_TS, 100)[0]
    assert r["status"] == 0 and r["n"] == len(g["hashes"])
    assert [f"{h:016x}" for h in r["hashes"]] == g["hashes"]
    assert r["pts"] == g["pts"]


def test_embedded_clip_es_input(efx, clips):
    es = oracle.ts_to_es(clips["vmedia"])
    r = gpu_hashes(efx, [es], efx.FORMAT_ES, 100)[0]
    n, h, _, _ = oracle.decode(es, 0)
    assert r["n"] == n and r["hashes"] == [int(x) for x in h] and r["pts"] == list(range(n))


@pytest.mark.parametrize("flags", common.SYN_FLAGS)
def test_synthetic_vs_reference_golden(efx, flags, golden):
    from espflix_amd import gen
    b = gen.Batch(0, 8, 12, 12, flags)
    res = gpu_hashes(efx, b.all_es(), efx.FORMAT_ES, 12)
    for k in common.SYN_IDS:
        assert res[k]["status"] == 0
        assert [f"{h:016x}" for h in res[k]["hashes"]] == golden["synthetic"][f"{flags}:{k}"]["hashes"]
    # the other five streams of the batch against the oracle
    for k in range(8):
        n, h, _, _ = oracle.decode(b.es(k), 0)
        assert res[k]["n"] == n and res[k]["hashes"] == [int(x) for x in h]


def test_handmade_stuffing_and_escapes_vs_reference_golden(efx, golden):
    """1 ... 300 macroblock_stuffing codes in front of a macroblock, address escapes, stuffing behind an escape
    (player.cpp:1267-1275): ES and TS input against what the unmodified reference decoded."""
    g = golden["handmade"]["stuffing"]
    es = common.stuffing_es()
    ts = common.one_pes_per_picture(es)
    r = gpu_hashes(efx, [np.frombuffer(es, dtype=np.uint8)] * 3, efx.FORMAT_ES, 8)
    for k in range(3):
        assert r[k]["status"] == 0 and [f"{h:016x}" for h in r[k]["hashes"]] == g["hashes"]
    r = gpu_hashes(efx, [np.frombuffer(ts, dtype=np.uint8)], efx.FORMAT_TS, 8)[0]
    assert r["status"] == 0 and [f"{h:016x}" for h in r["hashes"]] == g["hashes"] and r["pts"] == g["pts"]


def test_ts_input_synthetic_pts(efx, golden):
    from espflix_amd import gen
    b = gen.Batch(0, 8, 12, 12, 0)
    res = gpu_hashes(efx, [b.ts(k) for k in range(8)], efx.FORMAT_TS, 12)
    for k in common.SYN_IDS:
        g = golden["synthetic"][f"0:{k}"]
        assert [f"{h:016x}" for h in res[k]["hashes"]] == g["hashes"] and res[k]["pts"] == g["pts"]


def test_two_gops_sequence_header_repeated(efx):
    from espflix_amd import gen
    b = gen.Batch(40, 6, 24, 12, gen.FLAG_CUSTOM_MATRICES | gen.FLAG_WIDE_SLICES)
    res = gpu_hashes(efx, b.all_es(), efx.FORMAT_ES, 24)
    for k in range(6):
        n, h, _, _ = oracle.decode(b.es(k), 0)
        assert n == 24 and res[k]["n"] == 24 and res[k]["hashes"] == [int(x) for x in h]


def test_frame_bytes_match_oracle(efx):
    """Whole frames, not only hashes (and the device hash kernel agrees with the host FNV)."""
    from espflix_amd import gen
    b = gen.Batch(3, 2, 12, 12, 0)
    dec = efx.Decoder(2, 12, 13)
    dec.upload(b.all_es(), efx.FORMAT_ES)
    dec.decode()
    dev_hashes = dec.frame_hashes()
    for k in range(2):
        _, h, _, frames = oracle.decode(b.es(k), 0, want_frames=True)
        for p in (0, 1, 6, 11):
            got = dec.download_picture(k, p)
            assert np.array_equal(got, frames[p])
            assert oracle.fnv1a64(got) == int(dev_hashes[k, dec.picture_slot(p)]) == int(h[p])
    dec.close()


def test_double_buffer_ring_matches_full_ring(efx):
    """ring_depth = 2 (the reference's _fb[2]) leaves the same last two pictures as keeping all."""
    from espflix_amd import gen
    b = gen.Batch(10, 16, 12, 12, 0)
    full = gpu_hashes(efx, b.all_es(), efx.FORMAT_ES, 12, ring_depth=13)
    dec = efx.Decoder(16, 12, 2)
    dec.upload(b.all_es(), efx.FORMAT_ES)
    dec.decode()
    h = dec.frame_hashes()
    for k in range(16):
        assert int(h[k, dec.picture_slot(11)]) == full[k]["hashes"][11]
        assert int(h[k, dec.picture_slot(10)]) == full[k]["hashes"][10]
    dec.close()


def bench_golden(name, rows, pictures):
    """Per-picture frame hashes the unmodified reference decoder produced (tests/golden/make_bench_golden.py)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    return np.fromfile(path, dtype="<u8").reshape(rows, pictures)


def picture_table(dec, n_streams, n_pictures):
    """[stream][picture] frame hashes of the last decode of a context that keeps every picture."""
    h = dec.frame_hashes()
    return np.stack([h[:, dec.picture_slot(p)] for p in range(n_pictures)], axis=1)[:n_streams]


def test_config2_batch256_i_frames(efx):
    """BASELINE configs[1]: 256 I-frame-only streams, EVERY stream and picture against the reference decoder's
    output (bench_ionly.u64)."""
    from espflix_amd import gen
    want = bench_golden("bench_ionly.u64", 256, 8)
    b = gen.Batch(0, 256, 8, 12, gen.FLAG_I_ONLY)
    es = b.all_es()
    dec = efx.Decoder(256, 8, 9)
    dec.upload(es, efx.FORMAT_ES)
    dec.decode()
    assert all(dec.picture_count(i) == 8 and dec.stream_status(i) == 0 for i in range(256))
    assert np.array_equal(picture_table(dec, 256, 8), want)
    # I pictures do not depend on history: decoding the same batch again into a dirty ring is identical
    dec.erase_frames()
    dec.decode()
    assert np.array_equal(picture_table(dec, 256, 8), want)
    dec.close()


def test_config3_batch1024_gop12_every_stream(efx):
    """BASELINE configs[2] at full size: 1024 streams x GOP(12), EVERY stream and picture against the reference
    decoder's output (bench_gop12.u64); the decoder keeps going across calls (the ring rotates), and streams are
    independent (a permuted batch gives permuted results)."""
    from espflix_amd import gen
    want = bench_golden("bench_gop12.u64", 8192, 12)[:1024]
    b = gen.Batch(0, 1024, 12, 12, 0)
    es = b.all_es()
    dec = efx.Decoder(1024, 12, 13, max_stream_bytes=sum(e.size for e in es) + 65536)
    dec.upload(es, efx.FORMAT_ES)
    dec.decode()
    assert all(dec.picture_count(i) == 12 and dec.stream_status(i) == 0 for i in range(1024))
    slots1 = [dec.picture_slot(p) for p in range(12)]
    assert np.array_equal(picture_table(dec, 1024, 12), want)
    # the same GOP again: P pictures start from the I picture, the ring position moves on.  (1024 streams are one group;
    # the slices are short, so from now on the parse kernel's residency is capped: results do not depend on it)
    dec.set_timing(True)
    dec.decode()
    assert dec.timing().groups == 1
    assert [dec.picture_slot(p) for p in range(12)] == [(s + 12) % 13 for s in slots1]
    assert np.array_equal(picture_table(dec, 1024, 12), want)
    perm = np.random.default_rng(0).permutation(1024)
    dec.upload([es[j] for j in perm], efx.FORMAT_ES)
    dec.decode()
    assert np.array_equal(picture_table(dec, 1024, 12), want[perm])
    dec.close()


def test_groups_of_streams_with_transport_stream_input(efx):
    """A call that runs as two groups of streams (the second decode of 2048 short-slice streams: groups of 1024) gathers
    its per-stream results -- picture counts, status, PTS, ring positions -- from the hand-over slots of both groups."""
    from espflix_amd import gen
    N = 2048
    want = bench_golden("bench_gop12.u64", 8192, 12)[:N]
    b = gen.Batch(0, N, 12, 12, 0)
    ts = [b.ts(k) for k in range(N)]
    dec = efx.Decoder(N, 12, 13, max_stream_bytes=sum(len(t) for t in ts) + 65536)
    dec.upload(ts, efx.FORMAT_TS)
    dec.decode()
    dec.set_timing(True)
    dec.decode()
    assert dec.timing().groups == 2
    assert all(dec.picture_count(i) == 12 and dec.stream_status(i) == 0 for i in range(N))
    pts = [129003 + 3003 * f for f in range(12)]
    for i in (0, 1, 7, 511, 1023, 1024, 1025, 1777, 2047):  # (both sides of the group boundary)
        assert [dec.picture_pts(i, p) for p in range(12)] == pts, i
    h = dec.frame_hashes()
    got = np.stack([[h[i, dec.picture_slot(p, i)] for p in range(12)] for i in range(N)])
    assert np.array_equal(got, want)
    dec.close()


def test_real_stream_shape_every_stream(efx):
    """The service's stream shape (5 slices per picture over 2-3 macroblock rows, ~6.25 kB per picture): 256 streams
    against the reference decoder's output (bench_wide1500k.u64)."""
    from espflix_amd import gen
    want = bench_golden("bench_wide1500k.u64", 1024, 12)[:256]
    b = gen.Batch(0, 256, 12, 12, gen.FLAG_WIDE_SLICES | gen.FLAG_RATE_1500K)
    es = b.all_es()
    dec = efx.Decoder(256, 12, 13, max_stream_bytes=sum(e.size for e in es) + 65536)
    dec.upload(es, efx.FORMAT_ES)
    dec.decode()
    assert all(dec.picture_count(i) == 12 and dec.stream_status(i) == 0 for i in range(256))
    assert np.array_equal(picture_table(dec, 256, 12), want)
    dec.close()


def test_decoder_keeps_going_across_odd_chunks(efx):
    """ring_depth = 2 and uploads that end after an odd number of pictures, different for every stream: the frame
    index is per-stream state (MpegDecoder::_fb_index), not a property of the call."""
    from espflix_amd import gen
    b = gen.Batch(20, 3, 12, 12, 0)
    es = [b.es(k) for k in range(3)]
    offs = [b.picture_offsets(k) for k in range(3)]
    cuts = [(5, 8), (4, 9), (7, 10)]  # pictures per stream after the first and second upload
    want = [oracle.decode(e, 0)[1] for e in es]
    dec = efx.Decoder(3, 8, 2)
    bounds = [(0, c[0], c[1], 12) for c in cuts]
    last = [None] * 3
    for step in range(3):
        chunk = [es[k][offs[k][bounds[k][step]]:offs[k][bounds[k][step + 1]]] for k in range(3)]
        dec.upload(chunk, efx.FORMAT_ES)
        dec.decode()
        h = dec.frame_hashes()
        for k in range(3):
            n = dec.picture_count(k)
            assert n == bounds[k][step + 1] - bounds[k][step] and dec.stream_status(k) == 0
            first = bounds[k][step]
            for i in (n - 1, n - 2):  # the two pictures the double buffer holds
                if i >= 0:
                    assert int(h[k, dec.picture_slot(i, k)]) == int(want[k][first + i]), (step, k, i)
    dec.close()


def test_decode_from_walks_a_long_stream(efx):
    """A stream with more pictures than max_pictures: efx_decode_from(0), (5), (10) decode it in passes; the
    truncation is reported, the ring carries over."""
    from espflix_amd import gen
    b = gen.Batch(30, 2, 12, 12, 0)
    es = b.all_es()
    want = [oracle.decode(e, 0)[1] for e in es]
    dec = efx.Decoder(2, 5, 2)
    dec.upload(es, efx.FORMAT_ES)
    for first, n, trunc in ((0, 5, True), (5, 5, True), (10, 2, False)):
        dec.decode(first_picture=first)
        h = dec.frame_hashes()
        for k in range(2):
            assert dec.picture_count(k) == n
            assert bool(dec.stream_status(k) & efx.STREAM_TRUNCATED) == trunc
            assert int(h[k, dec.picture_slot(n - 1, k)]) == int(want[k][first + n - 1])
            assert int(h[k, dec.picture_slot(n - 2, k)]) == int(want[k][first + n - 2])
    dec.close()


@pytest.mark.parametrize("flags,n,pictures", [(64 | 128 | 4 | 2, 12, 24), (64 | 8 | 16, 16, 12), (128 | 8 | 4 | 2, 16, 24), (64 | 1, 8, 8),
                                              (256 | 64 | 8, 16, 12), (256 | 4 | 2 | 128, 12, 24)])
def test_quirk_flavours_combined_fresh_ids(efx, flags, n, pictures):
    """Escape-level forms / ignored picture types / user data combined with custom matrices, wide slices, long skips,
    flat bright areas and I-only streams, on ids beyond the golden set (two GOPs where 24 pictures): HIP path = oracle
    (which tests/test_oracle_vs_ref.py pins against the live reference on these flavours)."""
    from espflix_amd import gen
    b = gen.Batch(1000, n, pictures, 12, flags)
    res = gpu_hashes(efx, b.all_es(), efx.FORMAT_ES, pictures)
    for k in range(n):
        cnt, h, _, _ = oracle.decode(b.es(k), 0)
        assert res[k]["status"] == 0 and res[k]["n"] == cnt == pictures
        assert res[k]["hashes"] == [int(x) for x in h]


def test_decode_range_picture_budget(efx):
    """efx_decode_range: a call for at most n pictures per stream (fewer reconstruction launches) on a context sized for
    more; the rest is flagged and picked up by the next call -- passes of 1, 3, 8 pictures give the reference's frames."""
    from espflix_amd import gen
    b = gen.Batch(60, 3, 12, 12, gen.FLAG_WIDE_SLICES)
    es = b.all_es()
    want = [oracle.decode(e, 0)[1] for e in es]
    dec = efx.Decoder(3, 12, 13)
    dec.upload(es, efx.FORMAT_ES)
    first = 0
    for n in (1, 3, 8):
        dec.decode(first_picture=first, n_pictures=n)
        h = dec.frame_hashes()
        for k in range(3):
            assert dec.picture_count(k) == n
            assert bool(dec.stream_status(k) & efx.STREAM_TRUNCATED) == (first + n < 12)
            for p in range(n):
                assert int(h[k, dec.picture_slot(p, k)]) == int(want[k][first + p])
        first += n
    with pytest.raises(efx.EfxError):
        dec.decode(n_pictures=13)
    dec.close()


def test_no_buffer_swap_before_the_first_pts(efx):
    """flush_picture() neither pushes nor swaps while no PES PTS has been latched (player.cpp:692-702): pictures
    ahead of the first PTS are decoded over each other; the oracle (pinned against the reference on the same
    streams, tests/test_oracle_vs_ref.py) and the HIP path agree on every pushed frame, and on the slots."""
    from espflix_amd import gen
    b = gen.Batch(50, 2, 8, 12, 0)
    for k in range(2):
        es = b.es(k).tobytes()
        ts = np.frombuffer(common.late_pts_ts(es, first_with_pts=2 + k), dtype=np.uint8)
        n, hashes, pts, _ = oracle.decode(ts, 1, flush_last=True)
        assert n == 8 - (2 + k)  # the pictures ahead of the first PTS are never pushed
        dec = efx.Decoder(1, 8, 9)
        dec.upload([ts], efx.FORMAT_TS)
        dec.decode()
        assert dec.picture_count(0) == 8
        got_pts = [dec.picture_pts(0, p) for p in range(8)]
        assert got_pts[:2 + k] == [-1] * (2 + k) and got_pts[2 + k:] == [int(x) for x in pts]
        slots = [dec.picture_slot(p) for p in range(8)]
        assert slots[:3 + k] == [1] * (3 + k) and slots[3 + k:] == list(range(2, 7 - k))
        h = dec.frame_hashes()
        assert [int(h[0, slots[p]]) for p in range(2 + k, 8)] == [int(x) for x in hashes]
        dec.close()

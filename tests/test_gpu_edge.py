"""Edge cases of the batch API: empty / ragged / garbage / truncated inputs, capacity errors,
streams the reference leaves undefined (the HIP path and the oracle define the same result)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def efx():
    import espflix_amd
    espflix_amd.load_library()
    return espflix_amd


def run(efx, streams, max_pictures=12):
    dec = efx.Decoder(len(streams), max_pictures, max_pictures + 1, max_stream_bytes=sum(len(s) for s in streams) + 4096)
    dec.upload(streams, efx.FORMAT_ES)
    dec.decode()
    h = dec.frame_hashes()
    out = [(dec.picture_count(i), dec.stream_status(i), [int(h[i, dec.picture_slot(p)]) for p in range(dec.picture_count(i))])
           for i in range(len(streams))]
    dec.close()
    return out


def test_ragged_batch_with_empty_and_short_streams(efx):
    from espflix_amd import gen
    b = gen.Batch(0, 4, 12, 12, 0)
    offs1, offs2 = b.picture_offsets(1), b.picture_offsets(2)
    streams = [b.es(0), np.zeros(0, dtype=np.uint8), b.es(1)[:offs1[3]], np.zeros(777, dtype=np.uint8),
               b.es(2)[:offs2[1]], b.es(3)]
    res = run(efx, streams)
    assert [r[0] for r in res] == [12, 0, 3, 0, 1, 12]
    assert all(r[1] == 0 for r in res)
    for i in (0, 2, 4, 5):
        n, h, _, _ = oracle.decode(streams[i], 0)
        assert res[i][2] == [int(x) for x in h]


def test_garbage_streams_terminate(efx):
    rng = np.random.default_rng(5)
    streams = [rng.integers(0, 256, 20000, dtype=np.uint8) for _ in range(16)]
    res = run(efx, streams)          # must return; contents are undefined in the reference
    assert len(res) == 16


def test_too_many_pictures_is_flagged(efx):
    from espflix_amd import gen
    es = gen.Batch(0, 1, 12, 12, 0).es(0)
    n, st, hashes = run(efx, [es], max_pictures=5)[0]
    assert n == 5 and (st & efx.STREAM_TRUNCATED)
    _, h, _, _ = oracle.decode(es, 0)
    assert hashes == [int(x) for x in h[:5]]


def test_wrong_picture_size_is_rejected(efx):
    from espflix_amd import gen
    es = gen.Batch(0, 1, 2, 12, 0).es(0).copy()
    assert es[:4].tolist() == [0, 0, 1, 0xB3]
    es[4] = 0x14            # horizontal_size 0x140 = 320
    n, st, _ = run(efx, [es])[0]
    assert st & efx.STREAM_BAD_SIZE


def test_capacity_and_argument_errors(efx):
    from espflix_amd import gen
    b = gen.Batch(0, 3, 2, 12, 0)
    dec = efx.Decoder(2, 2, 2, max_stream_bytes=100000)
    with pytest.raises(efx.EfxError) as e:
        dec.upload(b.all_es(), efx.FORMAT_ES)          # 3 streams into a 2-stream context
    assert e.value.status == -4
    with pytest.raises(efx.EfxError) as e:
        dec.decode()                                    # nothing uploaded
    assert e.value.status == -5
    big = [np.zeros(200000, dtype=np.uint8)]
    with pytest.raises(efx.EfxError) as e:
        dec.upload(big, efx.FORMAT_ES)
    assert e.value.status == -4
    dec.upload(b.all_es()[:2], efx.FORMAT_ES)
    dec.decode()
    assert dec.picture_count(0) == 2
    with pytest.raises(efx.EfxError):
        dec.picture_count(5)
    dec.close()


def test_motion_vectors_outside_the_picture_are_clamped_like_the_oracle(efx):
    """Undefined in the reference (out-of-bounds reads); libefx and the oracle both clamp the
    source coordinates.  Build the case by patching a P picture's f_code so every vector doubles."""
    from espflix_amd import gen
    b = gen.Batch(1, 1, 4, 12, 0)          # stream id 1: forward_f_code = 2
    es = b.es(0).copy()
    offs = b.picture_offsets(0)
    hits = 0
    for p in range(1, 4):
        o = int(offs[p])
        assert es[o:o + 4].tolist() == [0, 0, 1, 0]
        # picture header: 10 + 3 + 16 bits, then full_pel_forward (bit 29) and f_code (bits 30-32)
        es[o + 4 + 3] |= 0x04               # set full_pel_forward: vectors are doubled
        hits += 1
    assert hits == 3
    n, st, hashes = run(efx, [es], max_pictures=4)[0]
    on, oh, _, _ = oracle.decode(es, 0)
    assert n == on == 4 and hashes == [int(x) for x in oh]


def test_decode_continues_across_uploads_with_double_buffer(efx):
    """ring_depth = 2: a second upload that starts with P pictures predicts from what the first
    left in the ring, like the reference decoder fed a file in pieces."""
    from espflix_amd import gen
    b = gen.Batch(2, 1, 12, 12, 0)
    es, offs = b.es(0), b.picture_offsets(0)
    _, h, _, _ = oracle.decode(es, 0)
    dec = efx.Decoder(1, 6, 2)
    dec.upload([es[:offs[6]]], efx.FORMAT_ES)
    dec.decode()
    assert dec.picture_count(0) == 6
    dec.upload([es[offs[6]:]], efx.FORMAT_ES)
    dec.decode()
    hh = dec.frame_hashes()
    # 6 pictures per call: picture 11 sits in the slot of "picture 5" of the second call
    assert int(hh[0, dec.picture_slot(5)]) == int(h[11])
    assert int(hh[0, dec.picture_slot(4)]) == int(h[10])
    dec.close()


def test_epoch_tags_wrap_under_back_to_back_decodes(efx):
    """Macroblock records carry an 8-bit epoch per hand-over slot; 800 pipelined efx_decode calls
    (3 slots, 2 parse streams) take every slot through the wrap-and-clear path."""
    from espflix_amd import gen
    b = gen.Batch(0, 3, 4, 12, 0)
    streams = b.all_es()
    dec = efx.Decoder(3, 4, 2)
    dec.upload(streams, efx.FORMAT_ES)
    dec.decode()
    want = dec.frame_hashes().copy()
    for i in range(800):
        dec.decode(sync=False)
        if i in (254, 255, 256, 511, 765, 766, 799):
            dec.sync()
            assert np.array_equal(dec.frame_hashes(), want), i
            assert all(dec.stream_status(k) == 0 and dec.picture_count(k) == 4 for k in range(3))
    dec.close()


def test_damaged_slice_that_runs_on_into_the_next_row(efx):
    """A bit flip can leave a slice syntactically valid but longer: it then decodes macroblocks of the row below,
    which the next slice decodes as well.  The reference (and the oracle) decode in bitstream order, so the later
    slice's macroblocks stay; here the two slices are parsed by two lanes at once, and the earlier one stops at
    the first macroblock of the later one (SliceDesc::mb_limit).  Two frame buffers, as in the reference:
    what a damaged slice leaves untouched is the picture before last."""
    from espflix_amd import gen
    b = gen.Batch(0, 8, 6)
    flips = [(7, 435, 2), (5, 6106, 4), (3, 7699, 3), (0, 41536, 3), (2, 22544, 7), (5, 8489, 5), (6, 11012, 3)]
    streams = []
    for k, pos, bit in flips:
        es = b.es(k).copy()
        es[pos] ^= 1 << bit
        streams.append(es)
    dec = efx.Decoder(len(streams), 8, 2, max_stream_bytes=sum(s.size for s in streams) + 4096)
    dec.upload(streams, efx.FORMAT_ES)
    dec.decode()
    h = dec.frame_hashes()
    for i, es in enumerate(streams):
        n, oh, _, _ = oracle.decode(es, 0, True)
        assert dec.picture_count(i) == n == 6, flips[i]
        assert [int(h[i, dec.picture_slot(p, i)]) for p in (n - 2, n - 1)] == [int(x) for x in oh[-2:]], flips[i]
    dec.close()


@pytest.mark.parametrize("ts_input", [False, True])
def test_heavily_damaged_streams_come_back(efx, ts_input):
    """Bit flips, overwritten and zeroed ranges, truncation and inserted start codes, as elementary and as transport
    streams: every decode returns (no hang, no fault), windows included, and what went wrong is in the status."""
    from espflix_amd import gen
    rng = np.random.default_rng(99)
    b = gen.Batch(0, 16, 6)
    base = [np.frombuffer(b.ts(k), dtype=np.uint8).copy() if ts_input else b.es(k).copy() for k in range(16)]
    blobs = []
    for i in range(640):
        x = base[i % 16].copy()
        kind = i % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 17))):
                x[int(rng.integers(0, x.size))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            a, n = int(rng.integers(0, x.size - 64)), int(rng.integers(1, 2000))
            x[a:a + n] = rng.integers(0, 256, min(n, x.size - a), dtype=np.uint8)
        elif kind == 2:
            x = x[:int(rng.integers(1, x.size))].copy()
        elif kind == 3:
            for _ in range(int(rng.integers(1, 9))):
                a = int(rng.integers(0, x.size - 8))
                x[a:a + 4] = [0, 0, 1, int(rng.integers(0, 256))]
        else:
            a, n = int(rng.integers(0, x.size - 64)), int(rng.integers(1, 4000))
            x[a:a + n] = 0
        blobs.append(x)
    dec = efx.Decoder(len(blobs), 16, 2, max_stream_bytes=sum(x.size for x in blobs) + 65536)
    dec.upload(blobs, efx.FORMAT_TS if ts_input else efx.FORMAT_ES)
    dec.decode(first_picture=3)
    dec.sync()
    dec.decode()
    status = np.array([dec.stream_status(i) for i in range(len(blobs))])
    counts = np.array([dec.picture_count(i) for i in range(len(blobs))])
    assert counts.max() <= 16 and (status != 0).sum() > len(blobs) // 2
    dec.close()


def test_bits_the_reference_marker_hunt_would_misread_are_flagged(efx):
    """The reference hunts for markers bit by bit and discards 24 bits without comparing them with 00 00 01
    (player.cpp:1360-1363): it passes an ignored picture header, user data or an extension only when what it walks
    through is zero bits or harmless 4-byte groups (that is how generator flavour 128 writes them: decoded bit-exact,
    status 0).  Anything else derails it onto phantom markers -- an extra flush_picture(), slices parsed from the
    middle of a header: the oracle, a bit-serial restatement, follows it there; the HIP path indexes byte-aligned
    start codes, cannot, and says so: EFX_STREAM_SERIAL_HUNT."""
    from espflix_amd import gen
    b = gen.Batch(0, 8, 12, 12, gen.FLAG_ODD_HEADERS)
    clean = [b.es(k) for k in range(8)]
    assert all(st == 0 for _, st, _ in run(efx, clean))

    es = clean[0].tobytes()
    user = es.index(b"\x00\x00\x01\xb2")
    user_end = es.index(b"\x00\x00\x01", user + 4)
    ext = es.index(b"\x00\x00\x01\xb5")
    # a picture header of an ignored type: 13 bits, then zero padding up to the next start code
    odd = next(o for o in _picture_headers(es) if ((es[o + 5] >> 3) & 7) not in (1, 2))
    real_b_header = bytearray(es)
    real_b_header[odd + 5] |= 0x07            # vbv_delay as a B picture carries it: the hunt reads 24 bits of it as a start code
    variants = {
        "user data of 4 n + 1 bytes": es[:user + 4] + b"Z" + es[user + 4:],
        "user data whose fourth byte is a slice code": es[:user + 7] + b"\x03" + es[user + 8:],
        "extension with a payload": es[:ext + 4] + b"\x12\x34" + es[ext + 4:],
        "B header with its real fields": bytes(real_b_header),
        "bytes ahead of the first start code": b"\x55\xAA\x01" + es,
    }
    harmless = {
        "zero bytes ahead of the first start code": b"\x00" * 7 + es,
        "zero stuffing after user data": es[:user_end] + b"\x00" * 5 + es[user_end:],
    }
    names = list(variants) + list(harmless)
    res = run(efx, [np.frombuffer(v, dtype=np.uint8) for v in list(variants.values()) + list(harmless.values())])
    for name, (n, st, hashes) in zip(names, res):
        if name in variants:
            assert st & efx.STREAM_SERIAL_HUNT, name
        else:
            assert st == 0, name
            _, h, _, _ = oracle.decode(np.frombuffer(harmless[name], dtype=np.uint8), 0)
            assert hashes == [int(x) for x in h], name
    # and the flag is not cosmetic: on the derailed streams the reference (its restatement) really decodes something else
    _, h0, _, _ = oracle.decode(clean[0], 0)
    n1, h1, _, _ = oracle.decode(np.frombuffer(variants["B header with its real fields"], dtype=np.uint8), 0)
    assert n1 != len(h0) or [int(x) for x in h1] != [int(x) for x in h0]


def _picture_headers(es: bytes):
    at = -1
    while True:
        at = es.find(b"\x00\x00\x01\x00", at + 1)
        if at < 0:
            return
        yield at


@pytest.mark.parametrize("n,pictures", [(1, 1), (1, 2), (2, 1), (1, 3), (1, 17), (3, 5)])
def test_tiny_batches_get_a_parse_lane_per_slice_slot(efx, n, pictures):
    """n x pictures x 16 slice slots not a multiple of 64: the k_parse grid is rounded UP (a truncating division left
    the last slots -- all of them below four pictures -- without a lane, and k_recon with stale records)."""
    from espflix_amd import gen
    b = gen.Batch(20, n, pictures, 12, 0)
    streams = b.all_es()
    dec = efx.Decoder(n, pictures, 2)
    for _ in range(2):
        dec.upload(streams, efx.FORMAT_ES)
        dec.decode()
        h = dec.frame_hashes()
        for i, es in enumerate(streams):
            cnt, oh, _, _ = oracle.decode(es, 0, True)
            assert dec.picture_count(i) == cnt == pictures and dec.stream_status(i) == 0
            assert int(h[i, dec.picture_slot(pictures - 1, i)]) == int(oh[-1]), (i, pictures)
            if pictures > 1:
                assert int(h[i, dec.picture_slot(pictures - 2, i)]) == int(oh[-2]), (i, pictures)
    dec.close()


def _units(es: np.ndarray):
    """Byte ranges of the start-code units of an elementary stream: [(start code value, first byte, end byte)]."""
    raw = es.tobytes()
    at, pos = [], raw.find(b"\x00\x00\x01")
    while pos >= 0:
        at.append(pos)
        pos = raw.find(b"\x00\x00\x01", pos + 3)
    return [(raw[p + 3], p, at[i + 1] if i + 1 < len(at) else len(raw)) for i, p in enumerate(at)]


def test_slices_out_of_raster_order_are_flagged(efx):
    """Complete slices in another order, or a row coded twice, decode to the same frames here and in the reference (which
    lets the later slice overwrite); the streams are flagged EFX_STREAM_SLICE_ORDER all the same, because the two decoders
    part ways as soon as such a slice is also damaged (include/efx.h).  A stream in raster order is not flagged."""
    from espflix_amd import gen
    es = gen.Batch(3, 1, 4, 12, 0).es(0)
    raw = es.tobytes()
    units = _units(es)
    pic = [i for i, (c, _, _) in enumerate(units) if c == 0x00][2]      # third picture (a P picture)
    sl = [i for i in range(pic + 1, len(units)) if 1 <= units[i][0] <= 0xAF][:12]
    assert [units[i][0] for i in sl] == list(range(1, 13))
    def rebuild(order):
        out = raw[:units[sl[0]][1]]
        for i in order:
            out += raw[units[i][1]:units[i][2]]
        return np.frombuffer(out + raw[units[sl[-1]][2]:], dtype=np.uint8)
    swapped = rebuild(sl[:4] + [sl[5], sl[4]] + sl[6:])                 # rows 5 and 6 change places
    twice = rebuild(sl[:8] + [sl[7]] + sl[8:])                          # row 8 coded twice
    res = run(efx, [es, swapped, twice], 4)
    want = [int(x) for x in oracle.decode(es, 0, True)[1]]
    for (n, st, h), name, flagged in zip(res, ("raster", "swapped", "twice"), (False, True, True)):
        assert n == 4 and h == want, name
        assert bool(st & efx.STREAM_SLICE_ORDER) == flagged and (st & ~efx.STREAM_SLICE_ORDER) == 0, (name, st)
    for s2 in (swapped, twice):
        assert [int(x) for x in oracle.decode(s2, 0, True)[1]] == want


def test_header_state_rules_at_their_edges(efx):
    """k_index applies MpegDecoder::marker()'s sequential rules to the whole unit list at once (ballot scans); the corners
    of those rules, each against the oracle: a sequence_end in the middle (the reference pauses there, player.cpp:1324-1327:
    nothing behind it is decoded), slices in front of the first picture header (the reference parses them with the P books and
    derails: dropped here and the stream flagged EFX_STREAM_SERIAL_HUNT), a sequence
    header of the wrong size after two good pictures (what was decoded stays, nothing after it is), a second sequence
    header + GOP in the middle (state carried on)."""
    from espflix_amd import gen
    es = gen.Batch(5, 1, 6, 12, 0).es(0)
    raw = es.tobytes()
    units = _units(es)
    pics = [i for i, (c, _, _) in enumerate(units) if c == 0x00]
    assert len(pics) == 6
    cut = units[pics[3]][1]                               # byte offset of the fourth picture's start code
    seq_hdr = raw[units[0][1]:units[1][1]]                # the sequence header unit
    ended = np.frombuffer(raw[:cut] + b"\x00\x00\x01\xB7" + raw[cut:], dtype=np.uint8)
    first_slice = units[pics[0] + 1]
    orphan = np.frombuffer(raw[:units[pics[0]][1]] + raw[first_slice[1]:first_slice[2]] + raw[units[pics[0]][1]:], dtype=np.uint8)
    bad = bytearray(seq_hdr)
    bad[4] = 0x14                                          # horizontal_size 320
    late_bad = np.frombuffer(raw[:units[pics[2]][1]] + bytes(bad) + raw[units[pics[2]][1]:], dtype=np.uint8)
    reseq = np.frombuffer(raw[:cut] + seq_hdr + raw[cut:], dtype=np.uint8)
    # the I picture cut out: the first picture is a P picture predicted from the (zeroed) frame store
    p_first = np.frombuffer(raw[:units[pics[0]][1]] + raw[units[pics[1]][1]:], dtype=np.uint8)
    streams = [es, ended, orphan, late_bad, reseq, p_first]
    res = run(efx, streams, 8)
    for (n, st, h), s, name in zip(res, streams, ("plain", "sequence_end", "orphan slice", "late bad size", "second sequence header",
                                                  "P picture first")):
        on, oh, _, _ = oracle.decode(s, 0, True)
        if name == "late bad size":
            # the reference goes on decoding slices into a frame of the wrong geometry (undefined); here and in the oracle the
            # stream is dead from that header on: the pictures before it are the reference's
            assert st & efx.STREAM_BAD_SIZE and h[:2] == [int(x) for x in oh[:2]], name
            continue
        if name == "orphan slice":
            assert st == efx.STREAM_SERIAL_HUNT and n == 6 and h == res[0][2], (name, st, n)
            assert on != 6  # (the flag is not cosmetic: the reference, one serial bit reader, ends up elsewhere)
            continue
        assert n == on and h == [int(x) for x in oh], (name, n, on)
        assert st == 0, (name, st)
    assert res[1][0] == 3 and res[2][0] == 6 and res[4][0] == 6 and res[5][0] == 5


def test_more_slices_than_slots_in_a_picture(efx):
    """A picture keeps kMaxSlicesPerPicture = 16 slice start codes; a seventeenth (here: rows coded again and again) is
    dropped and the stream flagged EFX_STREAM_TRUNCATED -- the first sixteen decode as ever."""
    from espflix_amd import gen
    es = gen.Batch(6, 1, 2, 12, 0).es(0)
    raw = es.tobytes()
    units = _units(es)
    pic = [i for i, (c, _, _) in enumerate(units) if c == 0x00][1]
    sl = [i for i in range(pic + 1, len(units)) if 1 <= units[i][0] <= 0xAF][:12]
    last = raw[units[sl[-1]][1]:units[sl[-1]][2]]
    many = np.frombuffer(raw[:units[sl[-1]][2]] + last * 6 + raw[units[sl[-1]][2]:], dtype=np.uint8)   # 18 slices
    (n, st, h), = run(efx, [many], 2)
    assert n == 2 and (st & efx.STREAM_TRUNCATED)
    assert h == [int(x) for x in oracle.decode(es, 0, True)[1]]   # (the repeated row decodes to the same pixels)


def test_structural_mutations_are_exact_or_flagged(efx):
    """96 streams, each with one structural mutation of its start-code units -- a unit deleted, duplicated, two units
    swapped, the stream cut at or inside a unit, a sequence_end inserted, a run of units deleted -- over four generator
    flavours, decoded with the reference's two-buffer ring: a stream whose status stays 0 must show the reference's last two
    pictures exactly (uncovered macroblocks keep the picture two back, missing pictures are simply absent ...); everything else
    must carry a status bit."""
    import re
    from espflix_amd import gen
    rng = np.random.default_rng(11)
    streams, what = [], []
    for k in range(96):
        fl = [0, 4, 128, 256][k % 4]
        raw = gen.Batch(100 + k, 1, 5, 12, fl).es(0).tobytes()
        at = [m.start() for m in re.finditer(b"\x00\x00\x01", raw)]
        parts = [raw[p:(at[i + 1] if i + 1 < len(at) else len(raw))] for i, p in enumerate(at)]
        codes = [raw[p + 3] for p in at]
        op, i = int(rng.integers(0, 7)), int(rng.integers(1, len(parts) - 1))
        if op == 0:
            del parts[i]
        elif op == 1:
            parts.insert(i, parts[i])
        elif op == 2:
            j = int(rng.integers(1, len(parts) - 1))
            parts[i], parts[j] = parts[j], parts[i]
        elif op == 3:
            parts = parts[:i] + [parts[i][:max(5, len(parts[i]) // 2)]]
        elif op == 4:
            parts.insert(i, b"\x00\x00\x01\xB7")
        elif op == 5:
            parts = parts[:i]
        else:
            del parts[i:i + int(rng.integers(2, 6))]
        streams.append(np.frombuffer(raw[:at[0]] + b"".join(parts), dtype=np.uint8))
        what.append(f"flavour {fl} op {op} at unit {codes[i]:02x}")
    dec = efx.Decoder(len(streams), 8, 2, max_stream_bytes=sum(len(s) for s in streams) + 8192)
    dec.upload(streams, efx.FORMAT_ES)
    dec.decode()
    h = dec.frame_hashes()
    clean = 0
    for i, s in enumerate(streams):
        n, st = dec.picture_count(i), dec.stream_status(i)
        if st:
            continue
        clean += 1
        on, oh, _, _ = oracle.decode(s, 0, True)
        if n == 0:
            assert on <= 1, what[i]   # (no picture at all: the reference's final flush_picture pushes one blank frame)
            continue
        got = [int(h[i, dec.picture_slot(p)]) for p in range(max(0, n - 2), n)]
        assert n == on and got == [int(x) for x in oh][max(0, on - 2):], (what[i], n, on)
    dec.close()
    assert clean >= 48   # (most mutations leave a stream the reference decodes without complaint)


def test_transport_packet_mutations_are_exact_or_flagged(efx):
    """96 transport streams (the generator's own multiplex and the hostile one of tests/common.py) with one packet-level
    mutation each -- a packet dropped, duplicated, two swapped, a sync byte broken, the stream cut inside a packet, the
    payload flag cleared, payload_unit_start set on a continuation packet: what reaches the video decoder is an elementary
    stream with a hole, a repeat or a bogus PES header in it.  Status 0 must mean the reference's pictures (two-buffer ring:
    the last two) AND its PTS; everything else must carry a status bit.  (Round 4 closed the three gaps this campaign found:
    a slice that stops at the next slice's first macroblock, a slice whose codes run through a start code, junk between a
    slice's end and the next start code are all flagged now.)"""
    import common
    from espflix_amd import gen
    rng = np.random.default_rng(21)
    streams = []
    for k in range(96):
        b = gen.Batch(300 + k, 1, 5, 12, [0, 4, 128, 256][k % 4])
        ts = bytes(b.ts(0)) if k % 3 == 0 else common.hostile_ts(b.es(0).tobytes(), 50 + k, noise=(k % 3 == 1))
        pk = [ts[i:i + 188] for i in range(0, len(ts) - 187, 188)]
        op, i = int(rng.integers(0, 7)), int(rng.integers(1, len(pk) - 1))
        if op == 0:
            del pk[i]
        elif op == 1:
            pk.insert(i, pk[i])
        elif op == 2:
            j = int(rng.integers(1, len(pk) - 1))
            pk[i], pk[j] = pk[j], pk[i]
        elif op == 3:
            pk[i] = b"\x48" + pk[i][1:]
        elif op == 4:
            pk = pk[:i] + [pk[i][:100]]
        elif op == 5:
            p = bytearray(pk[i]); p[3] &= ~0x10; pk[i] = bytes(p)
        else:
            p = bytearray(pk[i]); p[1] |= 0x40; pk[i] = bytes(p)
        streams.append(np.frombuffer(b"".join(pk), dtype=np.uint8))
    dec = efx.Decoder(len(streams), 8, 2, max_stream_bytes=sum(len(s) for s in streams) + 8192)
    dec.upload(streams, efx.FORMAT_TS)
    dec.decode()
    h = dec.frame_hashes()
    clean = 0
    for i, s in enumerate(streams):
        n, st = dec.picture_count(i), dec.stream_status(i)
        if st:
            continue
        clean += 1
        on, oh, opts, _ = oracle.decode(s, 1, True)
        if n == 0:
            assert on <= 1
            continue
        got = [int(h[i, dec.picture_slot(p)]) for p in range(max(0, n - 2), n)]
        assert n == on and got == [int(x) for x in oh][max(0, on - 2):], (i, n, on)
        assert [dec.picture_pts(i, p) for p in range(n)] == [int(x) for x in opts][:n], i
    dec.close()
    assert clean >= 15   # (most packet-level mutations damage a slice: flagged)

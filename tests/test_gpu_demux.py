"""transport-stream demultiplexing on the device (k_demux) against the
oracle's restatement of MpegDecoder::more/demux/parse_pts (player.cpp:294-307,381-493) and the
reference-derived goldens."""
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


def upload(efx, blobs, fmt, max_pictures=100, **kw):
    dec = efx.Decoder(max_streams=len(blobs), max_pictures=max_pictures, **kw)
    dec.upload(blobs, fmt)
    return dec


def check_es(dec, blobs):
    for i, ts in enumerate(blobs):
        want = oracle.ts_to_es(np.frombuffer(ts, dtype=np.uint8)).tobytes()
        got = dec.es(i)
        assert len(got) == len(want), (i, len(got), len(want))
        assert got == want, i


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_clip_es_bytes(efx, clips, clip):
    ts = clips[clip]
    dec = upload(efx, [ts], efx.FORMAT_TS)
    check_es(dec, [ts])


def test_synthetic_batch_es_equals_generator_es(efx):
    from espflix_amd import gen
    b = gen.Batch(0, 64, 12)
    blobs = [b.ts(k) for k in range(64)]
    dec = upload(efx, blobs, efx.FORMAT_TS, max_pictures=12)
    for k in range(64):
        assert dec.es(k) == b.es(k).tobytes(), k


def test_hostile_muxing_es_pts_and_frames(efx, clips):
    """Random payload sizes (1..184 bytes), PES boundaries around the picture start codes, PTS
    absent / PTS+DTS / header stuffing, interleaved null, audio and adaptation-only packets:
    ES bytes, per-picture PTS and decoded frames equal the oracle's."""
    from espflix_amd import gen
    b = gen.Batch(40, 6, 6)
    clip_es = oracle.ts_to_es(clips["vmedia"]).tobytes()
    es_list = [b.es(k).tobytes() for k in range(6)] + [clip_es[:common.picture_offsets(clip_es)[10]]]  # whole pictures
    blobs = [common.hostile_ts(es, 1000 + i) for i, es in enumerate(es_list)]
    dec = upload(efx, blobs, efx.FORMAT_TS, max_pictures=40)
    check_es(dec, blobs)
    dec.decode()
    for i, ts in enumerate(blobs):
        n, hashes, pts, _ = oracle.decode(np.frombuffer(ts, dtype=np.uint8), efx.FORMAT_TS, flush_last=True)
        assert dec.picture_count(i) == n
        assert [dec.picture_pts(i, p) for p in range(n)] == [int(x) for x in pts], i
        h = dec.frame_hashes()  # (double buffer: the last two pictures are resident)
        assert [int(h[i, dec.picture_slot(p, i)]) for p in (n - 2, n - 1)] == [int(x) for x in hashes[-2:]], i


def test_malformed_packets(efx):
    """Packets the reference would mis-handle are defined here exactly as in the oracle: lost sync
    -> one zero byte; no-payload flag, adaptation field swallowing the packet, PES header that
    does not fit -> nothing; a trailing partial packet is ignored."""
    rng = np.random.default_rng(7)
    es = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))
    good = common.packetize(es, [(0, 90000, False, 0), (2500, None, False, 2)], rng, noise=True)
    pk = [bytearray(good[i:i + 188]) for i in range(0, len(good), 188)]
    pk[3][0] = 0x48                                    # lost sync
    pk[5][3] = (pk[5][3] & 0xCF) | 0x20                # adaptation only: payload flag cleared
    pk.insert(7, bytearray(common.ts_packet(0x100, bytes(10), pusi=False)))
    pk[7][4] = 200                                     # adaptation_field_length beyond the packet
    short = bytearray(common.ts_packet(0x100, bytes(range(6)), pusi=True))  # 6 bytes cannot hold a PES header
    pk.insert(9, short)
    bad_prefix = bytearray(common.ts_packet(0x100, common.pes_header(12345) + bytes(20), pusi=True))
    bad_prefix[188 - 20 - 5] ^= 0x10                   # PTS marker nibble no longer matches the flags
    pk.insert(11, bad_prefix)
    blob = b"".join(bytes(p) for p in pk) + good[:100]  # + trailing partial packet
    tiny = [b"", good[:188], good[:187], bytes(188), bytes([0x47]) + bytes(187)]
    blobs = [blob] + tiny
    dec = upload(efx, blobs, efx.FORMAT_TS)
    check_es(dec, blobs)
    dec.decode()  # garbage ES: must terminate and report, not crash
    for i in range(len(blobs)):
        dec.stream_status(i)


def test_chunk_boundaries_and_tiny_payloads(efx):
    """> 128 packets per stream (k_demux works in chunks of 128) with 1..4-byte payloads: every
    ragged dword edge and zero-length packet case of the gather."""
    rng = np.random.default_rng(11)
    blobs = []
    for n_es in (300, 1000, 4097, 20000):
        es = bytes(rng.integers(0, 256, n_es, dtype=np.uint8))
        out = bytearray()
        pos = 0
        first = True
        while pos < len(es):
            n = int(rng.integers(0, 5)) if n_es < 5000 else int(rng.integers(0, 185))
            head = common.pes_header(int(rng.integers(0, 1 << 33))) if first else b""
            n = min(n, 184 - len(head), len(es) - pos)
            out += common.ts_packet(0x100, head + es[pos:pos + n], pusi=first)
            pos += n
            first = False
        blobs.append(bytes(out))
    dec = upload(efx, blobs, efx.FORMAT_TS)
    check_es(dec, blobs)


def test_es_download_for_es_input(efx):
    from espflix_amd import gen
    b = gen.Batch(3, 2, 2)
    dec = upload(efx, [b.es(0), b.es(1)], efx.FORMAT_ES, max_pictures=2)
    assert dec.es(0) == b.es(0).tobytes() and dec.es(1) == b.es(1).tobytes()


def test_demux_timing_reported(efx):
    from espflix_amd import gen
    b = gen.Batch(0, 32, 12)
    blobs = [b.ts(k) for k in range(32)]
    dec = efx.Decoder(max_streams=32, max_pictures=12)
    dec.set_timing(True)
    dec.upload(blobs, efx.FORMAT_TS)
    dec.decode()
    t = dec.timing()
    assert t.ts_bytes == sum(len(x) for x in blobs) and t.demux_ms > 0


def test_pts_survive_pipelined_decodes(efx, golden):
    """Per-picture PTS live in the hand-over slots: back-to-back efx_decode calls (three slots, two
    parse streams) report the same PTS and frames every time."""
    from espflix_amd import gen
    b = gen.Batch(0, 8, 12, 12, 0)
    blobs = [b.ts(k) for k in range(8)]
    dec = upload(efx, blobs, efx.FORMAT_TS, max_pictures=12, ring_depth=13)
    want = None
    for rounds in (1, 2, 3, 4, 7):
        for _ in range(rounds):
            dec.decode(sync=False)
        dec.sync()
        got = [[dec.picture_pts(i, p) for p in range(dec.picture_count(i))] for i in range(8)]
        hh = dec.frame_hashes()
        h = np.stack([hh[:, dec.picture_slot(p)] for p in range(12)], axis=1)  # the ring moves on with every call
        if want is None:
            want = (got, h)
            assert got[0] == golden["synthetic"]["0:0"]["pts"]
        assert got == want[0] and np.array_equal(h, want[1])


def test_pts_carry_with_uploads_running_ahead_of_the_reconstruction(efx):
    """A transport stream cut into uploads where the PES that carries a picture's PTS arrives one upload BEFORE the
    picture: every PES starts three bytes ahead of its picture's start code and its first transport packet carries just
    those three bytes, and every upload ends with that packet.  upload k, decode k, upload k + 1, decode k + 1 ... are
    queued back to back with a reconstruction stream that lags (a batch of long streams beside it): k_advance -- on the
    reconstruction stream -- must take the carried PTS from what k_index left in the hand-over slot, not from the upload
    buffers, which upload k + 2 is already overwriting (round-2 advisor finding).  Every picture carries the PTS the
    reference latches, frames included."""
    from espflix_amd import gen
    n_pic = 10
    es = gen.Batch(70, 1, n_pic, 12, 0).es(0).tobytes()
    offs = common.picture_offsets(es)
    starts = [0] + [o - 3 for o in offs[1:]] + [len(es)]
    first, rest = [], []   # per PES: its first packet (header + 3 payload bytes), the packets after it
    cc = 0
    for i in range(n_pic):
        chunk = es[starts[i]:starts[i + 1]]
        first.append(common.ts_packet(0x100, common.pes_header(500000 + 3003 * i) + chunk[:3], pusi=True, cc=cc))
        cc += 1
        pk = b""
        for o in range(3, len(chunk), 184):
            pk += common.ts_packet(0x100, chunk[o:o + 184], cc=cc)
            cc += 1
        rest.append(pk)
    ts = b"".join(f + r for f, r in zip(first, rest))
    ref_n, ref_h, ref_pts, _ = oracle.decode(np.frombuffer(ts, dtype=np.uint8), 1, flush_last=True)
    assert ref_n == n_pic and [int(x) for x in ref_pts] == [500000 + 3003 * i for i in range(n_pic)]
    # upload k: the packets of PES k behind its first one + the first packet of PES k + 1 (upload 0 also opens the stream)
    uploads = [np.frombuffer((first[0] if k == 0 else b"") + rest[k] + (first[k + 1] if k + 1 < n_pic else b""), dtype=np.uint8)
               for k in range(n_pic)]
    ballast = gen.Batch(71, 96, 12, 12, 0)   # keeps the GPU busy: the uploads run ahead of the reconstruction
    big = efx.Decoder(96, 12, 2)
    big.upload(ballast.all_es(), efx.FORMAT_ES)
    for attempt in range(3):
        dec = efx.Decoder(1, 12, 13, max_stream_bytes=len(ts) + 4096)
        got = []
        for k, up in enumerate(uploads):
            for _ in range(3):
                big.decode(sync=False)
            dec.upload([up], efx.FORMAT_TS)
            dec.decode(sync=False)
            if k % 3 == 2 or k + 1 == len(uploads):   # results are read only every third decode: uploads k + 1, k + 2 are
                assert dec.picture_count(0) == 1         # queued while decode k has not been reconstructed yet
                h = dec.frame_hashes()
                got.append((k, dec.picture_pts(0, 0), int(h[0, dec.picture_slot(0)])))
        assert dec.stream_state(0)[2] == int(ref_pts[-1])
        for k, pts, fh in got:
            assert pts == int(ref_pts[k]) and fh == int(ref_h[k]), (attempt, k)
        dec.close()
    big.close()


def test_one_pass_demux_equals_three_launches_and_the_oracle(efx, clips):
    """Round 6: the demultiplexer as ONE kernel -- a chunk's wave takes a ticket, stages its 16 packets once, publishes its totals and
    gets its base by decoupled look-back over the stream's chunk descriptors (k_demux_fused, EFX_OPT_DEMUX_FUSED; measured slower than
    scan / prefix / gather and not the default).  Both held against the oracle and against each other: a batch whose streams have 0, 1, 15,
    16, 17 packets, no whole packet at all, hundreds of chunks (the look-back window is 64 descriptors: longer than that loops),
    hostile muxing and both clips -- ES bytes, PTS lists (through the decoded pictures' PTS) and frames; uploaded five times over
    so that descriptors and tickets of earlier uploads are shown to be cleared."""
    from espflix_amd import gen
    rng = np.random.default_rng(5)
    b = gen.Batch(0, 8, 6)

    def packets(n):  # n packets of one random ES, PES header with a PTS in the first
        es = bytes(rng.integers(0, 256, max(0, 184 * n - 14), dtype=np.uint8))
        out, pos = bytearray(), 0
        for k in range(n):
            head = common.pes_header(int(rng.integers(0, 1 << 33))) if k == 0 else b""
            take = 184 - len(head)
            out += common.ts_packet(0x100, head + es[pos:pos + take], pusi=k == 0)
            pos += take
        return bytes(out)

    long_es = b.es(0).tobytes() * 24  # ~ 24 x 45 kB: > 5000 packets = > 320 chunks
    blobs = [b"", packets(1), packets(15), packets(16), packets(17), b"\x47" * 100, packets(33) + b"\x47\x01", b.ts(1).tobytes(),
             common.hostile_ts(b.es(2).tobytes(), 3), bytes(common.packetize(long_es, [(0, 129003, False, 0)], rng, noise=False)),
             clips["splash"].tobytes(), clips["vmedia"].tobytes()] + [b.ts(k).tobytes() for k in range(3, 8)]
    cap = sum(len(x) for x in blobs) + 64 * len(blobs) + 4096
    got = {}
    for three in (0, 1):  # the one-pass kernel first
        dec = efx.Decoder(max_streams=len(blobs), max_pictures=100, ring_depth=2, max_stream_bytes=cap)
        dec.set_option(efx.OPT_DEMUX_FUSED, 1 - three)
        for rep in range(5 if not three else 1):
            order = list(range(len(blobs))) if rep % 2 == 0 else list(range(len(blobs)))[::-1]
            dec.upload([blobs[i] for i in order], efx.FORMAT_TS)
            dec.decode()
            es = {order[j]: dec.es(j) for j in range(len(blobs))}
            pts = {order[j]: [dec.picture_pts(j, p) for p in range(dec.picture_count(j))] for j in range(len(blobs))}
            st = {order[j]: dec.stream_status(j) for j in range(len(blobs))}
            if (three, rep) == (0, 0):
                for i, ts in enumerate(blobs):
                    want = oracle.ts_to_es(np.frombuffer(ts, dtype=np.uint8)).tobytes()
                    assert es[i] == want, (i, len(es[i]), len(want))
                got = (es, pts, st)
            else:
                assert es == got[0] and pts == got[1] and st == got[2], (three, rep)
        dec.close()

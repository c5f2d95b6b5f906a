"""the trick-play index (indexer/indexer.cpp) built on the device for a batch
of transport streams, the video.idx file, the player's index arithmetic (espflix.cpp:589-627) and
a seek: decoding from the packet the index names."""
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


def titles(clips):
    return common.index_titles() + [("clips", [clips["vmedia"], clips["splash"], clips["vmedia"]])]


def test_index_file_and_queries(efx, clips, golden):
    for name, streams in titles(clips):
        dec = efx.Decoder(3, 1, 2, max_stream_bytes=sum(len(s) for s in streams) + 4096)
        res = dec.index_streams(streams, trick_speed=[1, 15, 15])
        for (rec, samples), ts in zip(res, streams):
            first, last, sp, so = oracle.ts_sequences(ts)
            assert (rec["first_pts"], rec["last_pts"]) == (first, last) and rec["sample_count"] == len(samples) > 0
        idx = efx.idx_build([r for r, _ in res], [s for _, s in res])
        want = oracle.make_idx(streams)
        assert idx == want, name
        assert f"{oracle.fnv1a64(oracle.idx_masked(idx)):016x}" == golden["index"][name]
        first, last = np.frombuffer(idx[8:24], dtype=np.int64)
        for pts, speed in common.index_queries(int(first), int(last)):
            assert (efx.idx_pts2offset(idx, pts, speed), efx.idx_pts2pts(idx, pts, speed)) == oracle.idx_query(idx, pts, speed)
        dec.close()


def test_seek_decodes_from_indexed_packet(efx):
    """Random access: the sample the player would fetch for a PTS names a packet; uploading the
    transport stream from there decodes the same pictures as the tail of the full decode."""
    name, streams = common.index_titles()[0]
    main = streams[0]
    dec = efx.Decoder(1, 100, 101, max_stream_bytes=len(main) + 4096)
    (rec, samples), = dec.index_streams([main])
    idx = efx.idx_build([rec, rec, rec], [samples, samples, samples])
    dec.upload([main], efx.FORMAT_TS)
    dec.decode()
    n_full = dec.picture_count(0)
    full = [(dec.picture_pts(0, p), int(dec.frame_hashes()[0, dec.picture_slot(p)])) for p in range(n_full)]
    target = rec["first_pts"] + 5 * 12 * 3003 + 1500   # inside the sixth GOP
    off = efx.idx_pts2offset(idx, target, 0)
    packet = int(np.frombuffer(idx[off:off + 4], dtype=np.uint32)[0])
    assert packet > 0 and packet * 188 < len(main)
    dec.reset()
    dec.upload([main[packet * 188:]], efx.FORMAT_TS)
    dec.decode()
    n_tail = dec.picture_count(0)
    tail = [(dec.picture_pts(0, p), int(dec.frame_hashes()[0, dec.picture_slot(p)])) for p in range(n_tail)]
    assert 0 < n_tail < n_full and n_tail % 12 == 0
    assert tail == full[n_full - n_tail:]
    assert abs(tail[0][0] - target) <= 6 * 3003 + 3003  # nearest sequence start
    dec.close()


def test_unsorted_and_degenerate_streams(efx):
    """PTS values that go backwards force the reference's linear first-minimum scan; streams without
    a sequence header or without packets yield an empty record."""
    rng = np.random.default_rng(3)
    seqhdr = bytes([0, 0, 1, 0xB3]) + bytes(60)
    pic = bytes([0, 0, 1, 0x00]) + bytes(60)
    def ts_of(pts_list, kinds):
        out = bytearray()
        for pts, k in zip(pts_list, kinds):
            out += common.ts_packet(0x100, common.pes_header(pts) + (seqhdr if k else pic), pusi=True)
            out += common.ts_packet(0x100, bytes(100))
        return bytes(out)
    pts = [int(x) for x in rng.integers(100000, 400000, 60)]          # unsorted
    dup = sorted(pts[:30]) + [sorted(pts[:30])[-1]] * 5               # sorted with duplicates
    streams = [ts_of(pts, [1] * 60), ts_of(dup, [i % 3 != 1 for i in range(len(dup))]), ts_of(pts[:5], [0] * 5), b"",
               ts_of([90000, 90000 + (1 << 31) + 5000, 90000 + (1 << 32)], [1, 1, 1])]
    dec = efx.Decoder(len(streams), 1, 2)
    res = dec.index_streams(streams, samples_cap=700000)
    for i, ((rec, samples), ts) in enumerate(zip(res, streams)):
        first, last, sp, so = oracle.ts_sequences(np.frombuffer(ts, dtype=np.uint8))
        assert (rec["first_pts"], rec["last_pts"]) == (first, last), i
        if len(sp) == 0 or last < first:
            assert rec["sample_count"] == 0
            continue
        want = oracle.make_idx([np.frombuffer(ts, dtype=np.uint8)] * 3)
        cnt = int(np.frombuffer(want[8 + 24:8 + 28], dtype=np.uint32)[0])
        assert rec["sample_count"] == cnt, i
        assert np.array_equal(samples, np.frombuffer(want[104:104 + 4 * cnt], dtype=np.uint32)), i
    dec.close()

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


def test_config2_batch256_i_frames(efx):
    """BASELINE configs[1]: 256 I-frame-only streams; a sample against the oracle, all of them
    through size-independent properties."""
    from espflix_amd import gen
    b = gen.Batch(0, 256, 8, 12, gen.FLAG_I_ONLY)
    es = b.all_es()
    res = gpu_hashes(efx, es, efx.FORMAT_ES, 8)
    assert all(r["n"] == 8 and r["status"] == 0 for r in res)
    for k in range(0, 256, 17):
        _, h, _, _ = oracle.decode(es[k], 0)
        assert res[k]["hashes"] == [int(x) for x in h]
    # I pictures do not depend on history: decoding the same batch into a dirty ring is identical
    dec = efx.Decoder(256, 8, 9)
    dec.erase_frames()
    dec.upload(es, efx.FORMAT_ES)
    dec.decode()
    h2 = dec.frame_hashes()
    for k in range(256):
        assert [int(h2[k, dec.picture_slot(p)]) for p in range(8)] == res[k]["hashes"]
    dec.close()


def test_config3_batch1024_gop12_properties(efx):
    """BASELINE configs[2] at full size: 1024 streams x GOP(12).  Oracle on a spread sample;
    determinism (two decodes, and a decode after a permuted upload) on all of them."""
    from espflix_amd import gen
    b = gen.Batch(0, 1024, 12, 12, 0)
    es = b.all_es()
    dec = efx.Decoder(1024, 12, 13, max_stream_bytes=sum(e.size for e in es) + 65536)
    dec.upload(es, efx.FORMAT_ES)
    dec.decode()
    h1 = dec.frame_hashes().copy()
    assert all(dec.picture_count(i) == 12 and dec.stream_status(i) == 0 for i in range(1024))
    dec.decode()                                   # P pictures start from the I picture: idempotent
    assert np.array_equal(h1, dec.frame_hashes())
    for k in range(5, 1024, 97):
        _, h, _, _ = oracle.decode(es[k], 0)
        assert [int(h1[k, dec.picture_slot(p)]) for p in range(12)] == [int(x) for x in h]
    # streams are independent: a permuted batch gives permuted results
    perm = np.random.default_rng(0).permutation(1024)
    dec.upload([es[j] for j in perm], efx.FORMAT_ES)
    dec.decode()
    h3 = dec.frame_hashes()
    assert np.array_equal(h3, h1[perm])
    dec.close()

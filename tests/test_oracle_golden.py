"""The CPU restatement (oracle/efx_oracle.c) against golden vectors produced by the unmodified
reference (tests/golden/golden.json, made by tests/golden/make_golden.py).  Runs anywhere."""
import ctypes

import numpy as np
import pytest

import common
import oracle
from espflix_amd import gen


def hx(h):
    return [f"{int(x):016x}" for x in h]


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_embedded_clip_ts(clip, clips, golden):
    g = golden["clips"][clip]
    n, h, pts, _ = oracle.decode(clips[clip], 1, flush_last=True)
    assert n == len(g["hashes"])
    assert hx(h) == g["hashes"]
    assert [int(p) for p in pts] == g["pts"]
    # without the final flush_picture(1) the last picture is never pushed (player.cpp:692-702)
    n2, h2, _, _ = oracle.decode(clips[clip], 1, flush_last=False)
    assert n2 == g["pushed_without_flush"]
    assert f"{oracle.chain_hash(h2):016x}" == g["chain"]


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_embedded_clip_es_equals_ts(clip, clips):
    """Feeding the demultiplexed ES gives the same frames (pts become picture indices)."""
    es = oracle.ts_to_es(clips[clip])
    n1, h1, _, _ = oracle.decode(clips[clip], 1)
    n2, h2, pts2, _ = oracle.decode(es, 0)
    assert n1 == n2 and (h1 == h2).all()
    assert list(pts2) == list(range(n2))


@pytest.mark.parametrize("flags", common.SYN_FLAGS)
def test_synthetic_streams(flags, golden):
    b = gen.Batch(0, 8, 12, 12, flags)
    for k in common.SYN_IDS:
        g = golden["synthetic"][f"{flags}:{k}"]
        assert f"{common.fnv_bytes(b.es(k)):016x}" == g["es_fnv"], "generator is not deterministic"
        n, h, pts, _ = oracle.decode(b.ts(k), 1)
        assert hx(h) == g["hashes"] and [int(p) for p in pts] == g["pts"]
        n2, h2, _, _ = oracle.decode(b.es(k), 0)
        assert hx(h2) == g["hashes"]


def test_handmade_stuffing_and_escapes(golden):
    """macroblock_stuffing 1 ... 300 times in front of a macroblock, address escapes, a stuffing code behind an escape
    (= increment 34), written bit by bit (common.stuffing_es); goldens from the unmodified reference."""
    g = golden["handmade"]["stuffing"]
    es = np.frombuffer(common.stuffing_es(), dtype=np.uint8)
    assert f"{common.fnv_bytes(es):016x}" == g["es_fnv"], "the hand-built stream changed: regenerate the goldens"
    n, h, pts, _ = oracle.decode(np.frombuffer(common.one_pes_per_picture(es.tobytes()), dtype=np.uint8), 1)
    assert hx(h) == g["hashes"] and [int(p) for p in pts] == g["pts"]
    n2, h2, _, _ = oracle.decode(es, 0)
    assert hx(h2) == g["hashes"]


def test_fixture_coverage_escape_levels_and_header_quirks(clips):
    """What the two quirk flavours are FOR, checked on the oracle's parse trace (so that a change of the generator
    cannot quietly empty the fixture): flavour 64 reaches every form of the escape level of player.cpp:1092-1099 --
    "xx", "00 xx" (128..255, also small levels and 0 the long way) and "80 xx" (-256..-129, also -128..-1) -- and zero
    runs beyond 31; flavour 128 carries B / D / forbidden / reserved picture types whose slices decode with the P
    books (player.cpp:710-717,1292) under f_code and full_pel values that change from one real P header to the next."""
    b = gen.Batch(0, 8, 12, 12, gen.FLAG_HUGE_LEVELS)
    for k in common.SYN_IDS:
        t = oracle.trace_levels(b.es(k), 0)
        assert t["max"] == 255 and t["min"] == -256 and t["m256"] > 0 and t["zero"] > 0 and t["wide"] > 300
        assert min(t["esc_forms"]) > 1000 and t["esc_small_long"] > 1000 and t["esc_max_run"] == 63
        assert t["abandoned"] == 0 and t["bad"] == 0
        # ... without ever emulating a start code inside a slice (the long forms of small levels can: the generator
        # steers around it), so that the stream means the same to a decoder that scans for start codes
        assert b.es(k).tobytes().count(b"\x00\x00\x01") == 12 * 13 + 2
    # the other flavours and both clips stay inside -127..127: without flavour 64 the 16-bit forms are never decoded
    assert oracle.trace_levels(gen.Batch(0, 1, 12, 12, 0).es(0), 0)["esc_forms"][1:] == [0, 0]
    b = gen.Batch(0, 8, 12, 12, gen.FLAG_ODD_HEADERS)
    types, r_sizes, full = set(), set(), set()
    for k in range(8):
        es = b.es(k)
        t = oracle.trace_levels(es, 0)
        types |= t["pic_types"]
        r_sizes |= t["r_sizes"]
        full |= t["full_pel"]
        assert t["bad"] == 0
        raw = es.tobytes()
        assert raw.count(b"\x00\x00\x01\xb2") > 0 and raw.count(b"\x00\x00\x01\xb5") > 0
    assert types == {0, 1, 2, 3, 4, 7} and r_sizes == {0, 1} and full == {0, 1}
    # flavour 256: extra_bit_slice = 1 + information bytes (player.cpp:1261-1262) -- one, two and three of them, in I and
    # in P pictures, in the 12-slice and in the 5-slice shape; no other fixture (nor either clip) has the bit set
    for fl, per_pic in ((gen.FLAG_SLICE_EXTRA, 12), (gen.FLAG_SLICE_EXTRA | gen.FLAG_WIDE_SLICES, 5)):
        seen = {}
        for k in common.SYN_IDS:
            t = oracle.trace_levels(gen.Batch(0, 8, 12, 12, fl).es(k), 0)
            assert t["bad"] == 0 and t["slices"] == 12 * per_pic
            for key, n in t["slice_extra"].items():
                seen[key] = seen.get(key, 0) + n
        assert {(1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3)} <= set(seen) and sum(seen.values()) > 12 * per_pic
    assert oracle.trace_levels(gen.Batch(0, 1, 12, 12, 0).es(0), 0)["slice_extra"] == {}
    for clip in ("splash", "vmedia"):
        assert oracle.trace_levels(clips[clip], 1)["slice_extra"] == {}


def test_composite_fields(golden):
    _, _, _, frames = oracle.decode(gen.Batch(0, 1, 12, 12, 0).ts(0), 1, want_frames=True)
    inputs = {"lcg": common.lcg_frames(), "random": common.random_frames(7),
              "decoded": np.concatenate([frames[10], frames[11]])}
    for name, fr in inputs.items():
        for ntsc in (True, False):
            f = oracle.video_field(fr, ntsc, 0, 3)
            got = [f"{common.fnv_bytes(f[i]):016x}" for i in range(3)]
            assert got == golden["composite"][f"{name}:{'ntsc' if ntsc else 'pal'}"]


def test_pdm(golden):
    pcm = common.pdm_pcm(0, 40)
    st = np.zeros(3, dtype=np.int32)
    beep = ctypes.c_int(0)
    words = []
    for c in range(40):
        if c == 3:
            beep.value = 5
        silent = (c % 7) == 6
        words.append(oracle.write_pcm_16(st, beep, None if silent else pcm[c * 128:(c + 1) * 128]))
    assert f"{common.fnv_bytes(np.concatenate(words)):016x}" == golden["pdm"]["sine220_silence7_beep3"]
    st = np.zeros(3, dtype=np.int32)
    assert f"{common.fnv_bytes(oracle.pdm(st, pcm)):016x}" == golden["pdm"]["sine220"]
    # state carries across calls: one long call == many short calls
    st2 = np.zeros(3, dtype=np.int32)
    parts = [oracle.pdm(st2, pcm[i:i + 128]) for i in range(0, pcm.size, 128)]
    assert np.array_equal(np.concatenate(parts), oracle.pdm(np.zeros(3, dtype=np.int32), pcm))
    assert np.array_equal(st, st2)


def test_tables(golden):
    zz, pm = oracle.tables()
    assert list(zz) == golden["tables"]["zig_zag"]
    assert list(pm) == golden["tables"]["scale_dct_q"]
    for ntsc in (True, False):
        key = "ntsc" if ntsc else "pal"
        assert list(oracle.video_params(ntsc)) == golden["tables"]["params_" + key]
        assert f"{common.fnv_bytes(oracle.color_tab(ntsc)):016x}" == golden["tables"]["color_tab_" + key]


def test_empty_and_garbage_inputs():
    # no picture start code -> nothing is ever pushed (flush_picture(1) alone would push the
    # untouched frame buffer, as load_poster does)
    assert oracle.decode(np.zeros(0, dtype=np.uint8), 0, flush_last=False)[0] == 0
    assert oracle.decode(np.zeros(1000, dtype=np.uint8), 0, flush_last=False)[0] == 0
    assert oracle.decode(np.zeros(0, dtype=np.uint8), 0, flush_last=True)[0] == 1
    rng = np.random.default_rng(3)
    for _ in range(20):  # must terminate and not crash on noise
        oracle.decode(rng.integers(0, 256, 5000, dtype=np.uint8), 0)
        oracle.decode(rng.integers(0, 256, 188 * 20, dtype=np.uint8), 1)


def test_truncated_stream_stops_cleanly():
    b = gen.Batch(0, 1, 12, 12, 0)
    es, offs = b.es(0), b.picture_offsets(0)
    full = oracle.decode(es, 0)[1]
    # cut at picture boundaries: exactly the pictures before the cut come out, unchanged
    for k in (1, 5, 11):
        n, h, _, _ = oracle.decode(es[:offs[k]], 0)
        assert n == k and (h == full[:k]).all()
    # cut inside a slice: the decoder must terminate; pictures well before the cut are unaffected
    # (the damaged tail may resynchronise on a phantom start code, as in the reference)
    for cut in (100, 5000, es.size // 2, es.size - 3):
        n, h, _, _ = oracle.decode(es[:cut], 0)
        k = int(np.searchsorted(offs, cut, side="right")) - 1   # pictures wholly before the cut
        assert n >= k and (h[:k] == full[:k]).all()


def test_display_state_fields(golden):
    """_hscroll slides and the composite() overlay / progress bar against the reference goldens."""
    disp = np.minimum(common.random_frames(77), 248)
    for name, front, hs, ov_seed, blend, progress in common.DISPLAY_CASES:
        n = len(hs) if hs is not None else 6
        ov = common.overlay_bytes(ov_seed) if ov_seed is not None else None
        for ntsc in (True, False):
            f = oracle.video_field_ex(disp, ntsc, 0, n, front, hs, ov, blend, progress)
            got = [f"{common.fnv_bytes(f[i]):016x}" for i in range(n)]
            assert got == golden["display"][f"{name}:{'ntsc' if ntsc else 'pal'}"], name


def test_sbc_pcm_and_tables(golden, clips):
    syn, pro = oracle.sbc_tables()
    assert f"{common.fnv_bytes(syn):016x}" == golden["tables"]["sbc_syn_8"]
    assert f"{common.fnv_bytes(pro):016x}" == golden["tables"]["sbc_proto_8"]
    for name, kw, n, probe in common.SBC_CASES:
        fr = common.sbc_frames(common.seed_of(name), n, **kw)
        fb = common.sbc_frame_bytes(kw["blocks"], 1 if kw["mode"] == 0 else 2, kw["bitpool"])
        pcm, _ = oracle.sbc_decode(fr, fb, probe)
        assert f"{common.fnv_bytes(pcm):016x}" == golden["sbc"][name], name
    for clip in ("splash", "vmedia"):
        g = golden["sbc"]["clip:" + clip]
        es = oracle.ts_audio_es(clips[clip])
        fb = common.CLIP_SBC_FRAME_BYTES[clip]
        assert es.size // fb == g["frames"] and f"{common.fnv_bytes(es):016x}" == g["audio_es_fnv"]
        pcm, _ = oracle.sbc_decode(es[:es.size // fb * fb], fb, True)
        assert f"{common.fnv_bytes(pcm):016x}" == g["pcm_fnv"]


def test_trick_play_index(golden, clips):
    for name, streams in common.index_titles() + [("clips", [clips["vmedia"], clips["splash"], clips["vmedia"]])]:
        idx = oracle.make_idx(streams)
        assert f"{oracle.fnv1a64(oracle.idx_masked(idx)):016x}" == golden["index"][name], name

"""Pins the CPU restatement against the reference ITSELF, run live: the unmodified reference
sources compiled into oracle/_ref by `make ref`.  Skipped where those binaries are absent."""
import ctypes

import numpy as np
import pytest

import common
import oracle
from espflix_amd import gen

pytestmark = pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_clips_frame_by_frame(clip, clips):
    rh, rpts, rframes = oracle.ref_decode(clips[clip], flush_last=True, want_frames=True)
    n, h, pts, frames = oracle.decode(clips[clip], 1, want_frames=True)
    assert n == len(rh) and (h == rh).all() and (pts == rpts).all()
    assert np.array_equal(frames, rframes)


@pytest.mark.parametrize("flags", [0, 2, 4, 8, 16, 2 | 4 | 16])
def test_fresh_synthetic_streams(flags):
    """Stream ids beyond the committed golden set."""
    b = gen.Batch(100, 6, 24, 12, flags)   # two GOPs: sequence header repeated mid-stream
    for k in range(6):
        rh, rpts, _ = oracle.ref_decode(b.ts(k))
        n, h, pts, _ = oracle.decode(b.ts(k), 1)
        assert n == len(rh) == 24 and (h == rh).all() and (pts == rpts).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_composite_random_frames(seed):
    fr = common.random_frames(seed)
    for ntsc in (True, False):
        assert np.array_equal(oracle.video_field(fr, ntsc, 0, 2), oracle.ref_video_field(fr, ntsc, 2))


def test_video_tables():
    for ntsc in (True, False):
        params, ctab, dither = oracle.ref_video_params(ntsc)
        assert np.array_equal(params, oracle.video_params(ntsc))
        assert np.array_equal(ctab, oracle.color_tab(ntsc))


@pytest.mark.parametrize("k", [0, 5])
def test_pdm_random(k):
    pcm = common.pdm_pcm(k, 30)
    ref = oracle.ref_pdm(pcm, silence_every=5, beep_at=2)
    st = np.zeros(3, dtype=np.int32)
    beep = ctypes.c_int(0)
    words = []
    for c in range(30):
        if c == 2:
            beep.value = 5
        words.append(oracle.write_pcm_16(st, beep, None if (c % 5) == 4 else pcm[c * 128:(c + 1) * 128]))
    assert np.array_equal(np.concatenate(words), ref)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_hostile_muxing_pts_latch(seed, clips):
    """PES boundaries a few bytes either side of the picture start codes, PTS absent / with DTS /
    header stuffing, random payload sizes, interleaved null / audio / adaptation-only packets: the
    restatement's demux and its PTS latch (look-ahead of the bit reader) follow the reference."""
    b = gen.Batch(200 + seed, 2, 8)
    clip_es = oracle.ts_to_es(clips["vmedia"]).tobytes()
    es_list = [b.es(0).tobytes(), b.es(1).tobytes(), clip_es[:common.picture_offsets(clip_es)[10]]]  # whole pictures
    for i, es in enumerate(es_list):
        ts = np.frombuffer(common.hostile_ts(es, 10 * seed + i), dtype=np.uint8)
        rh, rpts, _ = oracle.ref_decode(ts)
        n, h, pts, _ = oracle.decode(ts, 1)
        assert n == len(rh) and (h == rh).all(), (seed, i)
        assert (pts == rpts).all(), (seed, i, pts.tolist(), rpts.tolist())


@pytest.mark.parametrize("ntsc", [True, False])
@pytest.mark.parametrize("case", common.DISPLAY_CASES, ids=[c[0] for c in common.DISPLAY_CASES])
def test_display_state_hscroll_and_overlay(case, ntsc):
    """The two-frame slide (_hscroll) and the overlay / progress bar (composite()) of video_isr."""
    name, front, hs, ov_seed, blend, progress = case
    fr = common.random_frames(77)
    fr = np.minimum(fr, 248)
    n = len(hs) if hs is not None else 6
    ov = common.overlay_bytes(ov_seed) if ov_seed is not None else None
    ref = oracle.ref_video_field_ex(fr, ntsc, n, front, hs, ov if ov is not None else (np.zeros(1280, np.uint8) if blend else None),
                                    blend, progress)
    got = oracle.video_field_ex(fr, ntsc, 0, n, front, hs, ov, blend, progress)
    assert np.array_equal(ref, got)

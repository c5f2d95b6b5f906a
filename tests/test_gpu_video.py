"""Composite-video field and PDM kernels against the reference-derived goldens and the oracle.
The bar for the composite line is +-1 LSB of the DAC byte; the path is all-integer so the tests
demand exact equality."""
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


def gpu_fields(efx, dec, frames2, ntsc, nfields):
    """Fields for frame_counter 0..nfields-1 of the front frame (Frame[0])."""
    dec.upload_frame(0, 0, frames2[:efx.FRAME_BYTES])
    vp = efx.video_params(ntsc)
    n = vp["line_width"] * vp["line_count"]
    dst = dec.alloc(n * 2)
    out = []
    for fc in range(nfields):
        dec.composite_fields(dst, 0, 1, 0, ntsc, fc)
        dec.sync()
        out.append(dst.download(np.uint16, n).reshape(vp["line_count"], vp["line_width"]))
    dst.free()
    return np.stack(out)


def test_composite_vs_reference_golden(efx, golden):
    from espflix_amd import gen
    _, _, _, frames = oracle.decode(gen.Batch(0, 1, 12, 12, 0).es(0), 0, want_frames=True)
    inputs = {"lcg": common.lcg_frames(), "random": common.random_frames(7),
              "decoded": np.concatenate([frames[10], frames[11]])}
    dec = efx.Decoder(1, 1, 2)
    for name, fr in inputs.items():
        for ntsc in (True, False):
            f = gpu_fields(efx, dec, fr, ntsc, 3)
            got = [f"{common.fnv_bytes(f[i]):016x}" for i in range(3)]
            assert got == golden["composite"][f"{name}:{'ntsc' if ntsc else 'pal'}"]
            want = oracle.video_field(fr, ntsc, 0, 3)
            # stated tolerance: +-1 LSB on the DAC (high) byte; observed: identical words
            assert np.abs((f >> 8).astype(int) - (want >> 8).astype(int)).max() <= 1
            assert np.array_equal(f, want)
    dec.close()


def test_composite_batch_of_decoded_streams(efx):
    """config 4: fields of the last decoded picture of a batch of streams, both dither phases."""
    from espflix_amd import gen
    b = gen.Batch(0, 32, 12, 12, 0)
    dec = efx.Decoder(32, 12, 2)
    dec.upload(b.all_es(), efx.FORMAT_ES)
    dec.decode()
    slot = dec.picture_slot(11)
    for ntsc in (True, False):
        vp = efx.video_params(ntsc)
        n = vp["line_width"] * vp["line_count"]
        dst = dec.alloc(32 * n * 2)
        for fc in (0, 1):
            dec.composite_fields(dst, 0, 32, slot, ntsc, fc)
            dec.sync()
            got = dst.download(np.uint16, 32 * n).reshape(32, -1)
            for k in (0, 13, 31):
                fr = dec.download_frame(k, slot)
                want = oracle.video_field(np.concatenate([fr, fr]), ntsc, fc, 1).reshape(-1)
                assert np.array_equal(got[k], want)
        dst.free()
    dec.close()


def test_pdm_vs_reference_golden_and_oracle(efx, golden):
    dec = efx.Decoder(1, 1, 2)
    S, calls = 64, 40
    pcm = np.stack([common.pdm_pcm(k, calls) for k in range(S)])
    n = pcm.shape[1]
    d_pcm, d_state, d_out = dec.alloc(pcm.nbytes), dec.alloc(S * 12), dec.alloc(S * n * 4)
    d_pcm.upload(pcm)
    d_state.upload(np.zeros(S * 3, dtype=np.int32))
    dec.pdm(S, d_pcm, n, d_state, d_out)
    dec.sync()
    got = d_out.download(np.uint16, S * 2 * n).reshape(S, 2 * n)
    assert f"{common.fnv_bytes(got[0]):016x}" == golden["pdm"]["sine220"]
    states = d_state.download(np.int32, S * 3).reshape(S, 3)
    for k in range(S):
        st = np.zeros(3, dtype=np.int32)
        assert np.array_equal(got[k], oracle.pdm(st, pcm[k]))
        assert np.array_equal(states[k], st)
    # state persists across calls (static _i0,_i1,_i2 in the sketch): 128-sample calls chained
    d_state.upload(np.zeros(S * 3, dtype=np.int32))
    d_small = dec.alloc(S * 128 * 2)
    d_o2 = dec.alloc(S * 128 * 4)
    for c in range(3):
        d_small.upload(np.ascontiguousarray(pcm[:, c * 128:(c + 1) * 128]))
        dec.pdm(S, d_small, 128, d_state, d_o2)
        dec.sync()
        part = d_o2.download(np.uint16, S * 256).reshape(S, 256)
        assert np.array_equal(part, got[:, c * 256:(c + 1) * 256])
    dec.close()


def test_display_state_hscroll_and_overlay(efx, golden):
    """the two-frame slide (_hscroll, video.cpp:1146-1154) and the overlay / progress
    bar (composite(), video.cpp:845-887) in k_composite: every field bit-exact against the
    reference goldens and the oracle; the caller decrements the blend per field as the ISR does."""
    disp = np.minimum(common.random_frames(77), 248)
    dec = efx.Decoder(1, 1, 2)
    dec.upload_frame(0, 0, disp[:efx.FRAME_BYTES])
    dec.upload_frame(0, 1, disp[efx.FRAME_BYTES:])
    d_ov = dec.alloc(1280)
    for name, front, hs, ov_seed, blend0, progress in common.DISPLAY_CASES:
        n = len(hs) if hs is not None else 6
        ov = common.overlay_bytes(ov_seed) if ov_seed is not None else None
        d_ov.upload(ov if ov is not None else np.zeros(1280, np.uint8))
        for ntsc in (True, False):
            vp = efx.video_params(ntsc)
            cnt = vp["line_width"] * vp["line_count"]
            dst = dec.alloc(cnt * 2)
            want = oracle.video_field_ex(disp, ntsc, 0, n, front, hs, ov, blend0, progress)
            blend = blend0
            got = []
            for fld in range(n):
                dec.composite_fields_ex(dst, 0, 1, front, ntsc, fld, other_slot=front ^ 1, hscroll=hs[fld] if hs else 0,
                                        overlay=d_ov if ov is not None else None, overlay_blend=blend, overlay_progress=progress)
                dec.sync()
                f = dst.download(np.uint16, cnt).reshape(vp["line_count"], vp["line_width"])
                assert np.array_equal(f, want[fld]), (name, ntsc, fld)
                got.append(f"{common.fnv_bytes(f):016x}")
                if blend > 0:
                    blend -= 1
            assert got == golden["display"][f"{name}:{'ntsc' if ntsc else 'pal'}"], name
            dst.free()
    dec.close()


def test_display_state_batch_and_arguments(efx):
    """Per-stream overlays (stride) and a shared overlay give each stream its own field; bad
    scroll values are rejected."""
    from espflix_amd import gen
    b = gen.Batch(0, 4, 2, 12, 0)
    dec = efx.Decoder(4, 2, 3)
    dec.upload(b.all_es())
    dec.decode()
    frames = [[dec.download_frame(i, s) for s in range(3)] for i in range(4)]
    ovs = np.stack([common.overlay_bytes(20 + i) for i in range(4)])
    d_ov = dec.alloc(ovs.size)
    d_ov.upload(ovs)
    vp = efx.video_params(True)
    cnt = vp["line_width"] * vp["line_count"]
    dst = dec.alloc(4 * cnt * 2)
    s1, s2 = dec.picture_slot(0), dec.picture_slot(1)
    dec.composite_fields_ex(dst, 0, 4, s2, True, 3, other_slot=s1, hscroll=-104, overlay=d_ov, overlay_stride=1280,
                            overlay_blend=17, overlay_progress=77)
    dec.sync()
    got = dst.download(np.uint16, 4 * cnt).reshape(4, vp["line_count"], vp["line_width"])
    for i in range(4):
        pair = np.concatenate([frames[i][s2], frames[i][s1]])  # oracle frame 0 = displayed, 1 = other
        want = oracle.video_field_ex(pair, True, 3, 1, 0, [-104], ovs[i], 17, 77)[0]
        assert np.array_equal(got[i], want), i
    for bad in (4, 352, -352, 1000):
        with pytest.raises(efx.EfxError):
            dec.composite_fields_ex(dst, 0, 4, s2, True, 0, other_slot=s1, hscroll=bad)
    dec.close()

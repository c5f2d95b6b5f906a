"""the SBC audio decoder as a batch kernel (k_sbc) against the oracle's
restatement of sbc_decoder.cpp and the reference-derived goldens; then SBC -> PCM -> PDM on the
device end to end."""
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


def gpu_sbc(efx, dec, streams, fb, probe=False, chunks=1):
    """streams: list of equal-length frame byte arrays.  Returns (pcm list, ret array[n, frames])."""
    n = len(streams)
    frames = streams[0].size // fb
    stride = (frames * fb + 15) & ~15
    buf = np.zeros(n * stride, dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i * stride:i * stride + frames * fb] = s[:frames * fb]
    d_fr, d_st = dec.alloc(buf.size), dec.alloc(n * efx.sbc_state_bytes())
    d_fr.upload(buf)
    d_st.upload(np.zeros(n * efx.sbc_state_bytes(), dtype=np.uint8))
    pcm_stride = frames * 256
    d_pcm, d_ret, d_cnt = dec.alloc(n * pcm_stride * 2), dec.alloc(n * frames * 4), dec.alloc(n * 4)
    pcm = [np.zeros(0, np.int16) for _ in range(n)]
    rets = np.zeros((n, frames), dtype=np.uint32)
    per = (frames + chunks - 1) // chunks
    for c in range(chunks):  # state carried across calls
        f0, f1 = c * per, min(frames, (c + 1) * per)
        if f1 <= f0:
            continue
        dec.sbc_decode(n, d_fr.ptr + f0 * fb, stride, fb, f1 - f0, d_st, d_pcm, pcm_stride, d_ret, d_cnt,
                       probe_first=probe and c == 0)
        dec.sync()
        cnt = d_cnt.download(np.uint32, n)
        allpcm = d_pcm.download(np.int16, n * pcm_stride).reshape(n, pcm_stride)
        r = d_ret.download(np.uint32, n * (f1 - f0)).reshape(n, f1 - f0)
        rets[:, f0:f1] = r
        for i in range(n):
            pcm[i] = np.concatenate([pcm[i], allpcm[i, :cnt[i]]])
    for b in (d_fr, d_st, d_pcm, d_ret, d_cnt):
        b.free()
    return pcm, rets


def unpack_ret(r):
    v = int(r) & 0xFFFF
    return (-1 if v == 0xFFFF else v, int(r) >> 16)


@pytest.mark.parametrize("case", common.SBC_CASES, ids=[c[0] for c in common.SBC_CASES])
def test_synthetic_frames(efx, golden, case):
    name, kw, n, probe = case
    fb = common.sbc_frame_bytes(kw["blocks"], 1 if kw["mode"] == 0 else 2, kw["bitpool"])
    fr = common.sbc_frames(common.seed_of(name), n, **kw)
    want, wret = oracle.sbc_decode(fr, fb, probe)
    if probe:  # decode_audio() discards the probe's PCM; the harness keeps it
        drop = wret[0][1] // 2
        want_emitted, wret = want[drop:], wret[1:]
    else:
        want_emitted = want
    dec = efx.Decoder(1, 1, 2)
    pcm, rets = gpu_sbc(efx, dec, [fr], fb, probe)
    assert [unpack_ret(r) for r in rets[0]] == wret
    assert np.array_equal(pcm[0], want_emitted)
    if not probe:
        assert f"{common.fnv_bytes(pcm[0]):016x}" == golden["sbc"][name]
    # the same stream in three calls: state (filter memory, stale samples, geometry) carries over
    pcm3, _ = gpu_sbc(efx, dec, [fr], fb, probe, chunks=3)
    assert np.array_equal(pcm3[0], want_emitted)
    dec.close()


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_clip_audio(efx, golden, clips, clip):
    es = oracle.ts_audio_es(clips[clip])
    fb = common.CLIP_SBC_FRAME_BYTES[clip]
    n = es.size // fb
    want, wret = oracle.sbc_decode(es[:n * fb], fb, True)
    assert f"{common.fnv_bytes(want):016x}" == golden["sbc"]["clip:" + clip]["pcm_fnv"]
    dec = efx.Decoder(1, 1, 2)
    pcm, rets = gpu_sbc(efx, dec, [es[:n * fb]], fb, True)
    assert np.array_equal(pcm[0], want[128:])
    assert all(unpack_ret(r) == (fb, 256) for r in rets[0])
    dec.close()


def test_rejected_frames(efx):
    fr = common.sbc_frames(5, 14, freq=3, blocks=16, mode=0, alloc=0, bitpool=28).reshape(14, -1).copy()
    fr[3, 0] = 0x9D   # bad sync: previous samples are synthesised again
    fr[6, 1] |= 0x0C  # joint stereo: geometry changes to 2 channels, stale samples
    fr[9, 1] &= 0xFE  # 4 subbands: returns -1 without synthesis ...
    fr[10, 0] = 0     # ... and a bad sync right after still yields nothing
    fr[12, 2] = 200   # bitpool the allocation loop cannot meet
    want, wret = oracle.sbc_decode(fr.reshape(-1), fr.shape[1])
    dec = efx.Decoder(1, 1, 2)
    pcm, rets = gpu_sbc(efx, dec, [fr.reshape(-1)], fr.shape[1])
    assert [unpack_ret(r) for r in rets[0]] == wret
    assert [r[0] for r in wret][3] == -1 and wret[9] == (-1, 0) and wret[10] == (-1, 0)
    assert np.array_equal(pcm[0], want)
    dec.close()


def test_parallel_and_serial_paths_side_by_side(efx):
    """Streams whose frames all decode with one geometry take the regular frame-parallel kernel (k_sbc_par), a stream with
    a rejected frame the general one (k_sbc_plan + k_sbc_gen) -- in the same call, and a stream changes sides from one call
    to the next with its state (filter memory, stale samples) carried over.  21 and 13 frames: chunks of eight with a tail."""
    kw = dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
    fb = common.sbc_frame_bytes(16, 1, 28)
    clean = [common.sbc_frames(100 + i, 34, **kw) for i in range(3)]
    dirty = common.sbc_frames(200, 34, **kw).reshape(34, -1).copy()
    dirty[5, 0] = 0x9D    # first call: a bad sync byte (the general kernel's case); second call: clean
    streams = [clean[0], dirty.reshape(-1), clean[1], clean[2].copy()]
    streams[3].reshape(34, -1)[30, 0] = 0x9D   # clean in the first call (frames 0..20), rejected frame in the second
    dec = efx.Decoder(1, 1, 2)
    n = len(streams)
    stride = 34 * fb + 16
    buf = np.zeros(n * stride, dtype=np.uint8)
    for i, st in enumerate(streams):
        buf[i * stride:i * stride + st.size] = st
    d_fr, d_st = dec.alloc(buf.size), dec.alloc(n * efx.sbc_state_bytes())
    d_fr.upload(buf)
    d_st.upload(np.zeros(n * efx.sbc_state_bytes(), dtype=np.uint8))
    d_pcm, d_ret, d_cnt = dec.alloc(n * 34 * 256 * 2), dec.alloc(n * 34 * 4), dec.alloc(n * 4)
    got = [np.zeros(0, np.int16) for _ in range(n)]
    for f0, f1 in ((0, 21), (21, 34)):
        dec.sbc_decode(n, d_fr.ptr + f0 * fb, stride, fb, f1 - f0, d_st, d_pcm, 34 * 256, d_ret, d_cnt)
        dec.sync()
        cnt = d_cnt.download(np.uint32, n)
        allpcm = d_pcm.download(np.int16, n * 34 * 256).reshape(n, 34 * 256)
        for i in range(n):
            got[i] = np.concatenate([got[i], allpcm[i, :cnt[i]]])
    for i, st in enumerate(streams):
        want, _ = oracle.sbc_decode(st, fb)
        assert np.array_equal(got[i], want), i
    for b in (d_fr, d_st, d_pcm, d_ret, d_cnt):
        b.free()
    dec.close()


def _decode_batch(efx, dec, streams, fb, frames, probe, calls, serial, state0=None):
    """streams of `frames` frames each, decoded in `calls` calls (state carried over).  Returns per stream the PCM and the
    return values, and the final states."""
    n = len(streams)
    stride = (frames * fb + 15) & ~15
    buf = np.zeros(n * stride + 1024, dtype=np.uint8)
    for i, st in enumerate(streams):
        buf[i * stride:i * stride + frames * fb] = st
    dec.set_option(efx.OPT_SBC_SERIAL, 1 if serial else 0)
    d_fr, d_st = dec.alloc(buf.size), dec.alloc(n * efx.sbc_state_bytes())
    d_fr.upload(buf)
    d_st.upload(np.zeros(n * efx.sbc_state_bytes(), dtype=np.uint8) if state0 is None else state0)
    pcm_stride = frames * 256
    d_pcm, d_ret, d_cnt = dec.alloc(n * pcm_stride * 2), dec.alloc(n * frames * 4), dec.alloc(n * 4)
    pcm = [[] for _ in range(n)]
    rets = [[] for _ in range(n)]
    per = (frames + calls - 1) // calls
    for c in range(calls):
        f0, f1 = c * per, min(frames, (c + 1) * per)
        if f1 <= f0:
            continue
        d_ret.upload(np.full(n * frames, 0xDEADBEEF, dtype=np.uint32))
        dec.sbc_decode(n, d_fr.ptr + f0 * fb, stride, fb, f1 - f0, d_st, d_pcm, pcm_stride, d_ret, d_cnt, probe_first=probe and c == 0)
        dec.sync()
        cnt = d_cnt.download(np.uint32, n)
        allpcm = d_pcm.download(np.int16, n * pcm_stride).reshape(n, pcm_stride)
        r = d_ret.download(np.uint32, n * (f1 - f0)).reshape(n, f1 - f0)
        for i in range(n):
            pcm[i].append(allpcm[i, :cnt[i]].copy())
            rets[i] += [unpack_ret(x) for x in r[i]]
    states = d_st.download(np.uint8, n * efx.sbc_state_bytes()).reshape(n, -1).copy()
    for b in (d_fr, d_st, d_pcm, d_ret, d_cnt):
        b.free()
    dec.set_option(efx.OPT_SBC_SERIAL, 0)
    return [np.concatenate(p) for p in pcm], rets, states


@pytest.mark.parametrize("case", [0, 2, 3, 1], ids=["mono16", "dual8", "stereo4", "mono12"])
def test_rejected_frames_decode_frame_parallel(efx, case):
    """A stream with frames the reference rejects -- or that change the geometry mid-stream -- is resolved by k_sbc_plan (the
    geometry, the stale samples, the PCM offsets and the rows a frame puts on each channel's timeline as prefix scans) and
    decoded in chunks of eight frames by k_sbc_gen: against the oracle (PCM and return values of every frame), and against
    the one-wave-per-stream kernel (the decoder states both leave, byte for byte).  48 streams per case: clean ones (the
    regular kernels, in the same launch), lightly and heavily damaged ones; in one call with and without decode_audio()'s
    probe, and in three calls with the state carried over."""
    name, kw, _, _ = common.SBC_CASES[case]
    ch = 1 if kw["mode"] == 0 else 2
    fb = common.sbc_frame_bytes(kw["blocks"], ch, kw["bitpool"])
    frames = 45
    rng = np.random.default_rng(77 + case)
    streams = []
    for i in range(48):
        fr = common.sbc_frames(3000 + 100 * case + i, frames, **kw)
        if i % 4:
            fr = common.sbc_mutate(rng, fr, fb, frames, hits=None if i % 4 == 3 else 1)
        streams.append(fr)
    dec = efx.Decoder(1, 1, 2)
    for probe, calls in ((False, 1), (True, 1), (False, 3)):
        got, rets, states = _decode_batch(efx, dec, streams, fb, frames, probe, calls, serial=False)
        ser, sret, sstates = _decode_batch(efx, dec, streams, fb, frames, probe, calls, serial=True)
        for i, fr in enumerate(streams):
            if calls == 1:
                want, wret = oracle.sbc_decode(fr, fb, probe)
                if probe:
                    want, wret = want[wret[0][1] // 2:], wret[1:]
                assert rets[i] == wret, (name, probe, i)
                assert np.array_equal(got[i], want), (name, probe, i)
            # (three calls: a frame that runs past the frame size reads zeros beyond its call's last frame -- the oracle,
            # one call, reads the next frame's bytes: the serial kernel, same calls, is the reference there)
            assert rets[i] == sret[i], (name, probe, calls, i)
            assert np.array_equal(got[i], ser[i]), (name, probe, calls, i)
            assert np.array_equal(states[i], sstates[i]), (name, probe, calls, i)
    dec.close()


@pytest.mark.parametrize("blocks,mode,frames", [(12, 0, 760), (12, 1, 700), (8, 0, 1150), (8, 2, 1100), (16, 0, 2450), (16, 1, 2400),
                                                (4, 0, 4300), (4, 2, 4200)],
                         ids=["mono12x760", "dual12x700", "mono8x1150", "stereo8x1100", "mono16x2450", "dual16x2400", "mono4x4300",
                              "stereo4x4200"])
def test_calls_of_thousands_of_frames(efx, blocks, mode, frames):
    """Round-5 ADVICE (high): the frame-parallel kernels divided a block's number ON THE CALL'S TIMELINE by the block count
    with a 16-bit reciprocal that is exact below 4096 only -- first wrong at frame 688 of a 12-block stream, 1032 (8 blocks),
    2312 (16), 4104 (4); no test had more than 130 frames per call.  One call of more frames than that for every block count,
    mono and two channels, with and without decode_audio()'s probe: a clean stream (k_sbc_par_*), one with a rejected frame
    near the end and one with a rejected frame near the start (k_sbc_plan + k_sbc_gen, and the regular granules of a general
    stream), against the oracle and against the one-wave kernel (states byte for byte)."""
    kw = dict(freq=3, blocks=blocks, mode=mode, alloc=frames & 1, bitpool=21 if mode == 0 else 37)
    ch = 1 if mode == 0 else 2
    fb = common.sbc_frame_bytes(blocks, ch, kw["bitpool"])
    streams = [common.sbc_frames(7000 + blocks + 3 * mode + i, frames, **kw) for i in range(3)]
    streams[1] = streams[1].copy()
    streams[1].reshape(frames, fb)[frames - 19, 0] = 0x9D  # bad sync byte: the samples before it are synthesised again
    streams[2] = streams[2].copy()
    streams[2].reshape(frames, fb)[11, 0] = 0x9D
    dec = efx.Decoder(1, 1, 2)
    for probe in (False, True):
        got, rets, states = _decode_batch(efx, dec, streams, fb, frames, probe, 1, serial=False)
        ser, sret, sstates = _decode_batch(efx, dec, streams, fb, frames, probe, 1, serial=True)
        for i, fr in enumerate(streams):
            want, wret = oracle.sbc_decode(fr, fb, probe)
            if probe:
                want, wret = want[wret[0][1] // 2:], wret[1:]
            assert rets[i] == wret, (probe, i)
            bad = np.flatnonzero(got[i] != want) if got[i].size == want.size else None
            assert bad is not None and bad.size == 0, (probe, i, got[i].size, want.size,
                                                        None if bad is None else int(bad[0]) // (ch * blocks * 8))
            assert np.array_equal(got[i], ser[i]) and rets[i] == sret[i], (probe, i)
            assert np.array_equal(states[i], sstates[i]), (probe, i)
    dec.close()


def test_frame_that_runs_past_the_stated_frame_size(efx):
    """The caller states ONE frame size (decode_audio() takes it from the probe); a frame with a larger bitpool runs past
    it and the reference reads on into the next frame's bytes (get_samples() never looks at the length again,
    sbc_decoder.cpp:297-343).  So does every kernel -- also when such a frame is the last of a chunk of eight."""
    kw = dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
    fb = common.sbc_frame_bytes(16, 1, 28)
    fr = common.sbc_frames(4242, 40, **kw).reshape(40, fb).copy()
    for f in (7, 15, 16, 39):
        fr[f, 2] = 60  # 124 bytes of sample bits in a 64-byte frame
    fr = fr.reshape(-1)
    want, wret = oracle.sbc_decode(fr, fb)
    dec = efx.Decoder(1, 1, 2)
    for serial in (False, True):
        got, rets, _ = _decode_batch(efx, dec, [fr], fb, 40, False, 1, serial)
        assert rets[0] == wret, serial
        assert np.array_equal(got[0], want), serial
    dec.close()


_GUARD_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import common, espflix_amd as efx
efx.load_library()
for n_streams, frames in ((48, 45), (7, 33), (130, 9), (3, 375)):
    kw = dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
    fb = common.sbc_frame_bytes(16, 1, 28)
    rng = np.random.default_rng(n_streams)
    streams = [common.sbc_frames(50 + i, frames, **kw) for i in range(n_streams)]
    streams = [common.sbc_mutate(rng, s, fb, frames, hits=2) if i % 2 else s for i, s in enumerate(streams)]
    dec = efx.Decoder(1, 1, 2)
    d_fr, d_st = dec.alloc(n_streams * frames * fb), dec.alloc(n_streams * efx.sbc_state_bytes())
    d_pcm, d_ret, d_cnt = dec.alloc(n_streams * frames * 256 * 2), dec.alloc(n_streams * frames * 4), dec.alloc(n_streams * 4)
    d_fr.upload(np.concatenate(streams))
    d_st.upload(np.zeros(n_streams * efx.sbc_state_bytes(), dtype=np.uint8))
    for probe in (False, True):
        dec.sbc_decode(n_streams, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 256, d_ret, d_cnt, probe_first=probe)
        dec.sync()
    dec.close()
print("GUARD_OK")
"""


@pytest.mark.parametrize("guard", ["1", "2"])
def test_every_buffer_of_a_call_under_the_guard_page_allocator(guard):
    """EFX_GUARD: every device buffer is its own mapping that ends (1) or starts (2) on an unmapped page -- a kernel that
    reads or writes one element past a buffer faults.  Stream and frame counts whose products do not divide evenly: the
    tables of (stream, granule) slots were once sized from streams x (frames + 1) / 4 and overran by a few entries at 48 x
    45, which only this allocator noticed.  The frame buffer ends with the last frame's last byte (no slack): the
    frame-parallel kernels read whole aligned dwords and the bit fields of a frame that runs past the frame size."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EFX_GUARD=guard)
    r = subprocess.run([sys.executable, "-c", _GUARD_CHILD, root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GUARD_OK" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-1500:])


@pytest.mark.parametrize("case", [5, 6], ids=["mono_bp128", "dual_bp128"])
def test_frames_too_fat_for_the_stage(efx, case):
    """The frame-parallel kernels copy the frame bytes a work item reaches into 4 KB of LDS; 264-byte mono frames x 17 or
    524-byte dual-channel frames x 9 do not fit, and the bit fields are read from global memory instead -- over several
    chunks (the frames before a chunk that hold its nine blocks of filter memory included), clean and with rejected frames,
    the probe on and off."""
    name, kw, _, _ = common.SBC_CASES[case]
    ch = 1 if kw["mode"] == 0 else 2
    fb = common.sbc_frame_bytes(kw["blocks"], ch, kw["bitpool"])
    frames = 41
    rng = np.random.default_rng(case)
    streams = [common.sbc_frames(8100 + i, frames, **kw) for i in range(6)]
    for i in (1, 3, 5):
        streams[i] = common.sbc_mutate(rng, streams[i], fb, frames, hits=2)
    dec = efx.Decoder(1, 1, 2)
    for probe in (False, True):
        got, rets, states = _decode_batch(efx, dec, streams, fb, frames, probe, 1, serial=False)
        ser, sret, sstates = _decode_batch(efx, dec, streams, fb, frames, probe, 1, serial=True)
        for i, fr in enumerate(streams):
            want, wret = oracle.sbc_decode(fr, fb, probe)
            if probe:
                want, wret = want[wret[0][1] // 2:], wret[1:]
            assert rets[i] == wret == sret[i], (name, probe, i)
            assert np.array_equal(got[i], want) and np.array_equal(ser[i], want), (name, probe, i)
            assert np.array_equal(states[i], sstates[i]), (name, probe, i)
    dec.close()


@pytest.mark.parametrize("n_streams,frames,fb_override", [(3000, 3, None), (1, 1, None), (5, 10, 3), (2, 130, None), (257, 17, None)])
def test_odd_batch_shapes(efx, n_streams, frames, fb_override):
    """Many streams of a few frames (the work lists are longer than one pass of the classifier, a chunk is the whole call), one
    frame, frames too short to hold a header (every one rejected: nothing is synthesised under a zero state), streams of more
    frames than a wave has lanes (the plan's carries from tile to tile): every other stream damaged, the frame-parallel
    kernels against the one-wave kernel -- PCM, return values, decoder states."""
    kw = dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
    fb = common.sbc_frame_bytes(16, 1, 28)
    rng = np.random.default_rng(n_streams * 1000 + frames)
    base = [common.sbc_frames(9000 + i, frames, **kw) for i in range(min(n_streams, 16))]
    streams = []
    for i in range(n_streams):
        st = base[i % len(base)]
        if i % 2 and frames > 1:
            st = common.sbc_mutate(rng, st, fb, frames, hits=1 + i % 3)
        streams.append(st)
    if fb_override:
        streams = [st[:frames * fb_override].copy() for st in streams]
        fb = fb_override
    dec = efx.Decoder(1, 1, 2)
    for probe in (False, True):
        got, rets, states = _decode_batch(efx, dec, streams, fb, frames, probe, 1, serial=False)
        ser, sret, sstates = _decode_batch(efx, dec, streams, fb, frames, probe, 1, serial=True)
        assert rets == sret
        assert all(np.array_equal(a, b) for a, b in zip(got, ser))
        assert np.array_equal(states, sstates)
    dec.close()


def test_state_no_call_leaves_goes_to_the_serial_kernel(efx):
    """A decoder state with a block count that is no multiple of four (no header leaves one) and frames that are synthesised
    under it -- bad sync bytes up front -- would put 5 rows per frame on a timeline: k_sbc_plan hands such a stream to the one
    wave that walks the frames (k_sbc_finish), next to regular and general streams in the same call."""
    kw = dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
    fb = common.sbc_frame_bytes(16, 1, 28)
    frames = 20
    streams = [common.sbc_frames(7000 + i, frames, **kw) for i in range(6)]
    for i in (1, 3, 4):
        streams[i] = streams[i].copy()
        streams[i].reshape(frames, fb)[0:3, 0] = 0  # synthesised under the state's geometry
    sb = efx.sbc_state_bytes()
    st = np.zeros((6, sb), dtype=np.uint8)
    rng = np.random.default_rng(9)
    for i, blocks in ((1, 5), (3, 7), (4, 8)):  # (4: a state a call CAN leave: the general kernel's)
        st[i, 1152:2176] = rng.integers(-2000, 2000, 256).astype(np.int32).view(np.uint8)
        st[i, :1152] = rng.integers(-30000, 30000, 288).astype(np.int32).view(np.uint8)
        st[i, 2176:2183] = (3, blocks, 1, 0, 0, 8, 28)
    dec = efx.Decoder(1, 1, 2)
    got, rets, states = _decode_batch(efx, dec, streams, fb, frames, False, 1, False, st.reshape(-1))
    ser, sret, sstates = _decode_batch(efx, dec, streams, fb, frames, False, 1, True, st.reshape(-1))
    for i in range(6):
        assert rets[i] == sret[i], i
        assert np.array_equal(got[i], ser[i]), i
        assert np.array_equal(states[i], sstates[i]), i
    assert len(got[1]) == 3 * 5 * 8 + 17 * 128 and len(got[4]) == 3 * 8 * 8 + 17 * 128
    dec.close()


def test_state_carried_into_a_short_second_call_repeated(efx):
    """A call of nine frames is two frame-parallel chunks -- frames 0..7 and a light tail of one -- and the tail leaves the
    stream's state while the first chunk may not have read the filter memory yet: the state is handed over through a scratch
    copy (k_sbc_commit), never written in the launch that reads it.  512 streams, non-zero carried-over state, the nine-frame
    call replayed from a snapshot of that state a dozen times; also a stereo stream count and a 17-frame call."""
    S = 512
    for kw, ch in ((dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28), 1), (dict(freq=2, blocks=8, mode=1, alloc=0, bitpool=35), 2)):
        fb = common.sbc_frame_bytes(kw["blocks"], ch, kw["bitpool"])
        spf = kw["blocks"] * 8 * ch
        n0 = 16
        for n1 in (9, 17):
            one = [common.sbc_frames(900 + i, n0 + n1, **kw) for i in range(8)]
            want = [oracle.sbc_decode(o, fb)[0] for o in one]
            stride = ((n0 + n1) * fb + 15) & ~15
            buf = np.zeros(S * stride, dtype=np.uint8)
            for i in range(S):
                buf[i * stride:i * stride + one[i % 8].size] = one[i % 8]
            dec = efx.Decoder(1, 1, 2)
            d_fr, d_st = dec.alloc(buf.size), dec.alloc(S * efx.sbc_state_bytes())
            d_pcm, d_cnt = dec.alloc(S * (n0 + n1) * spf * 2), dec.alloc(S * 4)
            d_fr.upload(buf)
            d_st.upload(np.zeros(S * efx.sbc_state_bytes(), dtype=np.uint8))
            dec.sbc_decode(S, d_fr, stride, fb, n0, d_st, d_pcm, (n0 + n1) * spf, None, d_cnt)
            dec.sync()
            snapshot = d_st.download(np.uint8, S * efx.sbc_state_bytes())
            assert snapshot.any()
            for rep in range(12):
                d_st.upload(snapshot)
                dec.sbc_decode(S, d_fr.ptr + n0 * fb, stride, fb, n1, d_st, d_pcm, (n0 + n1) * spf, None, d_cnt)
                dec.sync()
                assert (d_cnt.download(np.uint32, S) == n1 * spf).all()
                pcm = d_pcm.download(np.int16, S * (n0 + n1) * spf).reshape(S, -1)[:, :n1 * spf]
                for i in range(S):
                    assert np.array_equal(pcm[i], want[i % 8][n0 * spf:]), (ch, n1, rep, i)
            for b in (d_fr, d_st, d_pcm, d_cnt):
                b.free()
            dec.close()


def test_batch_of_streams_and_pdm_chain(efx):
    """256 streams with different content decode independently; the PCM then feeds k_pdm on the
    device (config 4's audio half: SBC -> PCM -> PDM) and matches the oracle chain."""
    S, frames = 256, 24
    kw = dict(freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
    fb = common.sbc_frame_bytes(16, 1, 28)
    streams = [common.sbc_frames(1000 + i, frames, **kw) for i in range(S)]
    dec = efx.Decoder(S, 1, 2)
    pcm, _ = gpu_sbc(efx, dec, streams, fb, True)
    for i in (0, 1, 77, 255):
        want, wret = oracle.sbc_decode(streams[i], fb, True)
        assert np.array_equal(pcm[i], want[128:]), i
    assert len({p.tobytes() for p in pcm}) == S
    n = frames * 128
    d_pcm, d_state, d_out = dec.alloc(S * n * 2), dec.alloc(S * 12), dec.alloc(S * n * 4)
    d_pcm.upload(np.concatenate(pcm))
    d_state.upload(np.zeros(S * 3, dtype=np.int32))
    dec.pdm(S, d_pcm, n, d_state, d_out)
    dec.sync()
    words = d_out.download(np.uint16, S * n * 2).reshape(S, n * 2)
    for i in (0, 200):
        st = np.zeros(3, dtype=np.int32)
        assert np.array_equal(words[i], oracle.pdm(st, pcm[i])), i
    dec.close()


def hostile_audio_ts(seed: int) -> bytes:
    """Hostile audio (common.hostile_audio_packets) inside a valid video stream: > 128 packets, so the
    gate crosses k_demux's chunk boundary."""
    from espflix_amd import gen
    return common.interleave_audio(gen.Batch(400 + seed, 1, 8).ts(0).tobytes(), seed)


def test_audio_demux_on_device_and_full_chain(efx, golden, clips):
    """efx_demux_audio: the bytes push_audio() receives, for the clips and for hostile muxing, equal
    the oracle's; then TS -> audio ES -> SBC -> PCM entirely on the device equals the golden PCM."""
    streams = [clips["splash"].tobytes(), clips["vmedia"].tobytes()] + [hostile_audio_ts(s) for s in (1, 2, 3)] + [b""]
    stride = (max(len(s) for s in streams) + 255) & ~255
    dec = efx.Decoder(len(streams), 1, 2, max_stream_bytes=sum(len(s) for s in streams) + 4096)
    d_audio, d_len = dec.alloc(len(streams) * stride), dec.alloc(len(streams) * 4)
    dec.demux_audio(streams, d_audio, stride, d_len)
    lens = d_len.download(np.uint32, len(streams))
    audio = d_audio.download(np.uint8, len(streams) * stride).reshape(len(streams), stride)
    for i, ts in enumerate(streams):
        want = oracle.ts_audio_es(np.frombuffer(ts, dtype=np.uint8))
        assert lens[i] == want.size, (i, lens[i], want.size)
        assert np.array_equal(audio[i, :lens[i]], want), i
    assert lens[2] > 0 and lens[5] == 0
    # the clips' audio straight from the device buffer into k_sbc
    for i, clip in enumerate(("splash", "vmedia")):
        fb = common.CLIP_SBC_FRAME_BYTES[clip]
        frames = int(lens[i]) // fb
        d_st, d_pcm, d_cnt = dec.alloc(efx.sbc_state_bytes()), dec.alloc(frames * 256 * 2), dec.alloc(4)
        d_st.upload(np.zeros(efx.sbc_state_bytes(), dtype=np.uint8))
        dec.sbc_decode(1, d_audio.ptr + i * stride, stride, fb, frames, d_st, d_pcm, frames * 256, None, d_cnt, probe_first=True)
        dec.sync()
        cnt = int(d_cnt.download(np.uint32, 1)[0])
        pcm = d_pcm.download(np.int16, cnt)
        want, _ = oracle.sbc_decode(audio[i, :frames * fb], fb, True)
        assert np.array_equal(pcm, want[128:])
        assert f"{common.fnv_bytes(want):016x}" == golden["sbc"]["clip:" + clip]["pcm_fnv"]
    dec.close()

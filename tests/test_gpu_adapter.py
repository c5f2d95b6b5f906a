"""The C++ drop-in layer (include/efx_player.hpp): a host program written against the reference's
MpegDecoder / Frame / push_video / video_isr / write_pcm_16 surface is compiled against it and must
deliver the frames, field and PDM words the reference delivers."""
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def build(tmp_path):
    exe = str(tmp_path / "adapter_main")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "adapter_main.cpp"), "-L", os.path.join(ROOT, "espflix_amd"), "-lefx",
                    "-Wl,-rpath," + os.path.join(ROOT, "espflix_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                    "-lamdhip64", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_reference_style_host_program(tmp_path, clip, golden):
    exe = build(tmp_path)
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", clip + ".ts")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    rows = [l.split() for l in p.stdout.splitlines()]
    frames = [r for r in rows if r[0] == "F"]
    g = golden["clips"][clip]
    assert [r[3] for r in frames] == g["hashes"]
    assert [int(r[2]) for r in frames] == g["pts"]
    # composite field of the last pushed frame, frame_counter 0
    ts = np.fromfile(os.path.join(ROOT, "tests", "golden", clip + ".ts"), dtype=np.uint8)
    _, _, _, fr = oracle.decode(ts, 1, want_frames=True)
    want = oracle.video_field(np.concatenate([fr[-1], fr[-1]]), True, 0, 1)
    v = [r for r in rows if r[0] == "V"][0][1]
    assert v == f"{oracle.fnv1a64(want.reshape(-1).view(np.uint8)):016x}"
    # slide (mode 3) to the previous picture under the fading overlay: fields 1..3
    ov = (np.arange(1280) * 7).astype(np.uint8)
    want = oracle.video_field_ex(np.concatenate([fr[-2], fr[-1]]), True, 1, 3, 0, [344, 336, 328], ov, 33, 120)
    w = [r for r in rows if r[0] == "W"][0]
    assert w[1] == f"{oracle.fnv1a64(want.reshape(-1).view(np.uint8)):016x}" and w[2] == "30"
    # three write_pcm_16 calls (the middle one silence), state carried across calls
    import ctypes
    st = np.zeros(3, dtype=np.int32)
    beep = ctypes.c_int(0)
    words = []
    for c in range(3):
        pcm = np.array([(i * 37 + c * 1000) % 4001 - 2000 for i in range(128)], dtype=np.int16)
        words.append(oracle.write_pcm_16(st, beep, None if c == 1 else pcm))
    a = [r for r in rows if r[0] == "A"][0][1]
    assert a == f"{oracle.fnv1a64(np.concatenate(words).view(np.uint8)):016x}"
    # beep(): five tone bursts replace the PCM, the sixth call is PCM again
    beep.value = 5
    words = []
    for c in range(6):
        pcm = np.array([(i * 11 + c * 300) % 2001 - 1000 for i in range(128)], dtype=np.int16)
        words.append(oracle.write_pcm_16(st, beep, pcm))
    b = [r for r in rows if r[0] == "B"][0][1]
    assert b == f"{oracle.fnv1a64(np.concatenate(words).view(np.uint8)):016x}"
    # the audio bytes push_audio() received (PID 0x102 of the clip)
    u = [r for r in rows if r[0] == "U"][0]
    es = oracle.ts_audio_es(ts)
    assert int(u[1]) == es.size and u[2] == f"{oracle.fnv1a64(es):016x}"


def test_streaming_play_of_a_thousand_pictures(tmp_path):
    """A play far longer than one decode window (84 GOPs): the adapter decodes window by window while the
    Buffers arrive, the decoder state travels on the device; every pushed frame and PTS against the oracle."""
    from espflix_amd import gen
    exe = build(tmp_path)
    ts = gen.Batch(3, 1, 1008, 12, 0).ts(0)
    path = str(tmp_path / "long.ts")
    ts.tofile(path)
    p = subprocess.run([exe, path], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    frames = [l.split() for l in p.stdout.splitlines() if l.startswith("F ")]
    n, h, pts, _ = oracle.decode(ts, 1, flush_last=True, max_frames=1100)
    assert n == 1008 and len(frames) == 1008
    assert [int(r[3], 16) for r in frames] == [int(x) for x in h]
    assert [int(r[2]) for r in frames] == [int(x) for x in pts]
    assert "status" not in p.stderr and "without a picture-aligned" not in p.stderr


def test_unaligned_pes_stream_is_decoded_at_its_end(tmp_path):
    """PES packets that never start at a picture (hostile muxing) offer no cut point: the adapter decodes the
    play in one window when the zero-length Buffer arrives -- same frames, same PTS."""
    import common
    from espflix_amd import gen
    exe = build(tmp_path)
    es = gen.Batch(5, 1, 24, 12, 0).es(0).tobytes()
    ts = np.frombuffer(common.hostile_ts(es, 77), dtype=np.uint8)
    path = str(tmp_path / "hostile.ts")
    ts.tofile(path)
    p = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    frames = [l.split() for l in p.stdout.splitlines() if l.startswith("F ")]
    n, h, pts, _ = oracle.decode(ts, 1, flush_last=True)
    assert [int(r[3], 16) for r in frames] == [int(x) for x in h]
    assert [int(r[2]) for r in frames] == [int(x) for x in pts]


@pytest.mark.parametrize("clip", ["splash", "vmedia"])
def test_paced_feeder_gets_the_reference_latency(tmp_path, clip, golden):
    """A feeder slower than the decoder (a real-time play): the adaptive window decodes whenever no Buffer is waiting and
    a picture is pushed when the PES of its successor has arrived -- the reference pushes it at its successor's header
    (flush_picture, player.cpp:692-702).  Checked on the Buffer count at each push_video(n): no Buffer beyond the one
    that holds the PES start of picture n + 1 has been handed over yet.  Frames and PTS stay exact."""
    exe = build(tmp_path)
    path = os.path.join(ROOT, "tests", "golden", clip + ".ts")
    # the feeder's pace: the clip at its own 30 pictures per second, Buffer by Buffer
    n_buffers = os.path.getsize(path) / 1504
    usec = int(2e6 * len(golden["clips"][clip]["hashes"]) / 30 / n_buffers)   # (half speed: the test must not depend on how
                                                                               # busy the box is; the decoder needs 2-5 ms per window)
    p = subprocess.run([exe, path, "paced", str(usec)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    frames = [l.split() for l in p.stdout.splitlines() if l.startswith("F ")]
    g = golden["clips"][clip]
    assert [r[3] for r in frames] == g["hashes"] and [int(r[2]) for r in frames] == g["pts"]
    # Buffer (8 transport packets = 1504 bytes, streamer.h:142) that holds the PES start of every picture
    ts = np.fromfile(path, dtype=np.uint8)
    starts = []
    for k in range(ts.size // 188):
        pkt = ts[k * 188:(k + 1) * 188]
        if pkt[0] == 0x47 and ((int(pkt[1]) << 8 | int(pkt[2])) & 0x1FFF) == 0x100 and pkt[1] & 0x40:
            d = 4 + (1 + int(pkt[4]) if pkt[3] & 0x20 else 0)
            es = d + 9 + int(pkt[d + 8])
            if es + 4 <= 188 and bytes(pkt[es:es + 3]) == b"\x00\x00\x01" and pkt[es + 3] in (0x00, 0xB3, 0xB8):
                starts.append(k * 188 // 1504)
    assert len(starts) >= len(frames)
    # picture n is pushed while the Buffer that holds the PES start of picture n + 1 is the newest one handed over
    # (fed == its index + 1) -- in particular before the Buffer of picture n + 2's first byte is, unless that is the same one
    # (picture 0 excepted: the first window of a play also seeds the device ring with the host's two Frames)
    late = [n for n, r in enumerate(frames[:-1]) if 0 < n and n + 1 < len(starts) and int(r[4]) > starts[n + 1] + 1]
    assert len(late) <= 2, (f"pictures pushed later than the reference would: {late[:10]}; "
                      f"fed {[int(r[4]) for r in frames[:6]]}, picture starts in Buffers {starts[:8]}")

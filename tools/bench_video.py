#!/usr/bin/env python3
"""Throughput of the video-out / audio-out kernels (BASELINE configs[3]): composite NTSC / PAL
fields and PDM blocks and the transport-stream demultiplexer for a batch of streams, with the HBM roofline fraction from the
algorithmic bytes of SURVEY.md section 8d (field: 101 376 B read + 477 888 B written NTSC,
708 864 B PAL; PDM: 2 B read + 4 B written per sample).  Prints one JSON line per kernel."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import espflix_amd as efx
from espflix_amd import gen

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = 50
b = gen.Batch(0, S, 2, 12, 0)
dec = efx.Decoder(S, 2, 2)
dec.upload(b.all_es())
dec.decode()
slot = dec.picture_slot(1)
for ntsc in (True, False):
    vp = efx.video_params(ntsc)
    n = vp["line_width"] * vp["line_count"]
    dst = dec.alloc(S * n * 2)
    for _ in range(3):
        dec.composite_fields(dst, 0, S, slot, ntsc, 0)
    dec.sync()
    t0 = time.perf_counter()
    for i in range(REPS):
        dec.composite_fields(dst, 0, S, slot, ntsc, i)
    dec.sync()
    dt = (time.perf_counter() - t0) / REPS
    alg = S * (efx.FRAME_BYTES + n * 2)
    print(json.dumps({"kernel": "k_composite", "standard": "ntsc" if ntsc else "pal", "streams": S,
                      "fields_per_s": S / dt, "ms_per_launch": dt * 1e3,
                      "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / dt / 8e12, "algorithmic_bytes_per_launch": alg}}))
    dst.free()
# PDM: S streams x 375 calls of 128 samples (one second of audio)
n = 128 * 375
pcm = np.round(8000 * np.sin(2 * np.pi * 220 * np.arange(n) / 48000)).astype(np.int16)
d_pcm, d_state, d_out = dec.alloc(S * n * 2), dec.alloc(S * 12), dec.alloc(S * n * 4)
d_pcm.upload(np.tile(pcm, S))
d_state.upload(np.zeros(S * 3, dtype=np.int32))
dec.pdm(S, d_pcm, n, d_state, d_out)
dec.sync()
t0 = time.perf_counter()
for i in range(5):
    dec.pdm(S, d_pcm, n, d_state, d_out)
dec.sync()
dt = (time.perf_counter() - t0) / 5
alg = S * n * 6
print(json.dumps({"kernel": "k_pdm", "streams": S, "samples_per_stream": n, "stream_seconds_per_s": S / dt,
                  "ms_per_launch": dt * 1e3,
                  "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / dt / 8e12,
                               "note": "serial 32x recurrence per sample: ALU bound, one lane per stream"}}))
dec.close()

# TS demux on the device: S streams x GOP(12), TS bytes read + ES bytes written
b = gen.Batch(0, S, 12, 12, 0)
ts = [b.ts(k) for k in range(S)]
dec = efx.Decoder(S, 12, 2, max_stream_bytes=sum(len(x) for x in ts) + 4096)
dec.set_timing(True)
best = 1e9
for _ in range(5):
    dec.upload(ts, efx.FORMAT_TS)
    dec.decode()
    t = dec.timing()
    best = min(best, t.demux_ms)
es_bytes = sum(b.es(k).size for k in range(S))
alg = t.ts_bytes + es_bytes
print(json.dumps({"kernel": "k_demux", "streams": S, "ts_bytes": t.ts_bytes, "es_bytes": es_bytes,
                  "ms_per_launch": best, "ts_GB_per_s": t.ts_bytes / best / 1e6,
                  "roofline": {"bound": "hbm", "achieved": alg / best / 1e6, "peak": 8000.0, "unit": "GB/s",
                               "frac": alg / best / 1e6 / 8000.0, "algorithmic_bytes_per_launch": alg}}))
dec.close()

# SBC audio decode: S streams x one second of 48 kHz mono audio (375 frames of 64 bytes)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
frames, fb = 375, 64
one = common.sbc_frames(1, frames, freq=3, blocks=16, mode=0, alloc=0, bitpool=28)
dec = efx.Decoder(1, 1, 2)
d_fr, d_st = dec.alloc(S * frames * fb), dec.alloc(S * efx.sbc_state_bytes())
d_fr.upload(np.tile(one, S))
d_st.upload(np.zeros(S * efx.sbc_state_bytes(), dtype=np.uint8))
d_pcm = dec.alloc(S * frames * 128 * 2)
dec.sbc_decode(S, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 128)
dec.sync()
t0 = time.perf_counter()
for i in range(5):
    dec.sbc_decode(S, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 128)
dec.sync()
dt = (time.perf_counter() - t0) / 5
alg = S * frames * (fb + 256)
print(json.dumps({"kernel": "k_sbc", "streams": S, "frames_per_stream": frames, "stream_seconds_per_s": S / dt,
                  "ms_per_launch": dt * 1e3,
                  "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / dt / 8e12,
                               "note": "frames of a stream are serial (filter memory): one wave per stream"}}))
dec.close()

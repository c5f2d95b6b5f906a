"""development aid: efx_sbc_decode timing, 1024 streams x one second of 48 kHz audio (mono bitpool 28 / stereo bitpool 53)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import espflix_amd as efx
import common
S, frames = 1024, 375
for label, kw in (("mono", dict(mode=0, bitpool=28)), ("stereo", dict(mode=2, bitpool=53))):
    one = common.sbc_frames(1, frames, freq=3, blocks=16, alloc=0, **kw)
    fb = one.size // frames
    ch = 1 if kw["mode"] == 0 else 2
    dec = efx.Decoder(1, 1, 2)
    d_fr, d_st = dec.alloc(S * frames * fb), dec.alloc(S * efx.sbc_state_bytes())
    d_fr.upload(np.tile(one, S))
    d_st.upload(np.zeros(S * efx.sbc_state_bytes(), dtype=np.uint8))
    d_pcm = dec.alloc(S * frames * 128 * ch * 2)
    dec.sbc_decode(S, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 128 * ch)
    dec.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(5):
            dec.sbc_decode(S, d_fr, frames * fb, fb, frames, d_st, d_pcm, frames * 128 * ch)
        dec.sync()
        best = min(best, (time.perf_counter() - t0) / 5)
    print(f"[{os.environ.get('SPEC','')}] sbc {label}: frame {fb} B, {best*1e3:.3f} ms per 1024 stream-seconds")
    dec.close()

#!/usr/bin/env python3
"""SBC batch timing (GPU): 1024 streams x 375 frames, clean and mixed (every fourth stream with rejected frames), with the PCM
of a few streams checked against the test oracle.  Usage: python tools/exp/r5_sbc.py [EFX_LIB=... in the environment]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import common  # noqa: E402
import espflix_amd as efx  # noqa: E402
import oracle  # noqa: E402


def main():
    efx.load_library()
    S, frames = 1024, 375
    out = {}
    only = sys.argv[1] if len(sys.argv) > 1 else ""   # e.g. mono_clean
    for label, kw in (("mono", common.SBC_CASES[0][1]), ("stereo", dict(freq=3, blocks=16, mode=2, alloc=0, bitpool=53))):
        if only and not only.startswith(label):
            continue
        ch = 1 if kw["mode"] == 0 else 2
        fb = common.sbc_frame_bytes(kw["blocks"], ch, kw["bitpool"])
        spf = kw["blocks"] * 8 * ch
        base = [common.sbc_frames(10 + i, frames, **kw) for i in range(16)]
        rng = np.random.default_rng(5)
        dirty = [common.sbc_mutate(rng, b, fb, frames, hits=1 + i % 3) for i, b in enumerate(base)]
        stream = torch.cuda.Stream()
        dec = efx.Decoder(1, 1, 2, device=torch.cuda.current_device(), hip_stream=stream.cuda_stream)
        for mix in ("clean", "mixed"):
            if only and not only.endswith(mix):
                continue
            streams = [dirty[i % 16] if (mix == "mixed" and i % 4 == 0) else base[i % 16] for i in range(S)]
            stride = frames * fb
            d_fr, d_st = dec.alloc(S * stride + 1024), dec.alloc(S * efx.sbc_state_bytes())
            d_pcm, d_cnt = dec.alloc(S * frames * 256 * 2), dec.alloc(S * 4)
            d_fr.upload(np.concatenate(streams + [np.zeros(1024, np.uint8)]))
            zeros = np.zeros(S * efx.sbc_state_bytes(), dtype=np.uint8)

            def run(_i=0):
                dec.sbc_decode(S, d_fr, stride, fb, frames, d_st, d_pcm, frames * 256, None, d_cnt)

            d_st.upload(zeros)
            run()
            dec.sync()
            cnt = d_cnt.download(np.uint32, S)
            pcm = d_pcm.download(np.int16, S * frames * 256).reshape(S, frames * 256)
            ok = True
            for i in (0, 1, 4, 5, 8, 1020, 1023):
                want, _ = oracle.sbc_decode(np.concatenate([streams[i], np.zeros(0, np.uint8)]), fb)
                # (the oracle reads past the stream's last frame into zeros; the kernels into the next stream's first bytes
                # only if a frame runs past the frame size AND is the last: the mutations' bitpool changes can do that)
                if cnt[i] != want.size or not np.array_equal(pcm[i, :cnt[i]], want):
                    nbad = int((pcm[i, :min(cnt[i], want.size)] != want[:min(cnt[i], want.size)]).sum())
                    print("MISMATCH", label, mix, i, cnt[i], want.size, nbad, file=sys.stderr)
                    ok = False
            times = {}
            for serial in (0, 1):
                dec.set_option(efx.OPT_SBC_SERIAL, serial)
                d_st.upload(zeros)
                run()
                dec.sync()
                with torch.cuda.stream(stream):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(5):
                        run()
                    e1.record(stream)
                e1.synchronize()
                times["serial_ms" if serial else "ms"] = e0.elapsed_time(e1) / 5
            dec.set_option(efx.OPT_SBC_SERIAL, 0)
            ms = times["ms"]
            out[f"{label}_{mix}"] = {"ms": round(ms, 4), "serial_kernel_ms": round(times["serial_ms"], 4), "parity": ok,
                                     "stream_seconds_per_s": round(S * frames * spf / ch / 48000 / ms * 1e3),
                                     "GB_per_s": round(S * frames * (fb + spf * 2) / ms / 1e6, 1)}
            for b in (d_fr, d_st, d_pcm, d_cnt):
                b.free()
        dec.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

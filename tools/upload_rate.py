#!/usr/bin/env python3
"""PCIe-inclusive figure for DESIGN.md: time of efx_upload_streams (host staging copy + H2D, and
k_demux for TS input) for the bench workload, next to the decode step."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espflix_amd as efx
from espflix_amd import gen

S, P = 1024, 12
b = gen.Batch(0, S, P, 12, 0, 16)
es = b.all_es()
ts = [b.ts(k) for k in range(S)]
out = {}
for name, streams, fmt in (("es", es, efx.FORMAT_ES), ("ts", ts, efx.FORMAT_TS)):
    nbytes = int(sum(len(s) for s in streams))
    dec = efx.Decoder(S, P, 2, max_stream_bytes=nbytes + 64 * S)
    dec.upload(streams, fmt)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        dec.upload(streams, fmt)
        t.append(time.perf_counter() - t0)
    dec.decode()
    t0 = time.perf_counter()
    for _ in range(10):
        dec.decode(sync=False)
    dec.sync()
    step = (time.perf_counter() - t0) / 10
    up = min(t)
    out[name] = {"bytes": nbytes, "upload_ms": up * 1e3, "upload_GBs": nbytes / up / 1e9, "step_ms": step * 1e3,
                 "frames_per_s_decode_only": S * P / step, "frames_per_s_upload_plus_decode": S * P / (step + up)}
    dec.close()
print(json.dumps(out))

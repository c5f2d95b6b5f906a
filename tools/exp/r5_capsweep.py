#!/usr/bin/env python3
"""Development aid (round 5): k_parse's residency cap once more, with three bitstream buffers and the pinned group -- 1024 streams
x GOP 12 and the 5-slice shape, back to back, 60 steps, median of 3."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import espflix_amd as efx
from espflix_amd import gen
P = 12
for flags in (0, 36):
    b = gen.Batch(0, 1024, P, 12, flags, max(1, (os.cpu_count() or 2) // 2))
    streams = b.all_es()
    es_bytes = sum(s.size for s in streams)
    for cap in (96, 128, 144, 160, 176, 192, 224, 0):
        os.environ["EFX_PARSE_WG_CAP"] = str(cap)
        dec = efx.Decoder(1024, P, 2, max_stream_bytes=es_bytes + 64 * 1024)
        dec.set_option(efx.OPT_GROUPS, 1)
        dec.upload(streams, 0)
        res = []
        for rep in range(3):
            for _ in range(5):
                dec.decode(sync=False)
            dec.sync()
            dec.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(60):
                dec.decode(sync=False)
            dec.sync()
            dt = time.perf_counter() - t0
            tp = dec.timing()
            res.append((1024 * P * 60 / dt / 1e6, tp.parse_ms, tp.recon_ms))
        r = sorted(res)[1]
        print(json.dumps({"flags": flags, "cap": cap, "Mfps": r[0], "parse_ms": r[1], "recon_ms": r[2]}), flush=True)
        dec.close()

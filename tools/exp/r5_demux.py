#!/usr/bin/env python3
"""TS demux timing (GPU): 1024 streams x GOP 12 as transport streams, efx_upload_streams(TS) stage time, best of five;
the elementary streams of eight of them checked against the generator's.  EFX_LIB selects a development build."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import espflix_amd as efx  # noqa: E402
from espflix_amd import gen  # noqa: E402


def main():
    efx.load_library()
    S = 1024
    b = gen.Batch(0, S, 12, 12, 0, 8)
    ts = [b.ts(k) for k in range(S)]
    es_bytes = sum(b.es(k).size for k in range(S))
    dec = efx.Decoder(S, 12, 2, max_stream_bytes=sum(x.size for x in ts) + 4096)
    dec.set_timing(True)
    best = None
    for _ in range(5):
        dec.upload(ts, 1)
        dec.decode()
        t = dec.timing()
        best = t.demux_ms if best is None else min(best, t.demux_ms)
    ok = all(np.array_equal(np.frombuffer(dec.es(k), dtype=np.uint8)[:b.es(k).size], b.es(k)) for k in (0, 1, 2, 3, 500, 1021, 1022, 1023))
    alg = t.ts_bytes + es_bytes
    print(json.dumps({"demux_ms": round(best, 4), "frac": round(alg / best / 1e6 / 8000, 4), "parity": bool(ok)}))
    dec.close()


if __name__ == "__main__":
    main()

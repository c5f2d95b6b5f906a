#!/usr/bin/env python3
"""Development aid: libefx variants side by side on one box (EFX_CHECK_LIBS=a,b,...: espflix_amd/libefx_<x>.so after the default):
1024 streams x GOP 12, parity against the goldens, serial stage times, back-to-back rate (60 steps, median of 3)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import espflix_amd as efx
from espflix_amd import gen
P = 12
golden = np.fromfile(os.path.join(ROOT, "tests", "golden", "bench_gop12.u64"), dtype="<u8").reshape(8192, P)
b = gen.Batch(0, 1024, P, 12, 0, max(1, (os.cpu_count() or 2) // 2))
streams = b.all_es()
es_bytes = sum(s.size for s in streams)
for tag in [""] + [x for x in os.environ.get("EFX_CHECK_LIBS", "").split(",") if x]:
    efx._lib = None
    efx.LIB_PATH = os.path.join(ROOT, "espflix_amd", f"libefx_{tag}.so" if tag else "libefx.so")
    efx.load_library()
    for rep_outer in range(2):
        dec = efx.Decoder(1024, P, 2, max_stream_bytes=es_bytes + 64 * 1024)
        dec.upload(streams, 0)
        dec.set_timing(True)
        for _ in range(4):
            dec.decode(sync=True)
        ts = dec.timing()
        dec.set_option(efx.OPT_GROUPS, 1)
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
        h = dec.frame_hashes()
        ok = all((h[:, dec.picture_slot(p)] == golden[:1024, p]).all() for p in (P - 2, P - 1)) and not any(dec.stream_status(i) for i in range(1024))
        r = sorted(res)[1]
        print(json.dumps({"lib": tag or "default", "parity": bool(ok), "serial_ms": [round(ts.index_ms, 3), round(ts.parse_ms, 3), round(ts.recon_ms, 3)],
                          "Mfps": round(r[0], 3), "parse_ms": round(r[1], 3), "recon_ms": round(r[2], 3)}), flush=True)
        dec.close()

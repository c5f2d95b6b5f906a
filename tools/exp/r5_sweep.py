#!/usr/bin/env python3
"""Development aid (round 5): how many k_recon_all waves per compute unit (persistent grid, EFX_OPT_RECON_ITEMS = 0) leave
the parse half of the next call room to run beside them -- 1024 streams x GOP 12, back to back, pinned launch structure.
Prints one JSON line per configuration and, last, {"best": {...}}."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import espflix_amd as efx
from espflix_amd import gen

P = 12
golden = np.fromfile(os.path.join(ROOT, "tests", "golden", "bench_gop12.u64"), dtype="<u8").reshape(8192, P)
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gold = golden if flags == 0 else np.fromfile(os.path.join(ROOT, "tests", "golden", "bench_wide1500k.u64"), dtype="<u8").reshape(1024, P)
b = gen.Batch(0, 1024, P, 12, flags, max(1, (os.cpu_count() or 2) // 2))
streams = b.all_es()
es_bytes = sum(s.size for s in streams)
best = None
configs = [(0, 16, 0)] + [(2, 16, 0), (2, 8, 0)] + [(2, 0, r) for r in (8, 9, 10, 11, 12, 13, 14, 16)]
for mode, items, waves in configs:
    dec = efx.Decoder(1024, P, 2, max_stream_bytes=es_bytes + 64 * 1024)
    dec.set_option(efx.OPT_RECON_MODE, mode)
    dec.set_option(efx.OPT_RECON_ITEMS, items)
    dec.set_option(efx.OPT_RECON_WAVES, waves)
    dec.upload(streams, 0)
    dec.set_timing(True)
    for _ in range(4):
        dec.decode(sync=True)
    ts = dec.timing()
    dec.set_option(efx.OPT_GROUPS, 1)
    dec.set_option(efx.OPT_PARSE_CAP, 1)
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
        res.append((1024 * P * 60 / dt / 1e6, tp.index_ms, tp.parse_ms, tp.recon_ms))
    h = dec.frame_hashes()
    ok = all((h[:, dec.picture_slot(p)] == gold[:1024, p]).all() for p in (P - 2, P - 1)) and not any(dec.stream_status(i) for i in range(1024))
    r = sorted(res)[1]
    line = {"flags": flags, "mode": mode, "items_per_wave": items, "waves_per_cu": waves, "parity": bool(ok), "serial_recon_ms": ts.recon_ms, "serial_parse_ms": ts.parse_ms,
            "Mfps_median_of_3": r[0], "index_ms": r[1], "parse_ms": r[2], "recon_ms": r[3], "Mfps_all": [x[0] for x in res],
            "spins": dec.get_option(efx.OPT_RECON_SPINS)}
    print(json.dumps(line), flush=True)
    if ok and mode == 2 and (best is None or r[0] > best["Mfps_median_of_3"]):
        best = line
    dec.close()
print(json.dumps({"best": best}), flush=True)

"""development aid: wall time of ONE efx_decode call at a time (upload resident, sync after every call) for reconstruction group
counts 1 / 2 / 4 / 8 (EFX_OPT_GROUPS) -- what a caller that does not pipeline calls sees.  1024 streams x GOP 12 (or --shape wide)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen
wide = "--shape" in sys.argv and sys.argv[sys.argv.index("--shape") + 1] == "wide"
b = gen.Batch(0, 1024, 12, 12, (4 | 32) if wide else 0)
dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
dec.upload([b.es(k) for k in range(1024)], efx.FORMAT_ES)
dec.decode()
for g in (0, 1, 2, 4, 8):
    dec.set_option(efx.OPT_GROUPS, g)
    dec.decode()
    ts = []
    for _ in range(30):
        dec.sync()
        t0 = time.perf_counter()
        dec.decode(sync=True)
        ts.append(time.perf_counter() - t0)
    print("groups %d: median %.3f ms  min %.3f ms per call (%.2f M frames/s one call at a time)" % (g, statistics.median(ts) * 1e3, min(ts) * 1e3, 12288 / statistics.median(ts) / 1e6))

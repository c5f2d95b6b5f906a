"""development aid: wider parity sweep of the quirk flavours (HIP path vs oracle), ids beyond the committed goldens"""
import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import espflix_amd as efx
import oracle
from espflix_amd import gen
efx.load_library()
bad = 0
for flags, n, P in ((64, 96, 12), (128, 96, 12), (64 | 128 | 4 | 2, 64, 24), (64 | 8 | 16, 64, 12), (128 | 8 | 4 | 2, 64, 24), (64 | 1, 32, 8)):
    b = gen.Batch(1000, n, P, 12, flags)
    es = b.all_es()
    dec = efx.Decoder(n, P, P + 1, max_stream_bytes=sum(e.size for e in es) + 4096)
    dec.upload(es, efx.FORMAT_ES)
    dec.decode()
    h = dec.frame_hashes()
    nb = 0
    for k in range(n):
        cnt, oh, _, _ = oracle.decode(es[k], 0)
        ok = dec.picture_count(k) == cnt == P and dec.stream_status(k) == 0 and all(int(h[k, dec.picture_slot(p, k)]) == int(oh[p]) for p in range(P))
        nb += not ok
    print(f"flags {flags}: {n} streams x {P} pictures, {sum(e.size for e in es) / n / P:.0f} bytes per picture, mismatching streams: {nb}")
    bad += nb
    dec.close()
print("TOTAL mismatches", bad)

import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import espflix_amd as efx
from espflix_amd import gen
efx.load_library()
def run(name, streams, fmt, P):
    dec = efx.Decoder(len(streams), P, P + 1, max_stream_bytes=sum(len(s) for s in streams) + 4096)
    dec.set_timing(True)
    dec.upload(streams, fmt)
    dec.decode()
    t = dec.timing()
    print(name, dec.picture_count(0), dec.stream_status(0), "slices", t.slices, "coefs", t.coefficients)
    dec.close()
for npic in (42, 43, 60):
    b = gen.Batch(0, 1, npic, 12, 4)
    run(f"syn wide es {npic}", [b.es(0)], efx.FORMAT_ES, 60)

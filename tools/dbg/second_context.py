"""development aid: pipelined step time of the first, second and third decoder context of one process"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen
flags = int(os.environ.get("FLAGS", 0))
b = gen.Batch(0, 1024, 12, 12, flags)
blobs = [b.es(k) for k in range(1024)]
for k in range(3):
    dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
    dec.upload(blobs, efx.FORMAT_ES)
    for _ in range(5):
        dec.decode()
    dec.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        dec.decode(sync=False)
    dec.sync()
    dt = (time.perf_counter() - t0) / 100
    print(os.path.basename(os.environ.get("EFX_LIB", "libefx.so")), f"context {k + 1}: pipelined step {dt * 1e3:.3f} ms = {12288 / dt / 1e6:.2f} M frames/s")
    dec.close()

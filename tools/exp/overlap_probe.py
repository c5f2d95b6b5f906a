"""development aid: why does a plain Decoder loop overlap (or not)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get('PROBE_TORCH'):
    import torch
    torch.cuda.set_device(0)
import espflix_amd as efx
from espflix_amd import gen
b = gen.Batch(0, 1024, 12)
blobs = [b.es(k) for k in range(1024)]
mx = max(x.size for x in blobs)
def run(tag, **kw):
    dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2, **kw)
    dec.upload(blobs, efx.FORMAT_ES)
    for _ in range(5):
        dec.decode(sync=False)
    dec.sync()
    dec.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(100):
        dec.decode(sync=False)
    dec.sync()
    dt = (time.perf_counter() - t0) / 100
    t = dec.timing()
    print(tag, 'step %.3f ms' % (dt * 1e3), 'stages index %.3f parse %.3f recon %.3f' % (t.index_ms, t.parse_ms, t.recon_ms))
    dec.close()
tot = sum(x.size for x in blobs)
for k in range(5):
    run('ctx %d' % k, max_stream_bytes=tot + 65536)

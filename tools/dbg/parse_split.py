"""development aid: serial stage times of the bench batch cut to its first n pictures (how long are the I-slice waves?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen
n_streams = int(os.environ.get("N", 1024))
b = gen.Batch(0, n_streams, 12)
for npic in (1, 2, 3, 6, 12):
    blobs = [b.es(k)[:b.picture_offsets(k)[npic]] for k in range(n_streams)]
    dec = efx.Decoder(max_streams=n_streams, max_pictures=12, ring_depth=2)
    dec.upload(blobs, efx.FORMAT_ES)
    dec.decode()
    dec.set_timing(True)
    for _ in range(10):
        dec.decode()
    t = dec.timing()
    print(f"{n_streams} streams, first {npic} pictures: index {t.index_ms:.3f} parse {t.parse_ms:.3f} recon {t.recon_ms:.3f} ms, groups {t.groups}, bytes {sum(len(x) for x in blobs)}")
    dec.close()

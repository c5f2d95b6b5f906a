"""development aid: where do the HIP decoder and the oracle part on a bit-flipped stream?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import espflix_amd as efx
from espflix_amd import gen
import oracle
cases = [tuple(int(v) for v in x) for x in np.load('gpurun_out/flip_cases.npy')]
b = gen.Batch(0, 8, 6)
blobs = []
for k, pos, bit in cases:
    es = b.es(k).copy(); es[pos] ^= 1 << bit; blobs.append(es)
dec = efx.Decoder(max_streams=len(blobs), max_pictures=1, ring_depth=2, max_stream_bytes=sum(x.size for x in blobs) + 65536)
dec.upload(blobs, efx.FORMAT_ES)
orc = [oracle.decode(es, 0, True, want_frames=True) for es in blobs]
done = set()
for p in range(8):
    dec.decode(first_picture=p)
    for i, es in enumerate(blobs):
        n, oh, _, of = orc[i]
        if i in done or p >= n or dec.picture_count(i) < 1:
            continue
        g = dec.download_frame(i, dec.picture_slot(0, i))
        o = of[p]
        if not np.array_equal(g, o):
            done.add(i)
            d = np.nonzero(g != o)[0]
            strips = sorted(set((d // 8448).tolist()))
            rows = sorted(set(((d % 8448) // 528).tolist()))
            cols = d % 528
            luma = d[cols < 352]
            print('case', cases[i], 'FIRST differing picture', p, 'bytes', d.size, 'strips', strips, 'rows', rows[:4], '..', rows[-1], 'luma cols', (int((luma % 528).min()), int((luma % 528).max())) if luma.size else None,
                  'gpu', g[d[:6]], 'oracle', o[d[:6]], 'status', hex(dec.stream_status(i)))

"""development aid: heavily damaged streams (bit flips, overwritten ranges, truncation, inserted start codes, ES and TS input):
every decode must come back, with status bits, never hang or fault"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import espflix_amd as efx
from espflix_amd import gen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(99)
b = gen.Batch(0, 16, 6)
for fmt, name in ((efx.FORMAT_ES, 'ES'), (efx.FORMAT_TS, 'TS')):
    base = [np.frombuffer(b.ts(k), dtype=np.uint8).copy() if fmt == efx.FORMAT_TS else b.es(k).copy() for k in range(16)]
    blobs = []
    for i in range(N):
        x = base[i % 16].copy()
        kind = i % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 17))):
                x[int(rng.integers(0, x.size))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            a = int(rng.integers(0, x.size - 64)); n = int(rng.integers(1, 2000))
            x[a:a + n] = rng.integers(0, 256, min(n, x.size - a), dtype=np.uint8)
        elif kind == 2:
            x = x[:int(rng.integers(1, x.size))].copy()
        elif kind == 3:
            for _ in range(int(rng.integers(1, 9))):
                a = int(rng.integers(0, x.size - 8))
                x[a:a + 4] = [0, 0, 1, int(rng.integers(0, 256))]
        else:
            a = int(rng.integers(0, x.size - 64)); n = int(rng.integers(1, 4000))
            x[a:a + n] = 0
        blobs.append(x)
    dec = efx.Decoder(max_streams=N, max_pictures=16, ring_depth=2, max_stream_bytes=sum(x.size for x in blobs) + 65536)
    t0 = time.perf_counter()
    dec.upload(blobs, fmt)
    dec.decode(first_picture=3)  # (a window that starts inside the stream)
    dec.sync()
    dec.decode()
    st = np.array([dec.stream_status(i) for i in range(N)])
    pc = np.array([dec.picture_count(i) for i in range(N)])
    print(name, 'streams', N, 'decoded in %.2f s' % (time.perf_counter() - t0), 'status histogram', {hex(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
          'pictures min/mean/max', int(pc.min()), round(float(pc.mean()), 1), int(pc.max()))
    dec.close()
print('all decodes returned')

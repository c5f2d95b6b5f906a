"""development aid: single-bit flips in the slice data of synthetic streams, HIP decoder vs the oracle (last two pictures)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import espflix_amd as efx
from espflix_amd import gen
import oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(20260924)
b = gen.Batch(0, 8, 6)
base = [b.es(k) for k in range(8)]
blobs, meta = [], []
for i in range(N):
    es = base[i % 8].copy()
    pos = int(rng.integers(200, es.size - 8))
    bit = int(rng.integers(0, 8))
    es[pos] ^= 1 << bit
    blobs.append(es); meta.append((i % 8, pos, bit))
dec = efx.Decoder(max_streams=N, max_pictures=8, ring_depth=2)
dec.upload(blobs, efx.FORMAT_ES)
dec.decode()
h = dec.frame_hashes()
agree = differ = 0
bad = []
for i, es in enumerate(blobs):
    n, oh, _, _ = oracle.decode(es, 0, True)
    npic = dec.picture_count(i)
    ok = npic == n and all(int(h[i, dec.picture_slot(p, i)]) == int(oh[p]) for p in range(max(0, n - 2), n))  # (two frame buffers, as the reference)
    if ok: agree += 1
    else:
        differ += 1; bad.append((meta[i], npic, n, hex(dec.stream_status(i))))
print('flips', N, 'agree', agree, 'differ', differ)
import collections
def unit(es, pos):
    # the start code unit that holds byte `pos`
    i = pos
    while i >= 3 and not (es[i-3] == 0 and es[i-2] == 0 and es[i-1] == 1 and i - 1 <= pos):
        i -= 1
    code = int(es[i]) if i >= 3 else -1
    return code, pos - i
cls = collections.Counter()
for (m, npic, n, st) in bad:
    k, pos, bit = m
    code, off = unit(base[k], pos)
    kind = 'slice' if 1 <= code <= 0xAF else {0: 'picture', 0xB3: 'sequence', 0xB8: 'gop'}.get(code, hex(code))
    cls[(kind, st, 'same count' if npic == n else 'count differs')] += 1
    if st == '0x0' or kind != 'slice':
        print('  ', m, 'unit', kind, 'offset in unit', off, 'pictures', npic, n, st)
for k, v in sorted(cls.items(), key=lambda x: -x[1]): print(v, k)
np.save('gpurun_out/flip_cases.npy', np.array([m for (m, npic, n, st) in bad if st == '0x0' and npic == n][:16]))

"""development aid (round 6): finer phases of a k_recon wave from a -DEFX_PROBE -DEFX_PROBE_FINE build: stamps (100 MHz) at
start [1], record arrived [3], owner search done / entries requested [2], prediction done + first entries arrived [6], entries
dealt out [5], end [4].   EFX_LIB=espflix_amd/libefx_fine.so python tools/dbg/probe_fine.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen
lib = efx.load_library()
N = 1 << 17
fn = lib.efx_probe_read_recon
fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint)]
b = gen.Batch(0, 1024, 12)
dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
dec.upload([b.es(k) for k in range(1024)], efx.FORMAT_ES)
for _ in range(3):
    dec.decode()
fn(None, 0, None)
dec.decode()
buf = np.zeros((N, 8), dtype=np.uint64)
nxt = C.c_uint(0)
assert fn(buf.ctypes.data, N, C.byref(nxt)) == 0
r = buf[(buf[:, 0] & 0xFF) == 2]
r = r[(r[:, 4] > r[:, 1]) & (r[:, 5] >= r[:, 6]) & (r[:, 6] >= r[:, 2]) & (r[:, 2] >= r[:, 3]) & (r[:, 3] >= r[:, 1])]
t = r.astype(np.int64)
ph = {"start -> record": t[:, 3] - t[:, 1], "record -> search done": t[:, 2] - t[:, 3], "search -> prediction + first entries": t[:, 6] - t[:, 2],
      "-> entries dealt out": t[:, 5] - t[:, 6], "IDCT, sum, stores": t[:, 4] - t[:, 5], "life": t[:, 4] - t[:, 1]}
print(len(r), "waves")
for k, v in ph.items():
    print("%-40s p10 %5.2f  p50 %5.2f  p90 %5.2f  mean %5.2f us" % (k, *(np.percentile(v, q) / 100 for q in (10, 50, 90)), v.mean() / 100))

"""Development aid: per-slice timing of k_parse from the instrumented build (make prof)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EFX_LIB"] = os.path.join(ROOT, "espflix_amd", "libefx_prof.so")
sys.path.insert(0, ROOT)
import espflix_amd as efx
from espflix_amd import gen
S, P = 1024, 12
b = gen.Batch(0, S, P, 12, 0)
es = b.all_es()
dec = efx.Decoder(S, P, 2, max_stream_bytes=sum(e.size for e in es) + 64 * S)
dec.upload(es)
n = S * P * 16
buf = dec.alloc(n * 16)
buf.upload(np.zeros(n * 4, dtype=np.uint32))
lib = efx.load_library()
lib.efx_debug_parse_profile.argtypes = [C.c_void_p, C.c_void_p]
lib.efx_debug_parse_profile(dec._ctx, buf.ptr)
dec.decode(); dec.decode()
r = buf.download(np.uint32, n * 4).reshape(n, 4)
r = r[r[:, 0] > 0]
cyc, iters, coefs, ln = r[:, 0].astype(np.int64), r[:, 1], r[:, 2], r[:, 3]
isI = (ln >> 31) == 1
ln = ln & 0x7FFFFFFF
print("slices", len(r), "I", isI.sum())
w = len(r) // 64
cw = cyc[:w * 64].reshape(w, 64); iw = iters[:w * 64].reshape(w, 64); kw = coefs[:w * 64].reshape(w, 64); lw = ln[:w * 64].reshape(w, 64)
wave_cycles = cw.max(1); wave_iters = iw.sum(1); wave_coefs_max = kw.max(1); wave_coefs_mean = kw.mean(1)
order = np.argsort(-wave_cycles)
print("top waves: cycles, iters, max coefs/lane, mean coefs/lane, len min/max")
for j in order[:8]:
    print(wave_cycles[j], wave_iters[j], wave_coefs_max[j], round(wave_coefs_mean[j], 1), lw[j].min(), lw[j].max())
print("median wave: cycles", np.median(wave_cycles), "iters", np.median(wave_iters))
print("sum over waves cycles", wave_cycles.sum(), "max", wave_cycles.max(), "cycles/iter (top)", wave_cycles[order[0]] / max(1, wave_iters[order[0]]))
print("total iters", wave_iters.sum(), "total coefs", coefs.sum(), "lane efficiency", coefs.sum() / (wave_iters.sum() * 64.0))

"""development aid: start / end times of every k_parse wave of one serial decode (libefx built with -DEFX_DEBUG_WAVES)"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen
L = efx.load_library()
n_streams = int(os.environ.get("N", 512))
flags = int(os.environ.get("FLAGS", 0))
b = gen.Batch(0, n_streams, 12, 12, flags)
dec = efx.Decoder(max_streams=n_streams, max_pictures=12, ring_depth=2)
dec.upload([b.es(k) for k in range(n_streams)], efx.FORMAT_ES)
for _ in range(3):
    dec.decode()
dec.sync()
buf = np.zeros(8 * 16384, dtype=np.uint64)
L.efx_debug_parse_waves.argtypes = [C.c_void_p, C.c_size_t]
# clear, decode once, read
assert L.efx_debug_parse_waves(None, 0) == 0
dec.decode(); dec.sync()
assert L.efx_debug_parse_waves(buf.ctypes.data, buf.size) == 0
w = buf.reshape(-1, 8)
w = w[w[:, 0] != 0]
t0 = w[:, 0].min()
start = (w[:, 0] - t0) / 100.0   # us
end = (w[:, 1] - t0) / 100.0
typ = (w[:, 2] >> 8) & 3
pic = w[:, 2] & 0xFF
hw = w[:, 2] >> 16
print("waves", len(w), "kernel span %.0f us" % end.max())
for t, name in ((1, "I"), (2, "P")):
    m = typ == t
    if m.any():
        d = end[m] - start[m]
        print(name, "waves", m.sum(), "start min/med/max %.0f %.0f %.0f" % (start[m].min(), np.median(start[m]), start[m].max()),
              "dur min/med/p90/max %.0f %.0f %.0f %.0f" % (d.min(), np.median(d), np.percentile(d, 90), d.max()), "end max %.0f" % end[m].max())
# CU placement: hw id bits: wave 0-3, simd 4-5, cu 8-11, sh 12, se 13-15 (gfx9), xcc?; just count distinct (hw >> 4)
print("distinct simd ids", len(set((hw >> 4).tolist())))
allc = np.maximum(w[:, 6].astype(np.float64), 1)
sh_loop, sh_hdr, sh_top = w[:, 3] / allc, w[:, 4] / allc, w[:, 5] / allc
for t, name in ((1, "I"), (2, "P")):
    m = typ == t
    if m.any():
        top = np.argsort(-(end - start) * m)[:16]
        print(name, "share of a wave's cycles: symbol loops (incl. refills inside) median %.2f / longest %.2f, macroblock headers %.2f / %.2f, "
              "ring refills (anywhere) %.2f / %.2f" % (np.median(sh_loop[m]), sh_loop[top].mean(), np.median(sh_hdr[m]), sh_hdr[top].mean(),
                                                      np.median(sh_top[m]), sh_top[top].mean()))
order = np.argsort(-(end - start))[:8]
for i in order:
    print("  longest: pic %d type %d start %.0f dur %.0f" % (pic[i], typ[i], start[i], end[i] - start[i]))

"""development aid: per-wave timing of k_parse / k_recon from a -DEFX_PROBE build (espflix_amd/csrc/efx_probe.h).

    tools/exp/build_variant.sh p "-DEFX_PROBE" && EFX_LIB=espflix_amd/libefx_p.so python tools/dbg/probe_waves.py [serial|pipelined] [flags]

serial: one efx_decode at a time; pipelined: 40 back-to-back calls (the last ones are what the rings hold).
Prints, per kernel: waves, lifetime percentiles (us), and for k_recon the number of waves resident per CU over time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import espflix_amd as efx
from espflix_amd import gen

mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = efx.load_library()
N = 1 << 17


def read(name):
    fn = getattr(lib, "efx_probe_read_" + name)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint)]
    buf = np.zeros((N, 8), dtype=np.uint64)
    nxt = C.c_uint(0)
    assert fn(buf.ctypes.data, N, C.byref(nxt)) == 0
    return buf[: min(nxt.value, N)], nxt.value


def clear():
    for name in ("parse", "recon"):
        fn = getattr(lib, "efx_probe_read_" + name)
        fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint)]
        fn(None, 0, None)


b = gen.Batch(0, 1024, 12, 12, flags)
dec = efx.Decoder(max_streams=1024, max_pictures=12, ring_depth=2)
dec.upload([b.es(k) for k in range(1024)], efx.FORMAT_ES)
for _ in range(3):
    dec.decode()
clear()
if mode == "serial":
    dec.decode()
else:
    for _ in range(6):
        dec.decode(sync=False)
    dec.sync()


def pct(a, qs=(0, 10, 50, 90, 99, 100)):
    return " ".join(f"{np.percentile(a, q):8.1f}" for q in qs)


def hw_cu(hw):
    # HW_ID (gfx9): wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13] ... ; XCC from a different register, so
    # CUs of different XCDs alias here: (se, sh, cu) identifies a CU inside an XCD
    return ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)


for name in ("parse", "recon"):
    r, total = read(name)
    if not len(r):
        continue
    t1 = r[:, 1].astype(np.int64)
    t4 = r[:, 4].astype(np.int64)
    ok = (t4 > t1) & (t1 > 0)
    r, t1, t4 = r[ok], t1[ok], t4[ok]
    t0 = t1.min()
    life = (t4 - t1) / 100.0  # 100 MHz -> us
    print(f"k_{name}: {len(r)} waves recorded ({total} claimed), span {(t4.max() - t0) / 100.0:.1f} us")
    print("  lifetime us  p0 p10 p50 p90 p99 p100:", pct(life))
    if name == "parse":
        t2 = r[:, 2].astype(np.int64)
        t3 = r[:, 3].astype(np.int64)
        trips = r[:, 5].astype(np.int64)
        ptype = (r[:, 6] >> np.uint64(8)) & np.uint64(3)
        print("  staging us  :", pct((t2 - t1) / 100.0))
        print("  pass 1 us   :", pct((t3 - t2) / 100.0))
        print("  pass 2 us   :", pct((t4 - t3) / 100.0))
        print("  trips       :", pct(trips))
        print("  ns per trip :", pct((t3 - t2) * 10.0 / np.maximum(trips, 1)))
        for ty, nm in ((1, "I"), (2, "P")):
            m = ptype == ty
            if m.any():
                print(f"  {nm} waves {m.sum()}: pass1 us {pct((t3 - t2)[m] / 100.0)} | trips {pct(trips[m])} | pass2 {pct((t4 - t3)[m] / 100.0)}")
        # launches: cluster by start time
        order = np.argsort(t1)
        starts = t1[order]
        gaps = np.where(np.diff(starts) > 2000)[0]
        print("  launches seen:", len(gaps) + 1)
    else:
        # resident waves per CU (identified inside an XCD by HW_ID) sampled every 2 us, averaged over busy CU-samples
        hw = (r[:, 0] >> np.uint64(32)).astype(np.int64)
        cu = np.array([hw_cu(int(h)) for h in hw])
        pic = r[:, 6].astype(np.int64)
        span = int(t4.max() - t0)
        ts = np.arange(0, span, 200)
        res = np.zeros(len(ts))
        for i, t in enumerate(ts):
            m = (t1 - t0 <= t) & (t4 - t0 > t)
            res[i] = m.sum()
        print(f"  resident waves chip-wide: mean {res.mean():.0f} max {res.max():.0f} (256 CUs: {res.mean() / 256:.1f} per CU)")
        for p in sorted(set(pic.tolist()))[:12]:
            m = pic == p
            print(f"  picture {p}: {m.sum()} waves, lifetime {pct(life[m])}, launch span {(t4[m].max() - t1[m].min()) / 100.0:.1f} us")
dec.close()

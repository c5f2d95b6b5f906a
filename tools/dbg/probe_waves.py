"""development aid: per-wave timing of k_parse / k_recon from a -DEFX_PROBE build (espflix_amd/csrc/efx_probe.h).

    tools/exp/build_variant.sh p "-DEFX_PROBE" && EFX_LIB=espflix_amd/libefx_p.so python tools/dbg/probe_waves.py [serial|pipelined] [flags]

serial: one efx_decode at a time; pipelined: back-to-back calls (the rings hold the last launches).
Prints, per kernel: waves, lifetime percentiles (us); for k_recon per launch: how long a wave lives and how many k_recon /
k_parse waves a CU holds while the launch runs (every CU of the chip identified by XCC_ID + HW_ID)."""
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
    return buf


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
    # k_recon keeps five launches of 1024 streams: the last ones of ONE hand-over epoch in the middle of a run of
    # back-to-back calls (a slot's epoch advances once per call that uses it: three slots, one group per call)
    lib.efx_probe_window_recon.argtypes = [C.c_uint, C.c_uint]
    lib.efx_probe_window_recon(5, 5)
    for _ in range(16):
        dec.decode(sync=False)
    dec.sync()


def pct(a, qs=(0, 10, 50, 90, 99, 100)):
    return " ".join(f"{np.percentile(a, q):8.1f}" for q in qs)


def cu_key(r):
    hw = (r[:, 0] >> np.uint64(32)).astype(np.int64)
    xcc = r[:, 7].astype(np.int64) & 15
    return (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)


def valid(r):
    t1 = r[:, 1].astype(np.int64)
    t4 = r[:, 4].astype(np.int64)
    ok = (t4 > t1) & (t1 > 0)
    return r[ok], t1[ok], t4[ok]


parse, p1, p4 = valid(read("parse"))
recon, r1, r4 = valid(read("recon"))
if len(parse):
    life = (p4 - p1) / 100.0
    t2 = parse[:, 2].astype(np.int64)
    t3 = parse[:, 3].astype(np.int64)
    trips = parse[:, 5].astype(np.int64)
    print(f"k_parse: {len(parse)} waves recorded")
    print("  lifetime us  p0 p10 p50 p90 p99 p100:", pct(life))
    print("  pass 1 us   :", pct((t3 - t2) / 100.0))
    print("  pass 2 us   :", pct((p4 - t3) / 100.0))
    print("  trips       :", pct(trips))
    print("  ns per trip :", pct((t3 - t2) * 10.0 / np.maximum(trips, 1)))
if len(recon):
    life = (r4 - r1) / 100.0
    print(f"k_recon: {len(recon)} waves recorded; lifetime us p0 p10 p50 p90 p99 p100: {pct(life)}")
    cyc = recon[:, 2].astype(np.int64)
    t3r = recon[:, 3].astype(np.int64)
    t5r = recon[:, 5].astype(np.int64)
    okc = (cyc > 0) & (t3r >= r1) & (t5r >= t3r) & (r4 >= t5r)
    if okc.any():
        print("  shader clock GHz (cycles / lifetime)  :", pct(cyc[okc] / (life[okc] * 1000.0)))
        print("  start -> record arrived us            :", pct((t3r[okc] - r1[okc]) / 100.0))
        print("  record -> entries dealt out us        :", pct((t5r[okc] - t3r[okc]) / 100.0))
        print("  IDCT, sum with the prediction, stores :", pct((r4[okc] - t5r[okc]) / 100.0))
    tagv = recon[:, 6].astype(np.int64)
    rcu = cu_key(recon)
    pcu = cu_key(parse) if len(parse) else np.zeros(0, np.int64)
    print("  CUs seen:", len(set(rcu.tolist())))
    rows = []
    for tg in sorted(set(tagv.tolist()), key=lambda t: r1[tagv == t].min()):
        m = tagv == tg
        if m.sum() < 1000:
            continue
        a, z = r1[m].min(), r4[m].max()
        ts = np.linspace(a, z, 12)[1:-1]
        res_r, res_p = [], []
        for t in ts:
            alive = m & (r1 <= t) & (r4 > t)
            cnt = np.bincount(rcu[alive], minlength=4096)
            res_r.append(cnt[cnt > 0].mean() if (cnt > 0).any() else 0)
            if len(parse):
                pa = (p1 <= t) & (p4 > t)
                cp = np.bincount(pcu[pa], minlength=4096)
                res_p.append(cp.sum() / 256.0)
        rows.append((tg & 0xFF, (tg >> 8) & 0xFF, tg >> 16, m.sum(), (z - a) / 100.0, np.median(life[m]), np.percentile(life[m], 90),
                     np.mean(res_r), np.mean(res_p) if res_p else 0.0))
    print("  per launch: picture epoch stream0 waves | span us | wave life p50 p90 us | k_recon waves per busy CU | k_parse waves per CU")
    for row in rows[-26:]:
        print("   pic %2d ep %3d s0 %4d  %5d | %6.1f | %6.1f %6.1f | %5.1f | %4.2f" % row)
dec.close()

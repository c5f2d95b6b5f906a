"""Quick on-GPU parity + timing check (development aid; the real checks live in tests/)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import espflix_amd as efx
from espflix_amd import gen
import oracle

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ok = True

def compare(dec, streams, fmt, label):
    global ok
    dec.upload(streams, fmt)
    dec.decode()
    hashes = dec.frame_hashes(0, len(streams))
    for i, s in enumerate(streams):
        n, oh, opts, _ = oracle.decode(s, fmt, flush_last=True)
        npic = dec.picture_count(i)
        gh = [hashes[i, dec.picture_slot(p)] for p in range(npic)]
        bad = [p for p in range(min(n, npic)) if int(gh[p]) != int(oh[p])]
        gpts = [dec.picture_pts(i, p) for p in range(npic)]
        pbad = [p for p in range(min(n, npic)) if gpts[p] != opts[p]]
        st = dec.stream_status(i)
        flag = "OK" if (not bad and n == npic and not pbad and st == 0) else "FAIL"
        if flag != "OK":
            ok = False
        print(f"{label}[{i}] pictures gpu={npic} oracle={n} status={st} first mismatches={bad[:6]} pts_bad={pbad[:4]} {flag}")

# 1. embedded clips (real ffmpeg streams), TS input
clips = [np.fromfile(os.path.join(G, n), dtype=np.uint8) for n in ("splash.ts", "vmedia.ts")]
dec = efx.Decoder(max_streams=2, max_pictures=100, ring_depth=101, max_stream_bytes=2 << 20)
compare(dec, clips, efx.FORMAT_TS, "clip")
dec.close()

# 2. synthetic streams, ES input, several flavours
for flags in (0, gen.FLAG_I_ONLY, gen.FLAG_CUSTOM_MATRICES, gen.FLAG_WIDE_SLICES, gen.FLAG_LONG_SKIPS, gen.FLAG_FLAT_BRIGHT):
    b = gen.Batch(0, 8, 12, 12, flags)
    dec = efx.Decoder(max_streams=8, max_pictures=12, ring_depth=13)
    compare(dec, b.all_es(), efx.FORMAT_ES, f"syn{flags}")
    dec.close()

# 3. timing, 256 streams x 12 pictures
b = gen.Batch(0, 256, 12, 12, 0)
es = b.all_es()
dec = efx.Decoder(max_streams=256, max_pictures=12, ring_depth=2)
dec.upload(es, efx.FORMAT_ES)
dec.set_timing(True)
for it in range(3):
    dec.decode()
    t = dec.timing()
    print(f"timing: index {t.index_ms:.3f} parse {t.parse_ms:.3f} recon {t.recon_ms:.3f} total {t.total_ms:.3f} ms; "
          f"pictures {t.pictures} slices {t.slices} coefs {t.coefficients} es {t.es_bytes} -> {t.pictures / t.total_ms * 1e3:.0f} frames/s")
dec.close()
print("ALL OK" if ok else "SOME FAILED")

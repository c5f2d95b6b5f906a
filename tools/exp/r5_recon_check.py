#!/usr/bin/env python3
"""Development aid (round 5): the reconstruction launch structures side by side on one box.

For EFX_OPT_RECON_MODE 0 (one k_recon launch per picture index), 1 and 2 (one persistent k_recon_all per call):
every picture of 256 streams against the reference decoder's hashes (tests/golden/bench_gop12.u64), batches of
1 / 3 / 8 / 40 streams (the dependency waits at the tail of a small batch), then 1024 streams x GOP 12 timed one call at a
time and back to back.  Prints one JSON line per configuration.  Usage: r5_recon_check.py [quick]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import espflix_amd as efx
from espflix_amd import gen

P = 12
golden = np.fromfile(os.path.join(ROOT, "tests", "golden", "bench_gop12.u64"), dtype="<u8").reshape(8192, P)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
LIBS = [x for x in os.environ.get("EFX_CHECK_LIBS", "").split(",") if x]  # e.g. "b,c,d": espflix_amd/libefx_<x>.so, after the default


def use_lib(tag):
    """Switch the ctypes binding to another build of the library (its own globals; contexts of the old one are closed)."""
    efx._lib = None
    efx.LIB_PATH = os.path.join(ROOT, "espflix_amd", f"libefx_{tag}.so" if tag else "libefx.so")
    efx.load_library()


def check(dec, streams, ids, what):
    dec.upload(streams, 0)
    dec.decode()
    h = dec.frame_hashes()
    bad = 0
    for p in range(P):
        bad += int((h[:len(streams), dec.picture_slot(p)] != golden[ids, p]).sum())
    st = sum(dec.stream_status(i) != 0 for i in range(len(streams)))
    spins = dec.get_option(efx.OPT_RECON_SPINS)
    if bad or st:
        print(json.dumps({"FAIL": what, "bad_pictures": bad, "flagged": st, "spins": spins}), flush=True)
    return bad == 0 and st == 0, spins


def main():
    threads = max(1, (os.cpu_count() or 2) // 2)
    b = gen.Batch(0, 1024, P, 12, 0, threads)
    streams = b.all_es()
    es_bytes = sum(s.size for s in streams)
    ok_all = True
    for tag in [""] + LIBS:
      use_lib(tag)
      full = not quick and tag == ""
      print(json.dumps({"library": efx.LIB_PATH}), flush=True)
      for mode in ((0, 2) if tag == "" else (2,)):
            # ---- parity: every picture kept ------------------------------------------------------------------------------
            for n in ((256, 1, 3, 8, 40) if full or tag == '' else (256, 8)):
                dec = efx.Decoder(n, P, P + 1, max_stream_bytes=sum(s.size for s in streams[:n]) + 4096)
                dec.set_option(efx.OPT_RECON_MODE, mode)
                ok, spins = check(dec, streams[:n], np.arange(n), f"mode {mode}, {n} streams, ring {P + 1}")
                ok_all &= ok or tag != ""
                print(json.dumps({"lib": tag, "mode": mode, "streams": n, "parity": ok, "spins": spins}), flush=True)
                dec.close()
            # ---- 1024 streams, the reference's two frame buffers: repeated calls, then timing ---------------------------------
            for items in ((16,) if (mode == 0 or not full) else (16, 4, 8, 32, 64, 0)):
                dec = efx.Decoder(1024, P, 2, max_stream_bytes=es_bytes + 64 * 1024)
                dec.set_option(efx.OPT_RECON_MODE, mode)
                dec.set_option(efx.OPT_RECON_ITEMS, items)
                dec.upload(streams, 0)
                for _ in range(3):
                    dec.decode(sync=False)
                dec.sync()
                h = dec.frame_hashes()
                ok = all((h[:, dec.picture_slot(p)] == golden[:1024, p]).all() for p in (P - 2, P - 1))
                ok &= not any(dec.stream_status(i) for i in range(1024))
                ok_all &= ok or tag != ""
                dec.set_timing(True)
                for _ in range(5):
                    dec.decode(sync=True)
                ts = dec.timing()
                # back to back, pinned structure (one group, capped parser) and the automatic one
                res = {}
                for name, groups, cap in (("auto", 0, 0), ("pinned", 1, 1)):
                    dec.set_option(efx.OPT_GROUPS, groups)
                    dec.set_option(efx.OPT_PARSE_CAP, cap)
                    for _ in range(5):
                        dec.decode(sync=False)
                    dec.sync()
                    dec.set_timing(True)
                    t0 = time.perf_counter()
                    for _ in range(40):
                        dec.decode(sync=False)
                    dec.sync()
                    dt = time.perf_counter() - t0
                    tp = dec.timing()
                    res[name] = {"ms_per_step": dt / 40 * 1e3, "Mfps": 1024 * P * 40 / dt / 1e6, "recon_ms": tp.recon_ms, "parse_ms": tp.parse_ms,
                                 "mixed": tp.mixed}
                dec.set_option(efx.OPT_GROUPS, 0)
                dec.set_option(efx.OPT_PARSE_CAP, 0)
                h = dec.frame_hashes()
                ok2 = all((h[:, dec.picture_slot(p)] == golden[:1024, p]).all() for p in (P - 2, P - 1))
                ok_all &= ok2 or tag != ""
                st = dec.recon_stats() if mode else None
                if st and st[16]:
                    # -DEFX_RA_STATS build: phase times of the most recent call's waves, units of 40 ns summed over the launch
                    w, it = st[16], max(1, st[17])
                    print(json.dumps({"lib": tag, "mode": mode, "stats_of_last_call": {
                        "waves": w, "items": st[17], "items_on_a_foreign_xcd": st[18],
                        "us_per_item": {"requests_for_the_next_items": st[19] * 0.04 / it, "dependency_wait": st[20] * 0.04 / it,
                                        "body": st[21] * 0.04 / it},
                        "mean_wave_life_us": st[23] * 0.04 / w, "waves_by_xcc": st[24:32]}}), flush=True)
                print(json.dumps({"lib": tag, "mode": mode, "items_per_wave": items, "parity_after_repeats": bool(ok), "parity_after_timing": bool(ok2),
                                  "serial_ms": {"index": ts.index_ms, "parse": ts.parse_ms, "recon": ts.recon_ms},
                                  "spins": dec.get_option(efx.OPT_RECON_SPINS), "back_to_back": res}), flush=True)
                dec.close()
    print(json.dumps({"ALL_OK": bool(ok_all)}), flush=True)
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
